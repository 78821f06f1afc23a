"""Data-parallel gradient exchange: one process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" on ROCm) - replaces the reference's single-process nn.DataParallel
(PyMIC/pymic/net_run_dsbn/agent_seg.py:692-698: replicate / scatter / gather / reduce_add).

The only collective of the hot path is the all-reduce(sum) of the flat fp32 gradient buffer
(fplx.net: shared parameters in gradient-PRODUCTION order, then per-domain BN tails).  It is
issued in a few large buckets as soon as backward has finished the layers of a bucket, so the
transfer of the deep (parameter-heavy, compute-light) blocks hides behind the backward of the
shallow (compute-heavy, parameter-light) ones.  BatchNorm statistics stay per rank, as they do
per replica under DataParallel; division by world size is folded into the Adam kernel.

Works on any backend: the CPU tests run it over gloo with world_size 2.
"""
import os

import torch
import torch.distributed as dist


class GradAllReducer(object):
    def __init__(self, bucket_ranges, domain_ranges, group=None):
        self.buckets = list(bucket_ranges)      # [(start, end)] ascending, contiguous from 0
        self.domain_ranges = list(domain_ranges)
        self.group = group
        # FPLX_DDP_FORCE=1: run the collectives even with a single rank (exercises the RCCL path on a 1-GPU box)
        forced = os.environ.get("FPLX_DDP_FORCE", "0") == "1"
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or forced)
        self.world = dist.get_world_size(group) if self.enabled else 1
        self._works, self._next, self._gflat = [], 0, None

    def begin(self, gflat):
        self._works, self._next, self._gflat = [], 0, gflat

    def pending(self, end):
        """would ready(end) launch a collective?  (the engine joins its weight-gradient stream only then)"""
        return self.enabled and self._next < len(self.buckets) and self.buckets[self._next][1] <= end

    def ready(self, end):
        """gradients of flat elements [0, end) are final: launch every bucket that is complete"""
        if not self.enabled:
            return
        while self._next < len(self.buckets) and self.buckets[self._next][1] <= end:
            s, e = self.buckets[self._next]
            self._works.append(dist.all_reduce(self._gflat[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                               async_op=True))
            self._next += 1

    def finish(self, active_domains):
        """flush remaining buckets + the BN segments of the domains used in this step; wait."""
        if not self.enabled:
            return
        self.ready(self.buckets[-1][1])
        for d in active_domains:
            s, e = self.domain_ranges[d]
            self._works.append(dist.all_reduce(self._gflat[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                               async_op=True))
        for w in self._works:
            w.wait()
        self._works = []


def broadcast_buffers_from_rank0(net, group=None):
    """running_mean / running_var of rank 0 win, as replica 0's do under nn.DataParallel."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for b in net.buffers():
        dist.broadcast(b, 0, group=group)


def broadcast_params_from_rank0(net, group=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    net._ensure_flat()
    dist.broadcast(net.flat_params, 0, group=group)
    broadcast_buffers_from_rank0(net, group)

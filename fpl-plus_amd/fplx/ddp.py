"""Data-parallel gradient exchange: one process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" on ROCm) - replaces the reference's single-process nn.DataParallel
(PyMIC/pymic/net_run_dsbn/agent_seg.py:692-698: replicate / scatter / gather / reduce_add).

The only collective of the hot path is the all-reduce(sum) of the flat fp32 gradient buffer
(fplx.net: shared parameters in gradient-PRODUCTION order, then per-domain BN tails).  It is
issued in a few large buckets as soon as backward has finished the layers of a bucket, so the
transfer of the deep (parameter-heavy, compute-light) blocks hides behind the backward of the
shallow (compute-heavy, parameter-light) ones.  BatchNorm statistics stay per rank, as they do
per replica under DataParallel.  Nothing is divided by the world size: the loss is evaluated
over the FULL batch of all ranks (fplx_seg_loss_sums -> all-reduce of the batch totals ->
fplx_seg_loss_from_sums), so the ranks' gradients ADD UP to its gradient (train.py, loss.py).

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
        self._works, self._next, self._gflat, self._acc = [], 0, None, None
        self.launched = []                      # (start, end) of every collective of the current step, in launch order (tests)

    def begin(self, gflat, acc=None):
        """gflat: the buffer that is all-reduced (and that Adam reads).
        acc: the buffer the CURRENT backward writes into when it is not gflat itself (the later domains of a training_all
        iteration, agent_seg.py:459-495: gflat already holds the earlier domains' sum): a range is then first added,
        gflat[s:e] += acc[s:e], on the stream the hook runs on, and all-reduced behind that - bucket by bucket while the
        backward still runs, instead of one add of the whole buffer and all collectives behind the last kernel."""
        self._works, self._next, self._gflat, self._acc = [], 0, gflat, acc
        self.launched = []

    def set_acc(self, acc):
        self._acc = acc

    def pending(self, end):
        """would ready(end) launch a collective?  (the engine joins its weight-gradient stream only then)"""
        return self.enabled and self._next < len(self.buckets) and self.buckets[self._next][1] <= end

    def _reduce(self, s, e):
        if self._acc is not None:
            self._gflat[s:e].add_(self._acc[s:e])
        self._works.append(dist.all_reduce(self._gflat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.launched.append((s, e))

    def ready(self, end):
        """gradients of flat elements [0, end) are final: launch every bucket that is complete"""
        if not self.enabled:
            return
        while self._next < len(self.buckets) and self.buckets[self._next][1] <= end:
            s, e = self.buckets[self._next]
            self._reduce(s, e)
            self._next += 1

    def reduce_domain(self, d):
        """the BatchNorm segment of domain d is final (its pass is over; no other pass writes it): all-reduce it now.
        Always from gflat itself - the caller has already folded a non-first pass's buffer in."""
        if not self.enabled:
            return
        s, e = self.domain_ranges[d]
        acc, self._acc = self._acc, None
        try:
            self._reduce(s, e)
        finally:
            self._acc = acc

    def finish(self, active_domains):
        """flush remaining buckets + the BN segments of the listed domains (those not yet sent by reduce_domain); wait."""
        if not self.enabled:
            return
        self.ready(self.buckets[-1][1])
        for d in active_domains:
            s, e = self.domain_ranges[d]
            self._reduce(s, e)
        for w in self._works:
            w.wait()
        self._works = []


def init_from_env(backend=None, device=None):
    """One process per GPU, launched by `python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment): select the GPU and create the process group - BEFORE anything else touches the device.  Returns True
    when a group with more than one rank (or FPLX_DDP_FORCE=1) is up.  Backend: nccl (= RCCL over xGMI); FPLX_DDP_BACKEND
    overrides it (the 2-rank tests on one GPU use gloo).
    device: the CUDA device index this rank's tensors will live on (default: LOCAL_RANK).  The process group is bound to
    THAT device - a config whose `gpus` list is not [0 .. n-1] must pass the same index the agent places its network on,
    or the collectives run on a device the tensors are not on."""
    if not dist.is_available():
        return False
    if dist.is_initialized():
        return dist.get_world_size() > 1 or os.environ.get("FPLX_DDP_FORCE", "0") == "1"
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        return False
    world = int(os.environ["WORLD_SIZE"])
    if world == 1 and os.environ.get("FPLX_DDP_FORCE", "0") != "1":
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or os.environ.get("FPLX_DDP_BACKEND", "nccl")
    local = local_rank() if device is None else int(device)
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return True


def local_rank():
    return int(os.environ.get("LOCAL_RANK", "0"))


def active(group=None):
    return dist.is_available() and dist.is_initialized() and (
        dist.get_world_size(group) > 1 or os.environ.get("FPLX_DDP_FORCE", "0") == "1")


def rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def barrier(group=None):
    if dist.is_available() and dist.is_initialized():
        dist.barrier(group=group)


def gather_objects_to_rank0(obj, group=None):
    """-> list of every rank's object on rank 0 (None elsewhere); [obj] without a process group"""
    if not active(group):
        return [obj]
    out = [None] * world_size(group) if rank(group) == 0 else None
    dist.gather_object(obj, out, dst=0, group=group)
    return out


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

"""Fused Adam + lr schedule with the reference's optimiser semantics
(reference: PyMIC/pymic/net_run_dsbn/get_optimizer.py:9-57: optim.Adam(params, lr, weight_decay=wd)
with torch defaults betas (0.9, 0.999), eps 1e-8; MultiStepLR(milestones, gamma, last_epoch)).

torch.optim.Optimizer is subclassed only so that torch's lr schedulers, `param_groups` and
`state_dict` plumbing keep working; the update itself is ONE HIP kernel launch per flat
segment (shared parameters, each domain's BN affine parameters): parameters whose gradient is
None are skipped entirely, exactly like torch.optim.Adam - with DSBN that is every BN set of
the domains that took no part in the step (dsbn.py:56).
"""
import torch
from torch.optim import Optimizer, lr_scheduler

from . import ops


def keyword_match(a, b):
    return a.lower() == b.lower()


class FusedAdam(Optimizer):
    def __init__(self, net, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        net._ensure_flat()
        self.net = net
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super(FusedAdam, self).__init__([net.get_param(k) for k in net._order], defaults)
        self.exp_avg = torch.zeros_like(net.flat_params)
        self.exp_avg_sq = torch.zeros_like(net.flat_params)
        shared, doms = net.segments()
        self.seg_ranges = [shared] + doms
        self.seg_steps = [0] * len(self.seg_ranges)
        self.grad_scale = 1.0            # e.g. 1/world_size after an all-reduce(sum)
        # data parallelism in autograd mode (SegmentationAgent): step() all-reduces (sum) the gradients first - the loss was
        # evaluated over the full batch of all ranks (fplx.loss dist_sync), so the sum IS its gradient
        self.dist_sync, self.dist_group = False, None

    def _segment_grad(self, start, end):
        """flat gradient tensor covering [start, end) if every parameter's .grad is the matching
        view of ONE flat buffer; 'none' if all grads are None; else None (mixed)."""
        net = self.net
        base, any_grad, all_grad, contiguous = None, False, True, True
        for k in net._order:
            o, n, _ = net._layout[k]
            if o < start or o >= end:
                continue
            g = net.get_param(k).grad
            if g is None:
                all_grad = False
                continue
            any_grad = True
            if not g.is_contiguous() or g.dtype != torch.float32:
                contiguous = False
                continue
            b = g.data_ptr() - (o - start) * 4
            if base is None:
                base = (b, g)
            elif b != base[0]:
                contiguous = False
        if not any_grad:
            return "none"
        if all_grad and contiguous:
            # rebuild a flat view over the underlying storage (all grads are views of one buffer)
            first = None
            for k in net._order:
                o, n, _ = net._layout[k]
                if o == start:
                    first = net.get_param(k).grad
                    break
            need = (first.storage_offset() + end - start) * 4
            if first.untyped_storage().nbytes() >= need:
                return torch.as_strided(first, (end - start,), (1,), first.storage_offset())
        return None

    @torch.no_grad()
    def step(self, closure=None):
        net = self.net
        net._ensure_flat()
        net.engine.invalidate()                 # raw-pointer update below: packs and eval-mode folds are stale afterwards
        group = self.param_groups[0]
        lr, (b1, b2), eps, wd = group['lr'], group['betas'], group['eps'], group['weight_decay']
        for si, (start, end) in enumerate(self.seg_ranges):
            g = self._segment_grad(start, end)
            if isinstance(g, str):
                continue                                        # whole segment has no gradient: skipped
            self.seg_steps[si] += 1
            if self.dist_sync:
                import torch.distributed as dist
                if g is not None:
                    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.dist_group)
                else:
                    for k in net._order:
                        o, n, _ = net._layout[k]
                        p = net.get_param(k)
                        if start <= o < end and p.grad is not None:
                            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.dist_group)
            if g is not None:
                ops.adam_step(net.flat_params[start:end], g, self.exp_avg[start:end], self.exp_avg_sq[start:end],
                              lr, self.seg_steps[si], wd, self.grad_scale, (b1, b2), eps)
            else:                                               # gradients not flat: one launch per tensor
                for k in net._order:
                    o, n, _ = net._layout[k]
                    p = net.get_param(k)
                    if o < start or o >= end or p.grad is None:
                        continue
                    gg = p.grad.float().contiguous()
                    ops.adam_step(net.flat_params[o:o + n], gg, self.exp_avg[o:o + n], self.exp_avg_sq[o:o + n],
                                  lr, self.seg_steps[si], wd, self.grad_scale, (b1, b2), eps)
        net.engine.invalidate()

    @torch.no_grad()
    def step_flat(self, gflat, active_domains):
        """engine-mode update: gflat is laid out like net.flat_params; only the shared segment and
        the listed domains' BN segments are updated."""
        net = self.net
        net.engine.invalidate()                 # the update goes through raw pointers: packs and eval-mode folds are stale now
        self._opt_called = True                 # torch's lr schedulers check that an optimiser step preceded theirs
        group = self.param_groups[0]
        lr, (b1, b2), eps, wd = group['lr'], group['betas'], group['eps'], group['weight_decay']
        fused = False
        for si, (start, end) in enumerate(self.seg_ranges):
            if si > 0 and (si - 1) not in active_domains:
                continue
            self.seg_steps[si] += 1
            plan = net.engine.adam_pack_plan() if si == 0 else None
            if plan:
                # the shared segment: Adam AND the bf16 packs of its 3x3x3 weights in one launch - the next forward finds
                # them in place instead of re-reading every master weight (get_optimizer.py:17 + unet2d5_dsbn.py:54-55)
                assert start == 0 and all(l[0] + l[1] * l[2] * 27 <= end for l in plan)
                ops.adam_pack_step(net.flat_params[start:end], gflat[start:end], self.exp_avg[start:end],
                                   self.exp_avg_sq[start:end], lr, self.seg_steps[si], wd, self.grad_scale, (b1, b2), eps, plan)
                fused = True
                continue
            ops.adam_step(net.flat_params[start:end], gflat[start:end], self.exp_avg[start:end],
                          self.exp_avg_sq[start:end], lr, self.seg_steps[si], wd, self.grad_scale, (b1, b2), eps)
        if fused:
            net.engine.packs_written_by_optimizer()

    def state_dict(self):
        """torch.optim.Adam's layout over the reference's parameter list (fplx/checkpoint.py): what the reference's
        agent saves as 'optimizer_state_dict' and what its create_optimizer loads back (agent_abstract.py:327-330)"""
        from .checkpoint import optimizer_to_reference
        return optimizer_to_reference(self)

    def flat_state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "seg_steps": list(self.seg_steps),
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        if "state" in sd:                                       # a torch.optim.Adam / reference checkpoint
            from .checkpoint import optimizer_from_reference
            return optimizer_from_reference(self, sd)
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.seg_steps = list(sd["seg_steps"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)


def get_optimizer(name, net, optim_params):
    """get_optimizer (get_optimizer.py:9-36): Adam is fused; anything else is refused loudly."""
    lr = optim_params['learning_rate']
    weight_decay = optim_params['weight_decay']
    if keyword_match(name, "Adam"):
        return FusedAdam(net, lr, weight_decay=weight_decay)
    raise ValueError("unsupported optimizer {0:}".format(name))               # get_optimizer.py:36


def get_lr_scheduler(optimizer, sched_params):
    """get_lr_scheduler (get_optimizer.py:39-57)."""
    name = sched_params["lr_scheduler"]
    if name is None:
        return None
    lr_gamma = sched_params["lr_gamma"]
    if keyword_match(name, "ReduceLROnPlateau"):
        patience = sched_params["ReduceLROnPlateau_patience".lower()] / sched_params["iter_valid"]
        return lr_scheduler.ReduceLROnPlateau(optimizer, mode="max", factor=lr_gamma, patience=patience)
    if keyword_match(name, "MultiStepLR"):
        return lr_scheduler.MultiStepLR(optimizer, sched_params["lr_milestones"], lr_gamma,
                                        sched_params["last_iter"])
    raise ValueError("unsupported lr scheduler {0:}".format(name))

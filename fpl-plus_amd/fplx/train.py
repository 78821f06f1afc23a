"""Engine-mode training step: forward -> fused loss -> backward -> (bucketed all-reduce) ->
fused Adam, all on flat buffers, no autograd bookkeeping, no host synchronisation.

One `step()` = one batch of ONE domain, i.e. one pass of the loop body of
SegmentationAgent.training (reference PyMIC/pymic/net_run_dsbn/agent_seg.py:336-357) with the
backward + optimizer.step the published loop forgot (it exists in training_all, 490-494, and in
the vanilla agent, net_run/agent_seg.py:153-159).  `step_all()` = one iteration of training_all
(agent_seg.py:459-495): every domain forward, loss (l0 + l1)/2, ONE Adam step.
"""
import torch

from . import ops, _lib
from .ddp import GradAllReducer
from .optim import FusedAdam


class TrainStep(object):
    def __init__(self, net, loss_terms=(1.0, 0.0, 0.0, 0.0), softmax=True, lr=1e-4, weight_decay=1e-5,
                 milestones=(), gamma=0.5, group=None, bucket_elems=1 << 21, optimizer=None):
        """optimizer: an existing fplx.FusedAdam over `net` (SegmentationAgent routes its training loops through this class and
        keeps ITS optimiser - state, param_groups, the torch lr scheduler stepping it); its learning rate is then whatever the
        param_group says at the time of the step, `lr` / `milestones` / `gamma` are not used."""
        net._ensure_flat()
        net.train()
        self.net = net
        self.terms = tuple(float(t) for t in loss_terms)
        self.softmax = bool(softmax)
        self.external_lr = optimizer is not None
        self.opt = optimizer if optimizer is not None else FusedAdam(net, lr, weight_decay=weight_decay)
        self.base_lr, self.milestones, self.gamma = lr, sorted(milestones), gamma
        self.iteration = 0
        self.gflat = torch.zeros_like(net.flat_params)
        self.gacc = None
        shared, doms = net.segments()
        if bucket_elems == 1 << 21:
            bucket_elems = int(_lib.host_knob("bucket_elems"))          # A/B switch (fplx/_lib.py:_HOST_KNOBS)
        self.reducer = GradAllReducer(net.bucket_ranges(bucket_elems), doms, group)
        # the loss is evaluated over the FULL batch of all ranks (reference: nn.DataParallel gathers the logits, one loss):
        # the ranks' gradients add up to its gradient, nothing is divided by the world size
        self.group = group
        self.opt.grad_scale = 1.0
        self.dist_loss = self.reducer.enabled
        self.overlap_all = _lib.host_knob("ddp_overlap_all") != 0      # A/B switch: step_all's collectives behind the last backward
        self._one = torch.ones(1, dtype=torch.float32, device=net.flat_params.device)
        self._half = torch.full((1,), 0.5, dtype=torch.float32, device=net.flat_params.device)
        self._loss_bufs = {}

    def _lr(self):
        return self.base_lr * self.gamma ** sum(1 for m in self.milestones if self.iteration >= m)

    def _loss_buffers(self, n, c, v, dev):
        key = (n, c, v)
        if key not in self._loss_bufs:
            self._loss_bufs[key] = (
                torch.empty((n, ops.loss_rows(v), ops.loss_k(c)), dtype=torch.float32, device=dev),
                torch.empty(n * c * 2 + 2, dtype=torch.float32, device=dev))
        return self._loss_bufs[key]

    def _fwd_bwd(self, x, label, domain, pw, iw, gscale, gflat, reduce_hook, reuse_packs=False):
        net = self.net
        step = net._fwd_counter
        net._fwd_counter += 1
        logits, sv = net.engine.forward(x, domain, True, net.dropout_active(), net.dropout_seed, step, keep=True,
                                        reuse_packs=reuse_packs)
        n, c = logits.shape[0], logits.shape[1]
        v = logits[0, 0].numel()
        if logits.shape != label.shape:                                       # as fplx.loss._FusedSegLoss.forward
            raise ValueError("fplx loss: prediction {0:} and ground_truth {1:} differ in shape".format(
                tuple(logits.shape), tuple(label.shape)))
        if self.terms[2] != 0.0 and (pw is None or iw is None):
            raise KeyError('pixel_weight')                                    # dice.py:109-110 index the dict
        if self.terms[2] == 0.0:
            iw = None
        label = label.float().contiguous()
        if pw is not None:
            pw = pw.float().contiguous()
            if pw.numel() != n * v:
                raise ValueError("fplx loss: pixel_weight must be [N,1,D,H,W]")
        if iw is not None:
            iw = iw.float().contiguous()
        part, coef = self._loss_buffers(n, c, v, logits.device)
        out = torch.empty(4 + c, dtype=torch.float32, device=logits.device)
        if self.dist_loss:
            ops.seg_loss_fwd_dist(logits, label, pw, iw, self.terms, self.softmax, part, out, coef, self.group)
        else:
            ops.seg_loss_fwd(logits, label, pw, iw, self.terms, self.softmax, part, out, coef)
        dlogits = torch.empty_like(logits)
        ops.seg_loss_bwd(logits, label, pw, coef, gscale, self.terms, self.softmax, dlogits)
        net.engine.backward(sv, dlogits, gflat, reduce_hook)
        return out

    def step(self, x, label, domain, pixel_weight=None, image_weight=None):
        """one batch of one domain; returns the device tensor [total, dice, ce, entropy, class dice...]"""
        if not self.external_lr:
            self.opt.param_groups[0]['lr'] = self._lr()
        self.reducer.begin(self.gflat)
        out = self._fwd_bwd(x, label, domain, pixel_weight, image_weight, self._one, self.gflat, self.reducer.ready)
        self.reducer.finish([domain])
        self.opt.step_flat(self.gflat, [domain])
        self.iteration += 1
        return out

    def step_all(self, batches):
        """batches: one dict per domain with 'image', 'label_prob' (+ 'pixel_weight', 'image_weight').

        Data parallel (agent_seg.py:459-495 + 692-698): the gradient exchange overlaps the LAST domain's backward.
        Domain 0 leaves its gradients in gflat, a later domain writes into gacc; a non-last domain's BatchNorm segment is
        final when its pass ends and travels during the next pass; during the last domain's backward every completed bucket
        is folded (gflat[s:e] += gacc[s:e]) and all-reduced from the weight-gradient stream at once (GradAllReducer.ready),
        so only the last bucket and the last domain's BatchNorm segment are exposed behind the last kernel.  The sums are
        the same additions as the single-rank path's (one add per element: domain 0 + domain 1), only cut into ranges."""
        nd = len(batches)
        outs = []
        red = self.reducer
        overlap = red.enabled and self.overlap_all
        if overlap and nd > 2:
            # reduce_domain(k) all-reduces gflat's domain-k BatchNorm range asynchronously while the next non-last domain's
            # `gflat.add_(gacc)` would read-modify-write the whole buffer on the main stream, and loss = (l0 + l1) / 2 fixes the
            # 1 / 2 weights: this path is the reference's two-domain iteration (agent_seg.py:459-495), nothing more
            raise ValueError("fplx: TrainStep.step_all overlaps the gradient exchange for at most two domains (got %d); set "
                             "overlap_all = False for more" % nd)
        if self.gacc is None and nd > 1:
            self.gacc = torch.zeros_like(self.gflat)
        red.begin(self.gflat)
        for k, b in enumerate(batches):
            # loss = (l0 + l1) / 2  (agent_seg.py:482): both terms carry 1/2; a single domain carries 1
            gs = self._one if nd == 1 else self._half
            last = k == nd - 1
            tgt = self.gflat if k == 0 else self.gacc
            hook = None
            if overlap and last:
                red.set_acc(None if k == 0 else self.gacc)
                hook = red.ready
            # the parameters do not change between the domains of one iteration: the first forward's weight packs serve all
            outs.append(self._fwd_bwd(b['image'], b['label_prob'], k, b.get('pixel_weight'), b.get('image_weight'),
                                      gs, tgt, hook, reuse_packs=k > 0))
            if overlap and last:
                break
            if k > 0:
                self.gflat.add_(self.gacc)
            if overlap:
                red.reduce_domain(k)               # final now: travels beside the next domain's forward + backward
        if overlap:
            red.finish([nd - 1])                   # remaining buckets + the last domain's BatchNorm segment (folded from gacc)
            red.set_acc(None)
        else:
            red.finish(list(range(nd)))
        if not self.external_lr:
            self.opt.param_groups[0]['lr'] = self._lr()
        self.opt.step_flat(self.gflat, list(range(nd)))
        self.iteration += 1
        return outs

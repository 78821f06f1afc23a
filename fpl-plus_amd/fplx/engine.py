"""Forward / backward schedule of the DSBN 3D U-Net over the fplx C ABI.

This is the host-side "graph": a fixed sequence of kernel launches on the current HIP stream
(no tracing compiler, no autograd inside).  Activations are NDHWC in `act_dtype`; the skip /
up-sample concat of every UpBlock is one pre-allocated [voxels, 2*C] buffer that the encoder
and the transposed convolution write into directly (reference torch.cat,
PyMIC/pymic/net/net3d/unet2d5_dsbn.py:182, is never executed).

Reference semantics implemented here (paths under /root/reference/PyMIC/pymic):
  UNet2D5_dsbn.forward            net/net3d/unet2d5_dsbn.py:296-309
  ConvBlockND.forward             net/net3d/unet2d5_dsbn.py:66-81 (both branches: a dim-2 level runs its 3x3 convolutions
                                  as 3x3x3 kernels with the taps in the middle depth plane, MaxPool2d / ConvTranspose2d
                                  per depth slice; depth is halved only by dim-3 levels)
  DownBlock / UpBlock             net/net3d/unet2d5_dsbn.py:108-129 / 156-188
  DomainSpecificBatchNorm3d       net_run_dsbn/dsbn.py:54-57 (bns[domain_label[0]] for the batch)
"""

import torch
from . import ops, _lib
from ._lib import F32, BF16

LEVEL_OF_BLOCK = [0, 1, 2, 3, 4, 3, 2, 1, 0]      # block0..4, up1..up4


class Saved(object):
    """what one forward leaves behind for its backward"""
    __slots__ = ("x", "dims", "domain", "train", "seed", "step", "blocks", "cats", "pooled", "packs",
                 "drop_on", "deconv_in", "skips", "split", "oc_fused", "oc_wg", "pack_gen")


class Engine(object):
    def __init__(self, net):
        self.net = net                     # fplx.net.UNet2D5_dsbn (parameter container)
        self.ws = None
        self.ws_side = None
        self._pack_cache = None
        self._train_packs = None                         # packs of the last train-mode forward (reuse_packs)
        # persistent bf16 packs of the 3x3x3 layers the fused Adam + pack launch takes (fplx_adam_pack_step, round 5):
        # (act dtype, device, {site: (wf, wb)}, {site: stamp}); _adam_packs = the _pack_key for which their CONTENT is current
        # (the optimiser step wrote it) - any other parameter write the host can see drops it (invalidate(), a version-counter
        # change) and everything is packed again.  Writers the host cannot see (p.data.mul_(), an EMA swap through `.data`,
        # init.*_(w.data), raw pointers: no version counter moves) are caught on the device: every pack tile carries a stamp of
        # the fp32 master values it was made from and a train-mode forward verifies them in one cheap launch, repacking exactly
        # the tiles whose masters changed (round 6; csrc/conv_generic.hip: pack27_tile)
        self._pack_bufs = None
        self._adam_packs = None
        # generation of the persistent packs' CONTENT: bumped by every host-visible overwrite (the optimiser launch, a full
        # repack).  A saved forward aliases those buffers (sv.packs): its backward refuses to run on another generation's
        # weights (forward A, optimiser step, backward A would compute A's data gradients with the NEW weights - ADVICE r05)
        self._pack_gen = 0
        self.use_adam_pack = _lib.host_knob("adam_pack") != 0
        self.use_pack_multi = _lib.host_knob("pack_small_multi") != 0       # transposed-convolution + out_conv packs in one launch
        self.allow_pack_reuse = _lib.host_knob("pack_reuse") != 0      # A/B switches: fplx/_lib.py:_HOST_KNOBS
        self._side = None                  # second HIP stream: weight gradients run beside the data-gradient chain
        # side_stream = 0 serialises all kernels on one stream (clean per-kernel profiles)
        self.use_side_stream = _lib.host_knob("side_stream") != 0
        self.use_split_cat = _lib.host_knob("split_cat") != 0
        self.use_fused_pool = _lib.host_knob("fused_pool") != 0
        # inference (eval-mode BatchNorm, nothing kept for a backward, bf16): BatchNorm folded into the packs, PReLU in the
        # convolution's write-out - the BN-apply passes of the sites without active dropout disappear (FPLX_EVAL_FUSE=0: off)
        self.use_eval_fusion = _lib.host_knob("eval_fuse") != 0
        self.stem_wgrad_on_main = _lib.host_knob("stem_wg_main") != 0
        # out_conv fused with the BatchNorm + PReLU passes of the site in front of it (fplx_outconv_fwd_bn / _dgrad_bn_*):
        # one pass over that site's tensor forward, two instead of three (+ the data gradient's write) backward
        self.use_outconv_fusion = _lib.host_knob("outconv_fuse") != 0
        # the stem site's backward: its dy has one consumer, the stem's weight gradient, which forms it from y and d(a) itself
        # (fplx_stem_wgrad_bn) - the apply pass of that site's BatchNorm backward (write dy, read it back) disappears
        self.use_stem_wgrad_bn = _lib.host_knob("stem_wgrad_bn") != 0
        # out_conv's weight gradient takes the last site's PRE-BatchNorm tensor too (fplx_outconv_wgrad_bn): with the fused
        # forward / backward above that site's activation has no reader left and is neither allocated nor written
        self.use_outconv_wgrad_bn = _lib.host_knob("outconv_wgrad_bn") != 0
        self._fold_cache = {}              # (act dtype, domain) -> {site key: (folded forward pack, folded bias)}
        # TIMING PROBE ONLY (tools/step_ab.py "@defer_probe=1"): the decoder's weight gradients of a step are
        # not launched in backward but beside the NEXT step's forward - their results are discarded by that step's gradient
        # zeroing, so training is wrong; what is measured is the step time a pipelined optimiser step could reach
        self.defer_probe = False
        self._deferred = []
        # block_joins: the main stream waits for the weight-gradient stream at every block boundary of backward.
        # None = decide per network (see backward()); a bisection tool may set it (round 2's tools/race25.py, git history).
        self.block_joins = None
        self.debug_tap = None              # callable(name, tensor) on intermediate gradients of backward (bisection tools)

    # ------------------------------------------------------------------ helpers
    def _workspace(self, nbytes, dev):
        if self.ws is None or self.ws.numel() < nbytes or self.ws.device != dev:
            self.ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            self.ws_side = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        return self.ws

    def _persistent_packs(self, act_dtype):
        """-> {site: (wf, wb)}: the pack buffers the optimiser step writes in place (bf16, the layers fplx_adam_pack_ok takes);
        empty when the fused step is off"""
        net = self.net
        if not self.use_adam_pack or act_dtype != torch.bfloat16:
            return {}
        dev = net.flat_params.device
        if self._pack_bufs is None or self._pack_bufs[0] != act_dtype or self._pack_bufs[1] != dev:
            first = next(iter(net.conv_sites()))[0]
            bufs, stamps = {}, {}
            for name, conv in net.conv_sites():
                w = conv.weight
                if w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3) and name != first and ops.adam_pack_ok(w.shape[0], w.shape[1]):
                    bufs[name] = (torch.empty((27, w.shape[0], w.shape[1]), dtype=act_dtype, device=dev),
                                  torch.empty((27, w.shape[1], w.shape[0]), dtype=act_dtype, device=dev))
                    stamps[name] = torch.empty(ops.pack_stamp_floats(w.shape[0], w.shape[1]), dtype=torch.float32, device=dev)
            self._pack_bufs = (act_dtype, dev, bufs, stamps)
            self._adam_packs = None
        return self._pack_bufs[2]

    def adam_pack_plan(self):
        """for FusedAdam.step_flat: [(element offset in the flat buffer, cout, cin, wf, wb, stamp)] of the layers whose packs the
        optimiser launch writes (ascending offsets, all inside the shared segment), or None"""
        net = self.net
        bufs = self._persistent_packs(net.act_dtype)
        if not bufs:
            return None
        stamps = self._pack_bufs[3]
        plan = []
        for name, (wf, wb) in bufs.items():
            o, n, shp = net._layout[name + ".weight"]
            plan.append((o, shp[0], shp[1], wf, wb, stamps[name]))
        plan.sort(key=lambda t: t[0])
        return plan

    def packs_written_by_optimizer(self):
        """the optimiser step has just written the persistent packs from the updated weights (after its invalidate())"""
        self._adam_packs = self._pack_key(self.net.act_dtype)
        self._pack_gen += 1

    def _pack(self, act_dtype):
        net = self.net
        packs = {}
        first = next(iter(net.conv_sites()))[0]
        batch, kept = [], []                             # the 3x3x3 layers: one launch for all of them
        bufs = self._persistent_packs(act_dtype)
        stamps = self._pack_bufs[3] if bufs else {}
        fresh = bool(bufs) and self._adam_packs is not None and self._adam_packs == self._pack_key(act_dtype)
        for name, conv in net.conv_sites():
            want_wb = name != first                      # no data gradient w.r.t. the network input
            if conv.weight.dim() == 4:                   # Conv2d of a 2.5D level
                packs[name] = ops.pack_conv2d_weight(conv.weight, act_dtype, want_wb)
            elif fresh and name in bufs:                 # written by the optimiser step itself (or by the previous forward)
                kept.append((name, conv.weight, want_wb, bufs[name], stamps[name]))
            else:
                batch.append((name, conv.weight, want_wb, bufs.get(name), stamps.get(name)))
        # kept packs: ONE launch checks every tile's stamp against the master weights as they are now and repacks the tiles that
        # differ - nothing, normally (a few microseconds); everything a `.data` writer touched otherwise
        for todo, verify in ((kept, True), (batch, False)):
            for i in range(0, len(todo), 32):
                part = todo[i:i + 32]
                res = ops.pack_conv_weights_batched([b[1] for b in part], act_dtype, [b[2] for b in part], [b[3] for b in part],
                                                    [b[4] for b in part], verify)
                for b, r in zip(part, res):
                    packs[b[0]] = r
        if bufs and not fresh:
            self._adam_packs = self._pack_key(act_dtype)     # the persistent buffers now hold the current weights' packs
            self._pack_gen += 1
        oc = net.out_conv
        if self.use_pack_multi and not net.bilinear:
            # the remaining small packs (transposed convolutions, out_conv) in ONE launch instead of six
            dev = net.flat_params.device
            jobs = []
            for name, tr in net.deconv_sites():
                w = tr.weight
                ci, co, taps = w.shape[0], w.shape[1], (8 if w.dim() == 5 else 4)
                wf = torch.empty((taps, co, ci), dtype=act_dtype, device=dev)
                wb = torch.empty((taps, ci, co), dtype=act_dtype, device=dev)
                packs[name] = (wf, wb)
                jobs.append((1, w, wf, wb, ci, co, taps))
            ncls, c0 = oc.weight.shape[0], oc.weight.shape[1]
            owf = torch.empty((9, ncls, c0), dtype=torch.float32, device=dev)      # fp32 planar logits
            owb = torch.empty((9, c0, ncls), dtype=act_dtype, device=dev)
            jobs.append((0, oc.weight, owf, None, ncls, c0, 9))
            jobs.append((0, oc.weight, None, owb, ncls, c0, 9))
            ops.pack_weights_multi(jobs)
            packs["out_conv"] = (owf, owb)
            return packs
        for name, tr in net.deconv_sites():
            if net.bilinear:                             # kernel-1 convolution in front of the (tri / bi)linear upsampling
                w5 = tr.weight.reshape(tr.weight.shape[0], tr.weight.shape[1], 1, 1, 1)
                packs[name] = ops.pack_conv_weight(w5, act_dtype, True)
            else:
                packs[name] = ops.pack_deconv_weight(tr.weight, act_dtype)
        wf, _ = ops.pack_conv_weight(oc.weight, torch.float32, False)       # fp32 planar logits
        _, wb = ops.pack_conv_weight(oc.weight, act_dtype, True)
        packs["out_conv"] = (wf, wb)
        return packs

    def default_block_joins(self):
        """No joins at block boundaries: backward's main stream never waits for the weight-gradient stream until the end.
        (Round 1 needed them for networks with 2D levels; the cause was a store-data hazard inside the march kernels'
        inline-asm 16-byte stores, fixed there - DESIGN section 7, profiles/r02_race25_hazard_location.txt.)"""
        return False

    _tuning_epoch = -1

    def _pack_key(self, adt):
        net = self.net
        return (adt, net.flat_params.data_ptr(), net.flat_params._version) + tuple(p._version for p in net._named.values())

    def invalidate(self):
        self._pack_cache = None
        self._train_packs = None
        self._fold_cache = {}
        self._adam_packs = None

    def _folded(self, adt, domain, key, conv, bn):
        """forward pack and bias of a convolution with its eval-mode BatchNorm folded in (dsbn.py:54-57 on running statistics:
        z = scale (conv(x; w) + b) + shift = conv(x; scale w) + (scale b + shift)); cached until the parameters or the
        running statistics change: invalidate(), any train-mode forward, an optimiser step, or a version-counter change of
        one of the six source tensors"""
        ck = (adt, domain)
        tab = self._fold_cache.setdefault(ck, {})
        bnm = bn.bns[domain]
        # the fold bakes in six tensors; torch-side in-place edits made while the net stays in eval mode (running_mean.copy_, an
        # EMA / SWA weight swap, a stock torch optimiser) bump their version counters - a stale fold would give silently wrong
        # logits.  (Writes through raw pointers - the engine's own Adam, the train-mode statistics kernels - do not; those
        # paths call invalidate(): FusedAdam.step_flat / step, every train-mode forward.)
        src = (conv.weight, conv.bias, bnm.weight, bnm.bias, bnm.running_mean, bnm.running_var)
        ver = tuple(-1 if t is None else (t.data_ptr(), t._version) for t in src)
        if key not in tab or tab[key][2] != ver:
            with torch.no_grad():
                scale = bnm.weight.detach().float() * torch.rsqrt(bnm.running_var.float() + bnm.eps)
                shift = bnm.bias.detach().float() - bnm.running_mean.float() * scale
                w = conv.weight.detach().float()
                wf_ = (w * scale.view(-1, *([1] * (w.dim() - 1)))).contiguous()
                b = conv.bias.detach().float() if conv.bias is not None else torch.zeros_like(scale)
                biasf = (b * scale + shift).contiguous()
                if w.dim() == 4:
                    wf, _ = ops.pack_conv2d_weight(wf_, adt, False)
                else:
                    wf, _ = ops.pack_conv_weight(wf_, adt, False)
            tab[key] = (wf, biasf, ver)
        return tab[key][:2]

    # ------------------------------------------------------------------ forward
    def forward(self, x, domain, train, drop_on, seed=0, step=0, keep=True, mc=1, out=None, reuse_packs=False):
        """x: fp32 [N, Cin, D, H, W] contiguous on the GPU -> logits fp32 [N, class_num, D, H, W].
        train: BatchNorm uses batch statistics (and updates the running ones);
        drop_on: list of 9 bools - dropout active per ConvBlockND.
        mc > 1 (inference only: eval-mode BatchNorm, nothing kept): `mc` Monte-Carlo passes of test-time dropout over the SAME
        input in one call -> logits [mc * N, ...], pass-major.  Equal to forward(x.repeat(mc, 1, 1, 1, 1)) - the dropout
        masks are keyed by the element index of the mc * N batch - but the encoder levels above the first active dropout see
        the same input in every pass and are computed ONCE (the shipped configs drop out at levels 2-4 only: levels 0 and 1,
        the two most expensive, run once instead of `mc` times; reference: agent_seg.py:898-909 runs the whole net per pass).
        out: where the logits go (fp32 [mc * N, class_num, D, H, W], contiguous) instead of a new tensor.
        reuse_packs: the caller guarantees that no parameter changed since the previous train-mode forward (the second domain
        of one training_all iteration: both losses are formed before the one optimiser step, agent_seg.py:462-486) - the
        weight packs of that forward are used again instead of being rebuilt."""
        net = self.net
        logits_out = out
        ops.require_gpu(x)
        if x.dim() != 5:
            raise ValueError('expected 5D input (got {}D input)'.format(x.dim()))   # dsbn.py:61-64
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        N, Cin, D, H, W = x.shape
        if Cin != net.in_chns:
            raise ValueError("fplx: input has {0:} channels, network expects {1:}".format(Cin, net.in_chns))
        pds = [2 if net.dims[l] == 3 else 1 for l in range(4)]          # depth factor of the pooling after level l
        dfac = pds[0] * pds[1] * pds[2] * pds[3]
        if (D % dfac) or (H % 16) or (W % 16):
            raise ValueError("fplx: H, W must be multiples of 16 and D of %d (four 2x poolings, depth only at the "
                             "3D levels), got %dx%dx%d" % (dfac, D, H, W))
        if domain < 0 or domain >= net.num_domains:
            raise IndexError("fplx: domain_label %d out of range" % domain)
        dev, adt = x.device, net.act_dtype
        a_dt = ops._DT[adt]
        ft = net.ft_chns
        if self._tuning_epoch != _lib.tuning_epoch:      # a kernel knob was flipped: packs / folds may have been laid out for another kernel
            self.invalidate()
            self._pack_bufs = None                       # (which layers have a tiled pack and stamps is knob-dependent too)
            self._tuning_epoch = _lib.tuning_epoch
        if train and reuse_packs and self.allow_pack_reuse and self._train_packs is not None and self._train_packs[0] == adt:
            packs = self._train_packs[1]
        elif train or self._pack_cache is None or self._pack_cache[0] != self._pack_key(adt):
            packs = self._pack(adt)
            # eval-mode packs are kept until the parameters change: invalidate() (the engine's own raw-pointer writers call it)
            # or a version-counter change of ANY parameter (torch-side in-place edits under no_grad: p.mul_(), p.copy_() - each
            # parameter is attached as p.data = flat[...] and owns its counter; the flat buffer's own does not move then).
            # Edits through `.data` (p.data.add_()) or raw pointers bump no counter: such writers call engine.invalidate().
            self._pack_cache = None if train else (self._pack_key(adt), packs)
            self._train_packs = (adt, packs) if train else None
        else:
            packs = self._pack_cache[1]
        if train:
            self._fold_cache = {}                        # the running statistics are about to change
        deferred_join = None
        if train and self._deferred and self._side is not None:
            # timing probe (defer_probe): the previous step's decoder weight gradients run beside this forward
            main_ = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record(main_)
            self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                for fn_, _ in self._deferred:
                    fn_()
                deferred_join = torch.cuda.Event()
                deferred_join.record(self._side)
            self._deferred_keep = self._deferred
            self._deferred = []
        fuse = self.use_eval_fusion and not train and not keep and adt == torch.bfloat16
        # the last site's BatchNorm + PReLU inside the out_conv kernel (not where that site runs the folded inference form)
        oc_fuse = (self.use_outconv_fusion and not fuse and mc == 1 and adt == torch.bfloat16 and
                   ops.outconv_bn_ok((N, D, H, W), net.ft_chns[0], net.n_class))
        oc_wg = (oc_fuse and self.use_outconv_wgrad_bn and
                 ops.outconv_wgrad_bn_ws_bytes((N, D, H, W), net.ft_chns[0], net.n_class) > 0)
        if mc > 1 and (train or keep):
            raise ValueError("fplx: Monte-Carlo replication (mc > 1) is an inference mode: eval-mode BatchNorm, keep=False")
        # first encoder level whose input differs between Monte-Carlo passes (the level after the first active dropout)
        rep = 5
        if mc > 1:
            act = [b for b in range(5) if drop_on[b] and net.block_modules[b].dropout_p > 0]
            rep = act[0] if act else 5
            if rep == 5:                                 # no dropout is active: every pass is the same forward
                logits, _ = self.forward(x, domain, train, drop_on, seed, step, keep, 1)
                if logits_out is None:
                    return logits.repeat(mc, 1, 1, 1, 1), None
                logits_out.view((mc,) + tuple(logits.shape))[:] = logits
                return logits_out, None
        NM = N * mc
        dims = [(NM, D, H, W)]                           # decoder view: every level holds all passes
        for l in range(4):
            dims.append((NM, dims[l][1] // pds[l], dims[l][2] // 2, dims[l][3] // 2))
        vox = [n * d * h * w for (n, d, h, w) in dims]
        # encoder view: levels < rep hold ONE copy.  (The first conv site of block `rep` is still shared - its dropout is
        # what makes the passes differ - but replicating its input keeps the element indices the masks are keyed by.)
        edims = [((N if l < rep else NM),) + dims[l][1:] for l in range(5)]
        evox = [n * d * h * w for (n, d, h, w) in edims]

        sv = Saved()
        sv.x, sv.dims, sv.domain, sv.train, sv.seed, sv.step = x, dims, domain, train, seed, step
        sv.blocks, sv.cats, sv.pooled, sv.packs, sv.drop_on, sv.deconv_in = [], [], [], packs, list(drop_on), []
        sv.pack_gen = self._pack_gen

        def empty(v, c):
            return torch.empty((v, c), dtype=adt, device=dev)

        # levels 0..3: skip and up halves of the decoder input.  Normally two channel slices of ONE [V, 2C] buffer (the
        # reference's torch.cat, never copied); where the halves are only 32 channels (64 bytes) wide every kernel that
        # touches one half would move half-used 128-byte lines, so there they are two tensors and the first decoder
        # convolution takes both (fplx_conv3d_*_cat2 / _split2)
        # (a backward through eval-mode BatchNorm also needs the conv bias gradient: that case keeps the buffer)
        can_split = self.use_split_cat and adt == torch.bfloat16 and (train or not keep)
        split = [can_split and ops.conv3d_cat2_ok(dims[l], 2 * ft[l], ft[l]) for l in range(4)]
        if fuse:
            # inference: also wherever the fused kernel has a two-tensor form and runs (no active dropout at that site) - the
            # Monte-Carlo passes then share ONE copy of the level's skip tensor (skip_mod below)
            for l in range(4):
                blk = net.block_modules[8 - l]
                if (can_split and not split[l] and not (drop_on[8 - l] and blk.dropout_p > 0) and
                        ops.conv3d_fwd_act_ok(dims[l], 2 * ft[l], ft[l], blk.conv_of(1).weight.dim() == 4, True)):
                    split[l] = True
        cats, skips, ups = [], [], []
        for l in range(4):
            if split[l]:
                cats.append(None)
                skips.append(empty(vox[l], ft[l]))
                ups.append(empty(vox[l], ft[l]))
            else:
                cats.append(empty(vox[l], 2 * ft[l]))
                skips.append(cats[l][:, :ft[l]])
                ups.append(cats[l][:, ft[l]:])
        sv.cats, sv.skips, sv.split = cats, skips, split
        # shared encoder levels write their skip once; it is copied into the decoder's (all-passes) buffer afterwards
        eskips = [empty(evox[l], ft[l]) if evox[l] != vox[l] else skips[l] for l in range(4)]

        # Monte-Carlo passes over a shared encoder: where the decoder's first convolution of a level is the fused two-tensor
        # kernel, it reads the ONE copy of the skip tensor modulo the batch (n_x0) instead of a copy per pass
        skip_mod = [0, 0, 0, 0]
        if fuse and mc > 1:
            for l in range(4):
                blk = net.block_modules[8 - l]
                p_on = blk.dropout_p if drop_on[8 - l] else 0.0
                if (l < rep and split[l] and p_on == 0.0 and
                        ops.conv3d_fwd_act_ok(dims[l], 2 * ft[l], ft[l], blk.conv_of(1).weight.dim() == 4, True)):
                    skip_mod[l] = edims[l][0]

        def conv_site(xin, xs, x_dt, cin, key, site, l, out_view, p, sid, dropout_active, pool=None, dims=dims, vox=vox, n_x0=0,
                      defer_act=False):
            """conv3x3x3 (+stats) -> DSBN finalize -> BN-apply + PReLU (+dropout) into out_view; pool = (pooled, pd): the
            MaxPool of out_view is produced by the same pass (tail of a DownBlock)"""
            conv, bn, prelu = site
            cout = conv.weight.shape[0]
            mid = conv.weight.dim() == 4           # Conv2d of a 2.5D level: its pack lives in the middle depth plane
            pp = p if dropout_active else 0.0
            cat2 = isinstance(xin, tuple)
            if (fuse and pp == 0.0 and (cat2 or x_dt == a_dt) and ops.conv3d_fwd_act_ok(dims[l], cin, cout, mid, cat2) and
                    all(ops.ld_of(t) % 8 == 0 and t.data_ptr() % 16 == 0 for t in ((xin if cat2 else (xin,)) + (out_view,)))):
                # inference: conv + folded BatchNorm + PReLU in one kernel, straight into the site's output
                wf, biasf = self._folded(adt, domain, key, conv, bn)
                ops.conv3d_fwd_act(xin[0] if cat2 else xin, xin[1] if cat2 else None, wf, biasf, prelu.weight, out_view, dims[l],
                                   cin, cout, mid, n_x0)
                if pool is not None:
                    ops.maxpool2_fwd(out_view, pool[0], dims[l], cout, pool[1])
                return None, None, pp
            if n_x0:
                raise RuntimeError("fplx: the shared skip tensor of level %d was planned for the fused kernel" % l)
            y = empty(vox[l], cout)
            bnbuf = torch.empty((4, cout), dtype=torch.float32, device=dev)
            bnm = bn.bns[domain]
            if train:
                rows = ops.conv3d_stats_rows(dims[l], cin, cout, (3, 3, 3), x_dt, a_dt, mid)
                stats = torch.empty((rows, 2, cout), dtype=torch.float32, device=dev)
            else:
                rows, stats = 0, None
            if isinstance(xin, tuple):
                ops.conv3d_fwd_cat2(xin[0], xin[1], packs[key][0], conv.bias, y, dims[l], cin, cout, stats, mid)
            else:
                ops.conv3d_fwd(xin, xs, x_dt, packs[key][0], conv.bias, y, ops.cl_strides(*dims[l][1:], cout), a_dt,
                               dims[l], cin, cout, (3, 3, 3), stats, mid=mid)
            if train:
                ops.bn_train_finalize(stats, rows, cout, vox[l], bnm.weight, bnm.bias, bnm.running_mean,
                                      bnm.running_var, bnm.num_batches_tracked, bnbuf, bnm.momentum, bnm.eps)
            else:
                ops.bn_eval_prepare(bnm.weight, bnm.bias, bnm.running_mean, bnm.running_var, bnbuf, bnm.eps)
            if pool is not None:
                ops.bn_act_pool_fwd(y, out_view, pool[0], bnbuf, prelu.weight, dims[l], cout, pool[1])
            elif not defer_act:                          # defer_act: the out_conv kernel applies BatchNorm + PReLU (oc_fuse)
                ops.bn_act_fwd(y, out_view, bnbuf, prelu.weight, pp, seed, sid, cout)
            return y, bnbuf, pp

        def conv_block(b, xin, xs, x_dt, cin, l, out_view, pool=None, dims=dims, vox=vox, n_x0=0):
            blk = net.block_modules[b]
            key = net.block_keys[b]
            c = ft[l]
            a1 = empty(vox[l], c)
            sid = step * 16 + b
            y1, bn1, p1 = conv_site(xin, xs, x_dt, cin, key + "." + blk.cname(1),
                                    (blk.conv_of(1), blk.bn_of(1), blk.relu_1), l, a1, blk.dropout_p, sid, drop_on[b],
                                    dims=dims, vox=vox, n_x0=n_x0)
            y2, bn2, _ = conv_site(a1, ops.cl_strides(*dims[l][1:], c), a_dt, c, key + "." + blk.cname(2),
                                   (blk.conv_of(2), blk.bn_of(2), blk.relu_2), l, out_view, 0.0, 0, False, pool,
                                   dims=dims, vox=vox, defer_act=(b == 8 and oc_fuse))
            sv.blocks.append(dict(xin=xin, xs=xs, x_dt=x_dt, cin=cin, l=l, y1=y1, bn1=bn1, p1=p1, sid=sid, a1=a1,
                                  y2=y2, bn2=bn2, out=out_view))

        # ---- encoder
        cur, cur_s, cur_dt, cur_c = x, ops.planar_strides(Cin, D, H, W), F32, Cin
        if rep == 0 and mc > 1:
            cur = x.repeat(mc, 1, 1, 1, 1)
        for i in range(5):
            out_view = eskips[i] if i < 4 else empty(evox[4], ft[4])
            fused = i < 4 and self.use_fused_pool and ops.bn_pool_fused_ok(ft[i], adt)
            pooled = empty(evox[i] // (pds[i] * 4), ft[i]) if i < 4 else None
            conv_block(i, cur, cur_s, cur_dt, cur_c, i, out_view, (pooled, pds[i]) if fused else None, dims=edims, vox=evox)
            if i < 4:
                if not fused:
                    ops.maxpool2_fwd(out_view, pooled, edims[i], ft[i], pds[i])
                if eskips[i] is not skips[i] and not skip_mod[i]:      # one copy -> every pass's slot of the decoder input
                    skips[i].view(mc, evox[i], ft[i])[:] = eskips[i]
                if evox[i + 1] != pooled.shape[0]:     # the next level is the first one that differs between passes
                    pooled = pooled.repeat(mc, 1)
                sv.pooled.append(pooled)
                cur, cur_s, cur_dt, cur_c = pooled, ops.cl_strides(*edims[i + 1][1:], ft[i]), a_dt, ft[i]
            else:
                cur = out_view
        # ---- decoder
        for j in range(4):
            l = 3 - j
            up = net.up_modules[j]
            tr = up.trans()
            sv.deconv_in.append(cur)
            if net.bilinear:                             # unet2d5_dsbn.py:172-176: conv(kernel 1) -> Upsample(align_corners)
                low = empty(vox[l + 1], ft[l])
                ops.conv3d_fwd(cur, ops.cl_strides(*dims[l + 1][1:], ft[l + 1]), a_dt, packs["up%d.%s" % (j + 1, up.tname())][0],
                               tr.bias, low, ops.cl_strides(*dims[l + 1][1:], ft[l]), a_dt, dims[l + 1], ft[l + 1], ft[l],
                               (1, 1, 1), None)
                ops.upsample2_fwd(low, ups[l], dims[l + 1], ft[l], pds[l])
            else:
                ops.deconv2_fwd(cur, packs["up%d.%s" % (j + 1, up.tname())][0], tr.bias, ups[l], dims[l + 1], ft[l + 1],
                                ft[l], pds[l])
            out = None if (j == 3 and oc_wg) else empty(vox[l], ft[l])     # oc_wg: nothing reads the last activation
            xin = ((eskips[l] if skip_mod[l] else skips[l]), ups[l]) if split[l] else cats[l]
            conv_block(5 + j, xin, ops.cl_strides(*dims[l][1:], 2 * ft[l]), a_dt, 2 * ft[l], l, out, n_x0=skip_mod[l])
            cur = out
        # ---- out_conv (1x3x3) -> fp32 planar logits
        ncls = net.n_class
        logits = logits_out
        if logits is None:
            logits = torch.empty((NM, ncls, D, H, W), dtype=torch.float32, device=dev)
        elif (tuple(logits.shape) != (NM, ncls, D, H, W) or logits.dtype != torch.float32 or logits.device != dev
              or not logits.is_contiguous()):
            raise ValueError("fplx: out must be a contiguous fp32 tensor of shape %s on %s" % ((NM, ncls, D, H, W), dev))
        if oc_fuse:
            blk8, mod8 = sv.blocks[8], net.block_modules[8]
            ops.outconv_fwd_bn(blk8["y2"], blk8["bn2"], mod8.relu_2.weight, cur, packs["out_conv"][0], net.out_conv.bias, logits,
                               dims[0], ft[0], ncls)
        else:
            ops.conv3d_fwd(cur, ops.cl_strides(D, H, W, ft[0]), a_dt, packs["out_conv"][0], net.out_conv.bias, logits,
                           ops.planar_strides(ncls, D, H, W), F32, dims[0], ft[0], ncls, (1, 3, 3), None)
        sv.oc_fused = oc_fuse
        sv.oc_wg = oc_wg
        if deferred_join is not None:
            torch.cuda.current_stream().wait_event(deferred_join)
            self._deferred_keep = None
        return logits, (sv if keep else None)

    # ------------------------------------------------------------------ backward
    def backward(self, sv, dlogits, gflat, on_ready=None):
        """dlogits fp32 [N, class_num, D, H, W]; gflat: fp32 flat gradient buffer laid out like
        net.flat_params (zeroed first; BN parameters of the other domains stay zero).
        on_ready(end): optional callback, called whenever the gradients of flat elements [0, end)
        have been enqueued (flat order = production order) - fplx.ddp launches all-reduce buckets."""
        net = self.net
        dims, domain, packs = sv.dims, sv.domain, sv.packs
        if sv.pack_gen != self._pack_gen:
            raise RuntimeError("fplx: backward of a forward whose weight packs have been overwritten since (an optimiser step or a "
                               "repack ran between this forward and its backward): the data gradients would be taken with the new "
                               "weights.  Run backward before the optimiser step.")
        dev, adt = dlogits.device, net.act_dtype
        a_dt = ops._DT[adt]
        ft = net.ft_chns
        N, D, H, W = dims[0]
        ncls = net.n_class
        vox = [n * d * h * w for (n, d, h, w) in dims]
        dlogits = dlogits.contiguous()
        gflat.zero_()        # BN-affine / PReLU gradients are accumulated by their finalize kernels
        gv = net.grad_views(gflat)

        def empty(v, c):
            return torch.empty((v, c), dtype=adt, device=dev)

        # workspace: max over layers
        need = max(ops.conv3d_wgrad_ws_bytes(dims[0], ft[0], ncls, (1, 3, 3)), ops.outconv_wgrad_bn_ws_bytes(dims[0], ft[0], ncls))
        for b in range(9):
            l = LEVEL_OF_BLOCK[b]
            cin1 = sv.blocks[b]["cin"]
            need = max(need, ops.conv3d_wgrad_ws_bytes(dims[l], cin1, ft[l], (3, 3, 3)),
                       ops.conv3d_wgrad_ws_bytes(dims[l], ft[l], ft[l], (3, 3, 3)))
            if net.block_modules[b].dim == 2:
                need = max(need, ops.conv2d_wgrad_ws_bytes(dims[l], cin1, ft[l]), ops.conv2d_wgrad_ws_bytes(dims[l], ft[l], ft[l]))
        pds = [2 if net.dims[l] == 3 else 1 for l in range(4)]
        for j in range(4):
            l = 3 - j
            if net.bilinear:
                need = max(need, ops.conv3d_wgrad_ws_bytes(dims[l + 1], ft[l + 1], ft[l], (1, 1, 1)))
            else:
                need = max(need, ops.deconv2_wgrad_ws_bytes(dims[l + 1], ft[l + 1], ft[l], pds[l]))
        ws = self._workspace(need, dev)
        # Weight-gradient kernels hang off the dependency chain (dgrad -> bn backward -> dgrad ...): they run
        # on a second stream with their own workspace, overlapping the HBM-bound BN/pool passes with MFMA work.
        main = torch.cuda.current_stream()
        side_on = self.use_side_stream
        if side_on and (self._side is None or self._side.device != dev):
            self._side = torch.cuda.Stream(device=dev)
        side = self._side if side_on else None
        keep = []                          # tensors the side stream still reads: kept alive until the join
        block_joins = self.block_joins
        tap = self.debug_tap
        if block_joins is None:
            block_joins = self.default_block_joins()
        ws_w = self.ws_side if side_on else ws

        decoder_phase = [True]

        def on_side(fn, *tensors):
            if self.defer_probe and side_on and decoder_phase[0]:
                self._deferred.append((fn, tensors))
                return
            if not side_on:
                fn()
                return
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            keep.extend(tensors)
            with torch.cuda.stream(side):
                fn()

        def join_side():
            if side_on:
                ev = torch.cuda.Event()
                ev.record(side)
                main.wait_event(ev)

        maxc = max(ft) * 2
        part_floats = ops.num_partials(vox[0]) * (2 * maxc + 1)
        if sv.oc_fused:        # the fused out_conv backward writes one partial row per tile block: up to 2048 rows of 2 ft[0] + 1
            part_floats = max(part_floats, ops.outconv_bn_rows(dims[0], ft[0], ncls) * (2 * ft[0] + 1))
        part = torch.empty(part_floats, dtype=torch.float32, device=dev)
        coef = torch.empty((2, maxc), dtype=torch.float32, device=dev)

        # ---- out_conv
        last = sv.blocks[8]["out"]

        def outconv_wgrad():
            if getattr(sv, "oc_wg", False):            # (the forward stored no activation: it is formed again from y)
                b8 = sv.blocks[8]
                on_side(lambda: ops.outconv_wgrad_bn(b8["y2"], b8["bn2"], net.get_param(net.block_keys[8] + ".relu_2.weight"),
                                                     dlogits, gv["out_conv.weight"], gv["out_conv.bias"], dims[0], ft[0], ncls,
                                                     ws_w), dlogits)
                return
            on_side(lambda: ops.conv3d_wgrad(last, ops.cl_strides(D, H, W, ft[0]), a_dt, dlogits,
                                             ops.planar_strides(ncls, D, H, W), F32, gv["out_conv.weight"],
                                             gv["out_conv.bias"], dims[0], ft[0], ncls, (1, 3, 3), ws_w), dlogits)

        def ready(last_name):
            """block boundary: the gradients of flat elements [0, end of last_name) have been enqueued"""
            if block_joins:
                join_side()
            if on_ready is None:
                return
            o, n, _ = net._layout[last_name]
            end = (o + n + 3) // 4 * 4                 # parameters start on 4-float boundaries (net._ensure_flat)
            # a reducer that would launch nothing here (single rank, bucket not complete) must not cost a join:
            # the main stream would wait for the weight gradients ten times per step
            pending = getattr(getattr(on_ready, "__self__", None), "pending", None)
            if pending is not None and not pending(end):
                return
            if side_on and pending is not None and not block_joins:
                # the bucket's weight gradients were produced on the side stream, its BN / bias gradients on this
                # one: launch the collective FROM the side stream once that has caught up with this stream's
                # position - the communication stream then depends on both and the data-gradient chain never waits
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    on_ready(end)
                return
            join_side()                # generic callback: make everything visible on the current stream first
            on_ready(end)

        d_cur = empty(vox[0], ft[0])
        oc_fused = bool(sv.oc_fused)
        if not oc_fused:       # (fused: out_conv's data gradient is never stored - site_bwd of the last site recomputes it twice)
            ops.conv3d_fwd(dlogits, ops.planar_strides(ncls, D, H, W), F32, packs["out_conv"][1], None, d_cur,
                           ops.cl_strides(D, H, W, ft[0]), a_dt, dims[0], ncls, ft[0], (1, 3, 3), None)
        # the weight gradient BEHIND the data gradient, as at every 3x3x3 site (either order measures the same step time,
        # 9.64-9.78 ms on one box: the two queues are scheduled dynamically; one convention is kept)
        outconv_wgrad()
        ready("out_conv.bias")

        def site_bwd(key, bnkey, relukey, y, bnbuf, p, sid, d_out, xin, xs, x_dt, cin, l, want_dx, dx_view, reduced=False,
                     wgrad_here=False, from_logits=False):
            """backward of conv -> DSBN -> PReLU -> dropout.  d_out is overwritten with dy.
            reduced: the producer of d_out already wrote the BatchNorm reduction's partial rows (pool_bwd_bn_reduce)"""
            c = ft[l]
            gkey = "%s.bns.%d" % (bnkey, domain)
            if from_logits:     # d_out is not given: it is out_conv's data gradient, recomputed inside the two BatchNorm passes
                ops.outconv_dgrad_bn_bwd(dlogits, packs["out_conv"][1], y, bnbuf, net.get_param(relukey + ".weight"), sv.train,
                                         gv[gkey + ".weight"], gv[gkey + ".bias"], gv[relukey + ".weight"], part, coef, d_out,
                                         dims[l], c, ncls)
            else:
                # (the stem site: no data gradient, so dy's only consumer is the weight gradient - which forms it itself)
                stem_fused = (self.use_stem_wgrad_bn and not want_dx and tap is None and sv.train and x_dt == F32 and p == 0.0 and
                              gv[key + ".weight"].dim() == 5 and adt == torch.bfloat16 and not isinstance(xin, tuple) and
                              ops.ld_of(d_out) % 8 == 0 and d_out.data_ptr() % 16 == 0 and ops.stem_wgrad_bn_ok(dims[l], cin, c))
                ops.bn_act_bwd(y, d_out, d_out, bnbuf, net.get_param(relukey + ".weight"), p, sv.seed, sid, c, sv.train,
                               gv[gkey + ".weight"], gv[gkey + ".bias"], gv[relukey + ".weight"], part, coef, reduced,
                               apply=not stem_fused)
                if stem_fused:
                    fn = lambda: ops.stem_wgrad_bn(xin, y, d_out, bnbuf, net.get_param(relukey + ".weight"), coef,
                                                   gv[key + ".weight"], dims[l], cin, c, ws if wgrad_here else ws_w)
                    if wgrad_here:
                        fn()
                    else:
                        on_side(fn, d_out, xin, y, coef)
                    return
            if tap is not None:
                tap(key + ".dy", d_out)
            # conv bias followed by train-mode BatchNorm: d/d bias == sum of dy == 0 exactly
            db = None if sv.train else gv[key + ".bias"]
            gw = gv[key + ".weight"]
            two_d = gw.dim() == 4                      # Conv2d of a 2.5D level: only the middle-plane taps exist
            if isinstance(xin, tuple):                 # cat([skip, up]) as two tensors (train-mode BN only: db is None)
                if want_dx:
                    ops.conv3d_dgrad_split2(d_out, packs[key][1], dx_view[0], dx_view[1], dims[l], cin, c, two_d)
                    if tap is not None:
                        tap(key + ".dx0", dx_view[0])
                        tap(key + ".dx1", dx_view[1])
                on_side(lambda: ops.conv3d_wgrad_cat2(xin[0], xin[1], d_out, gw, dims[l], cin, c, ws_w, two_d),
                        d_out, xin[0], xin[1])
                return
            def wg():
                if wgrad_here:
                    # the LAST weight gradient of backward (the stem's): nothing follows it on this stream, while the second
                    # stream is still busy with the previous site's - on this stream the two run side by side instead of one
                    # after the other (the step's tail showed 70 us of stem_wgrad alone on the device, r03 timeline)
                    fn = ops.conv2d_wgrad if two_d else ops.conv3d_wgrad
                    args = (xin, xs, x_dt, d_out, ops.cl_strides(*dims[l][1:], c), a_dt, gw, db, dims[l], cin, c)
                    fn(*(args + ((ws,) if two_d else ((3, 3, 3), ws))))
                elif two_d:
                    on_side(lambda: ops.conv2d_wgrad(xin, xs, x_dt, d_out, ops.cl_strides(*dims[l][1:], c), a_dt, gw, db,
                                                     dims[l], cin, c, ws_w), d_out, xin)
                else:
                    on_side(lambda: ops.conv3d_wgrad(xin, xs, x_dt, d_out, ops.cl_strides(*dims[l][1:], c), a_dt, gw, db,
                                                     dims[l], cin, c, (3, 3, 3), ws_w), d_out, xin)
            if want_dx:
                ops.conv3d_fwd(d_out, ops.cl_strides(*dims[l][1:], c), a_dt, packs[key][1], None, dx_view,
                               ops.cl_strides(*dims[l][1:], ops.ld_of(dx_view)), a_dt, dims[l], c, cin, (3, 3, 3), None,
                               mid=two_d)
                if tap is not None:
                    tap(key + ".dx", dx_view)
            # the weight gradient is enqueued BEHIND the data gradient: the side stream's event then follows the data-gradient
            # kernel, so the two matrix kernels of a site do not start together and split the chip - the weight gradient runs
            # beside the next site's BatchNorm passes (memory-bound) instead (measured -0.4 % on the step)
            wg()

        def block_bwd(b, d_out, want_dx, reduced=False):
            """d_out: gradient w.r.t. the block output [V, C] (overwritten).  Returns d(block input) or None."""
            blk = sv.blocks[b]
            key = net.block_keys[b]
            mod = net.block_modules[b]
            l, c, cin = blk["l"], ft[blk["l"]], blk["cin"]
            d_a1 = empty(vox[l], c)
            site_bwd(key + "." + mod.cname(2), key + "." + mod.bname(2), key + ".relu_2", blk["y2"], blk["bn2"], 0.0, 0,
                     d_out, blk["a1"], ops.cl_strides(*dims[l][1:], c), a_dt, c, l, True, d_a1, reduced,
                     from_logits=(b == 8 and oc_fused))
            if isinstance(blk["xin"], tuple):
                d_in = (empty(vox[l], cin // 2), empty(vox[l], cin // 2)) if want_dx else None
            else:
                d_in = empty(vox[l], cin) if want_dx else None
            site_bwd(key + "." + mod.cname(1), key + "." + mod.bname(1), key + ".relu_1", blk["y1"], blk["bn1"], blk["p1"],
                     blk["sid"], d_a1, blk["xin"], blk["xs"], blk["x_dt"], cin, l, want_dx, d_in,
                     wgrad_here=(b == 0 and side_on and self.stem_wgrad_on_main))
            return d_in

        # ---- decoder, up4 .. up1
        d_skips = [None] * 4
        for j in range(3, -1, -1):
            l = 3 - j
            d_cat = block_bwd(5 + j, d_cur, True)                     # [V_l, 2*ft_l], or the two halves
            if isinstance(d_cat, tuple):
                d_skips[l], d_up = d_cat
            else:
                d_skips[l] = d_cat[:, :ft[l]]
                d_up = d_cat[:, ft[l]:]
            name = "up%d.%s" % (j + 1, net.up_modules[j].tname())
            xin = sv.deconv_in[j]
            if net.bilinear:
                d_low = empty(vox[l + 1], ft[l])
                ops.upsample2_bwd(d_up, d_low, dims[l + 1], ft[l], pds[l])
                lows, highs = ops.cl_strides(*dims[l + 1][1:], ft[l]), ops.cl_strides(*dims[l + 1][1:], ft[l + 1])
                on_side(lambda xin=xin, d_low=d_low, name=name, l=l, lows=lows, highs=highs: ops.conv3d_wgrad(
                    xin, highs, a_dt, d_low, lows, a_dt, gv[name + ".weight"], gv[name + ".bias"], dims[l + 1], ft[l + 1],
                    ft[l], (1, 1, 1), ws_w), d_low, d_up, d_cat, xin)
                ready(name + ".bias")
                d_cur = empty(vox[l + 1], ft[l + 1])
                ops.conv3d_fwd(d_low, lows, a_dt, packs[name][1], None, d_cur, highs, a_dt, dims[l + 1], ft[l], ft[l + 1],
                               (1, 1, 1), None)
                if tap is not None:
                    tap(name + ".dx", d_cur)
                continue
            def deconv_wgrad(xin=xin, d_up=d_up, name=name, l=l, d_cat=d_cat):
                on_side(lambda: ops.deconv2_wgrad(xin, d_up, gv[name + ".weight"], gv[name + ".bias"], dims[l + 1], ft[l + 1],
                                                  ft[l], ws_w, pds[l]), d_up, d_cat, xin)
            d_cur = empty(vox[l + 1], ft[l + 1])
            ops.deconv2_dgrad(d_up, packs[name][1], d_cur, dims[l + 1], ft[l + 1], ft[l], pds[l])
            deconv_wgrad()                         # behind the data gradient, like the out_conv pair above
            ready(name + ".bias")
            if tap is not None:
                tap(name + ".dx", d_cur)
        # ---- encoder, block4 .. block0
        decoder_phase[0] = False
        d_pool = block_bwd(4, d_cur, True)                            # grad w.r.t. pooled3 [V_4, ft_3]
        ready("block4.conv.relu_1.weight")
        for i in range(3, -1, -1):
            d_a2 = empty(vox[i], ft[i])
            fused = self.use_fused_pool and ops.bn_pool_fused_ok(ft[i], adt)
            if fused:      # pooling gradient + skip gradient AND the BatchNorm reduction over the result, in one pass
                blk = sv.blocks[i]
                ops.pool_bwd_bn_reduce(blk["y2"], d_pool, d_skips[i], d_a2, blk["bn2"],
                                       net.get_param(net.block_keys[i] + ".relu_2.weight"), dims[i], ft[i], part, pds[i])
            else:
                ops.maxpool2_bwd(sv.skips[i], d_pool, d_skips[i], d_a2, dims[i], ft[i], pds[i])
            d_pool = block_bwd(i, d_a2, i > 0, fused)
            ready("block%d.conv.relu_1.weight" % i)
        join_side()
        del keep
        return gflat

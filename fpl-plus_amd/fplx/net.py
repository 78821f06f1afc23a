"""UNet2D5_dsbn - drop-in for the reference network class behind PyMIC's SegNetDict
(reference: PyMIC/pymic/net/net3d/unet2d5_dsbn.py:239-309, registry PyMIC/pymic/net/net_dict_seg.py:44).

Same constructor (`params` dict), same call `net(x, domain_label)`, same state_dict key names
for the 3D branch (`block{0-4}.conv.conv3d_{1,2}.*`, `*.bn3d{1,2}.bns.{d}.*`, `*.relu_{1,2}.weight`,
`up{1-4}.trans3d.*`, `up*.conv.*`, `out_conv.*`).  torch.nn modules are used ONLY as parameter
containers (so init, state_dict and `net.apply(...)` hooks such as the reference's test-time
dropout switch, net_run_dsbn/agent_seg.py:843-852, behave identically); all arithmetic runs in
the HIP kernels of libfplx.so through fplx.engine.  There is no CPU path.

Differences, on purpose:
  * conv_dims[l] = 3 (the benchmark configuration) or 2 (the 2.5D levels of the shipped
    configs, config_dual/data_vs/vs_t1s_g.cfg:58 conv_dims = [2, 2, 3, 3, 3]): a dim-2 level holds conv2d_* / bn2d* /
    trans2d members (same state_dict keys and shapes as the reference's) and runs Conv2d / MaxPool2d / ConvTranspose2d
    on every depth slice, which is what the reference's fold of the depth axis into the batch computes
    (unet2d5_dsbn.py:110-127, 160-188).
    The reference also instantiates the twin of the OTHER dimensionality for every layer plus the bilinear branch's
    1x1 convolutions (8.1 M of 30.7 M parameters at 32 base channels, all 3D) - they never receive a gradient and are
    not created.  load_state_dict() accepts reference checkpoints and keeps those keys aside (fplx.checkpoint).
  * params['precision'] = 'fp32' (default, parity mode) | 'bf16' (activations stored as bf16,
    fp32 master weights, fp32 statistics / reductions).
"""
import torch
import torch.nn as nn

from .dsbn import DomainSpecificBatchNorm3d
from .engine import Engine

class ConvBlockND(nn.Module):
    """parameter container of reference ConvBlockND (unet2d5_dsbn.py:48-64): the members of its `dim` only.
    BatchNorm2d and BatchNorm3d hold the same parameters and buffers, so one container class serves both."""

    def __init__(self, in_channels, out_channels, num_domains, dropout_p, dim=3):
        super(ConvBlockND, self).__init__()
        self.dim = dim
        if dim == 3:
            self.conv3d_1 = nn.Conv3d(in_channels, out_channels, kernel_size=3, padding=1)
            self.conv3d_2 = nn.Conv3d(out_channels, out_channels, kernel_size=3, padding=1)
            self.bn3d1 = DomainSpecificBatchNorm3d(out_channels, num_domains=num_domains)
            self.bn3d2 = DomainSpecificBatchNorm3d(out_channels, num_domains=num_domains)
        else:
            self.conv2d_1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1)
            self.conv2d_2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1)
            self.bn2d1 = DomainSpecificBatchNorm3d(out_channels, num_domains=num_domains)
            self.bn2d2 = DomainSpecificBatchNorm3d(out_channels, num_domains=num_domains)
        self.dropout_p = float(dropout_p)
        self.dropout = nn.Dropout(self.dropout_p)
        self.relu_1 = nn.PReLU()
        self.relu_2 = nn.PReLU()

    # uniform access for the engine: member names follow the dimensionality
    def cname(self, i):
        return "conv%dd_%d" % (self.dim, i)

    def bname(self, i):
        return "bn%dd%d" % (self.dim, i)

    def conv_of(self, i):
        return getattr(self, self.cname(i))

    def bn_of(self, i):
        return getattr(self, self.bname(i))


class DownBlock(nn.Module):
    def __init__(self, in_channels, out_channels, num_domains, dropout_p, dim=3):
        super(DownBlock, self).__init__()
        self.dim = dim
        self.conv = ConvBlockND(in_channels, out_channels, num_domains, dropout_p, dim)


class UpBlock(nn.Module):
    """parameter container of reference UpBlock (unet2d5_dsbn.py:131-188): the up-sampling member its configuration uses -
    `trans{2,3}d` (bilinear = False: transposed convolution) or `conv{2,3}d` (bilinear = True: the kernel-1 convolution in
    front of nn.Upsample(scale 2, bi / trilinear, align_corners=True))"""

    def __init__(self, in_channels1, in_channels2, out_channels, num_domains, dropout_p, dim=3, bilinear=False):
        super(UpBlock, self).__init__()
        self.dim = dim
        self.bilinear = bool(bilinear)
        if self.bilinear:
            if dim == 3:
                self.conv3d = nn.Conv3d(in_channels1, in_channels2, kernel_size=1)
            else:
                self.conv2d = nn.Conv2d(in_channels1, in_channels2, kernel_size=1)
        elif dim == 3:
            self.trans3d = nn.ConvTranspose3d(in_channels1, in_channels2, kernel_size=2, stride=2)
        else:
            self.trans2d = nn.ConvTranspose2d(in_channels1, in_channels2, kernel_size=2, stride=2)
        self.conv = ConvBlockND(in_channels2 * 2, out_channels, num_domains, dropout_p, dim)

    def tname(self):
        return ("conv%dd" if self.bilinear else "trans%dd") % self.dim

    def trans(self):
        return getattr(self, self.tname())


class _UNetFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward/backward are fplx.engine schedules."""

    @staticmethod
    def forward(ctx, x, net, domain, train, drop_on, seed, step, keep, *params):
        reuse, net._reuse_packs_once = getattr(net, "_reuse_packs_once", False), False
        logits, sv = net.engine.forward(x, domain, train, drop_on, seed, step, keep=keep, reuse_packs=reuse)
        ctx.net, ctx.sv, ctx.n_params = net, sv, len(params)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        net, sv = ctx.net, ctx.sv
        if sv is None:
            raise RuntimeError("fplx: backward through a forward that ran under no_grad")
        gflat = torch.empty_like(net.flat_params)
        net.engine.backward(sv, dlogits, gflat)
        ctx.sv = None
        gv = net.grad_views(gflat)
        grads = tuple(gv[name] for name in net.active_param_names(sv.domain))
        return (None, None, None, None, None, None, None, None) + grads


class UNet2D5_dsbn(nn.Module):
    def __init__(self, params):
        super(UNet2D5_dsbn, self).__init__()
        self.params = params
        self.in_chns = params['in_chns']
        self.ft_chns = list(params['feature_chns'])
        self.dropout = list(params['dropout'])
        self.dims = list(params['conv_dims'])
        self.n_class = params['class_num']
        self.bilinear = params['bilinear']
        self.num_domains = params['num_domains']
        assert (len(self.ft_chns) == 5)                                    # unet2d5_dsbn.py:277
        if len(self.dims) != 5 or any(d not in (2, 3) for d in self.dims):
            raise ValueError("fplx UNet2D5_dsbn: conv_dims must be five values out of {{2, 3}} (got {0:})".format(self.dims))
        prec = params.get('precision', 'fp32')
        if prec not in ('fp32', 'bf16'):
            raise ValueError("fplx UNet2D5_dsbn: precision must be fp32 or bf16 (got {0:})".format(prec))
        self.act_dtype = torch.float32 if prec == 'fp32' else torch.bfloat16
        ft, nd, dp, dm = self.ft_chns, self.num_domains, self.dropout, self.dims
        self.block0 = DownBlock(self.in_chns, ft[0], nd, dp[0], dm[0])
        self.block1 = DownBlock(ft[0], ft[1], nd, dp[1], dm[1])
        self.block2 = DownBlock(ft[1], ft[2], nd, dp[2], dm[2])
        self.block3 = DownBlock(ft[2], ft[3], nd, dp[3], dm[3])
        self.block4 = DownBlock(ft[3], ft[4], nd, dp[4], dm[4])
        self.up1 = UpBlock(ft[4], ft[3], ft[3], nd, dp[3], dm[3], self.bilinear)          # unet2d5_dsbn.py:284-291: dims[3] .. dims[0]
        self.up2 = UpBlock(ft[3], ft[2], ft[2], nd, dp[2], dm[2], self.bilinear)
        self.up3 = UpBlock(ft[2], ft[1], ft[1], nd, dp[1], dm[1], self.bilinear)
        self.up4 = UpBlock(ft[1], ft[0], ft[0], nd, dp[0], dm[0], self.bilinear)
        self.out_conv = nn.Conv3d(ft[0], self.n_class, kernel_size=(1, 3, 3), padding=(0, 1, 1))

        self.block_keys = ["block0.conv", "block1.conv", "block2.conv", "block3.conv", "block4.conv",
                           "up1.conv", "up2.conv", "up3.conv", "up4.conv"]
        self.block_modules = [self.block0.conv, self.block1.conv, self.block2.conv, self.block3.conv,
                              self.block4.conv, self.up1.conv, self.up2.conv, self.up3.conv, self.up4.conv]
        self.up_modules = [self.up1, self.up2, self.up3, self.up4]
        # dropout stream: (seed, forward counter); deterministic given the seed
        self.dropout_seed = int(params.get('dropout_seed', 1))
        self._fwd_counter = 0
        self.engine = Engine(self)
        self.flat_params = None
        self._layout = None

    # ------------------------------------------------------------------ parameter bookkeeping
    def conv_sites(self):
        for key, m in zip(self.block_keys, self.block_modules):
            yield key + "." + m.cname(1), m.conv_of(1)
            yield key + "." + m.cname(2), m.conv_of(2)

    def deconv_sites(self):
        for j, u in enumerate(self.up_modules):
            yield "up%d.%s" % (j + 1, u.tname()), u.trans()

    def _ordered_param_names(self):
        """Flat layout: [shared parameters in the order their gradients are PRODUCED by backward
        (decoder first) | BN affine parameters of domain 0 | domain 1 | ...].
        A contiguous range of the shared part is complete as early as possible (gradient
        all-reduce buckets, fplx.ddp), and the per-domain tails let Adam skip the domains that
        took no part in a step with one launch per segment - torch.optim.Adam skips parameters
        whose grad is None (reference: only bns[domain_label[0]] is used, dsbn.py:56)."""
        named = dict(self.named_parameters())
        order = ["out_conv.weight", "out_conv.bias"]

        mods = dict(zip(self.block_keys, self.block_modules))

        def block(key):
            m = mods[key]
            for s in ("." + m.cname(2) + ".weight", "." + m.cname(2) + ".bias", ".relu_2.weight",
                      "." + m.cname(1) + ".weight", "." + m.cname(1) + ".bias", ".relu_1.weight"):
                order.append(key + s)

        for j in (4, 3, 2, 1):
            block("up%d.conv" % j)
            t = self.up_modules[j - 1].tname()
            order.extend(["up%d.%s.weight" % (j, t), "up%d.%s.bias" % (j, t)])
        for i in (4, 3, 2, 1, 0):
            block("block%d.conv" % i)
        self._n_shared_names = len(order)
        for d in range(self.num_domains):
            for key in self.block_keys:
                for bn in ("." + mods[key].bname(2), "." + mods[key].bname(1)):
                    order.append("%s%s.bns.%d.weight" % (key, bn, d))
                    order.append("%s%s.bns.%d.bias" % (key, bn, d))
        assert sorted(order) == sorted(named.keys()), "parameter bookkeeping out of sync"
        return order

    def _ensure_flat(self):
        """All parameters live in ONE flat fp32 buffer (views), cf. fused Adam / one all-reduce."""
        named = dict(self.named_parameters())
        first = next(iter(named.values()))
        fp = self.flat_params
        ok = fp is not None and fp.device == first.device
        if ok:
            base, end = fp.data_ptr(), fp.data_ptr() + fp.numel() * 4
            ok = all(base <= p.data_ptr() < end for p in named.values())
        if ok:
            return
        order = self._ordered_param_names()
        # every parameter starts on a 16-byte boundary (4 floats): the vectorised kernels (LDS-tiled weight pack, Adam)
        # need it, and a 1- or 2-element tensor (PReLU slope, out_conv bias) would otherwise knock everything behind it
        # off.  The padding elements are zero, get zero gradients and stay zero.
        total = sum((named[k].numel() + 3) // 4 * 4 for k in order)
        flat = torch.zeros(total, dtype=torch.float32, device=first.device)
        layout, off = {}, 0
        for k in order:
            p = named[k]
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1).float())
            p.data = flat[off:off + n].view(p.shape)
            layout[k] = (off, n, tuple(p.shape))
            off += (n + 3) // 4 * 4
        self.flat_params, self._layout, self._order = flat, layout, order
        self._named = named
        self.engine.invalidate()

    def get_param(self, name):
        return self._named[name]

    def grad_views(self, gflat):
        return {k: gflat[o:o + n].view(shp) for k, (o, n, shp) in self._layout.items()}

    def active_param_names(self, domain):
        tag = ".bns."
        return [k for k in self._order if tag not in k or (".bns.%d." % domain) in k]

    def segments(self):
        """-> (shared (start, end), [domain d (start, end)]) element ranges of the flat buffer"""
        first_bn = self._layout[self._order[self._n_shared_names]][0]
        per = (self.flat_params.numel() - first_bn) // self.num_domains
        return (0, first_bn), [(first_bn + d * per, first_bn + (d + 1) * per) for d in range(self.num_domains)]

    def bucket_ranges(self, min_elems=1 << 20):
        """contiguous (start, end) ranges of the SHARED part in gradient-production order, cut at
        block boundaries, each at least min_elems long (the last one may be shorter)"""
        shared_end = self.segments()[0][1]
        cuts, start, last_block = [], 0, None
        for k in self._order[:self._n_shared_names]:
            blockname = k.split(".")[0]
            o = self._layout[k][0]
            if last_block is not None and blockname != last_block and o - start >= min_elems:
                cuts.append((start, o))
                start = o
            last_block = blockname
        cuts.append((start, shared_end))
        return cuts

    # ------------------------------------------------------------------ nn.Module surface
    def _apply(self, fn, recurse=True):
        r = super(UNet2D5_dsbn, self)._apply(fn, recurse)
        self.flat_params = None          # parameters were re-created: re-flatten lazily
        return r

    def load_state_dict(self, state_dict, strict=True, **kw):
        from .checkpoint import reference_state_keys
        own = set(super(UNet2D5_dsbn, self).state_dict().keys())
        ref = set(reference_state_keys(self.num_domains))
        # the reference's members that are dead in this configuration (the twins of the other dimensionality, the
        # bilinear branch): kept on the host so that a re-saved checkpoint carries them unchanged
        kept = {k: v for k, v in state_dict.items() if k in own or k not in ref}
        self._dead_state = {k: v.detach().cpu().clone() for k, v in state_dict.items() if k not in kept}
        r = super(UNet2D5_dsbn, self).load_state_dict(kept, strict=strict, **kw)
        self.engine.invalidate()
        return r

    def train(self, mode=True):
        self.engine.invalidate()
        return super(UNet2D5_dsbn, self).train(mode)

    def parameters_changed(self):
        """Tell the engine that parameters or running statistics were written behind its back.  torch-side in-place edits of a
        parameter (p.mul_(), p.copy_() under no_grad) are seen through its version counter; writes through `.data`
        (p.data.add_(), the reference's init.*_(m.weight.data) paths), through another tensor that shares the storage, or
        through raw pointers bump NO counter.  Train-mode forwards still catch those on the device - every kept weight pack is
        verified against a stamp of the master values it was made from (whole-tensor writes always, single elements only at
        the sampled positions) - but eval-mode forwards reuse their packs and BatchNorm folds until something the host can see
        changes: call this after such a write (it is what load_state_dict(), train() / eval() and the optimisers do)."""
        self.engine.invalidate()

    def parameters_unchanged_since_last_forward(self):
        """A promise by the caller, consumed by the NEXT train-mode forward: no parameter was touched since the previous
        forward (the second domain of a `training_all` iteration, agent_seg.py:462-486) - that forward then uses the
        previous one's weight packs instead of rebuilding them."""
        self._reuse_packs_once = True

    def dropout_active(self):
        """per ConvBlockND: the nn.Dropout child's own training flag decides (this is what the
        reference flips for test-time dropout, agent_seg.py:845-852)"""
        return [m.dropout.training and m.dropout_p > 0 for m in self.block_modules]

    def forward_mc(self, x, domain_label, passes, out=None):
        """`passes` Monte-Carlo forwards (test-time dropout) of the same batch in one call -> [passes * N, classes, D, H, W],
        pass-major; the encoder levels above the first active dropout are computed once (Engine.forward, mc).  Inference
        only: eval-mode BatchNorm (the reference's test-time dropout, agent_seg.py:845-852, flips the Dropout children only).
        out: optional contiguous fp32 destination of that shape (the Inferer's prediction buffer: no copy afterwards)."""
        if self.training:
            raise RuntimeError("fplx: forward_mc needs eval-mode BatchNorm (net.eval(); Dropout children may be in train mode)")
        if not x.is_cuda:
            raise RuntimeError("fplx UNet2D5_dsbn runs on the GPU only (libfplx.so HIP kernels); got a CPU tensor")
        self._ensure_flat()
        domain = 0 if domain_label is None else int(domain_label[0])
        step = self._fwd_counter
        self._fwd_counter += 1
        with torch.no_grad():
            logits, _ = self.engine.forward(x, domain, False, self.dropout_active(), self.dropout_seed, step, keep=False,
                                            mc=int(passes), out=out)
        return logits

    def forward(self, x, domain_label=None):
        if not x.is_cuda:
            raise RuntimeError("fplx UNet2D5_dsbn runs on the GPU only (libfplx.so HIP kernels); got a CPU tensor")
        self._ensure_flat()
        domain = 0 if domain_label is None else int(domain_label[0])       # dsbn.py:56
        names = self.active_param_names(domain)
        params = [self._named[k] for k in names]
        step = self._fwd_counter
        self._fwd_counter += 1
        return _UNetFunction.apply(x, self, domain, self.training, self.dropout_active(), self.dropout_seed, step,
                                   torch.is_grad_enabled(), *params)

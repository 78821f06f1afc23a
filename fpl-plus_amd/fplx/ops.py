"""Tensor-level wrappers over the C ABI (include/fplx.h).  PyTorch is only plumbing here:
device memory, the current HIP stream and dtype tags.  Activations are 2-D views
[voxels, ld] of NDHWC tensors; a column slice `buf[:, a:b]` is a channel slice.
"""
import torch
from . import _lib
from ._lib import F32, BF16, call

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def dt_of(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise ValueError("fplx: unsupported activation dtype {0:}".format(t.dtype))


def ptr(t):
    return 0 if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("fplx: tensors must live on the GPU (HIP path only, no CPU fallback)")


def cl_strides(D, H, W, ld):
    """element strides (n,d,h,w,c) of an NDHWC tensor with voxel stride ld"""
    return (D * H * W * ld, H * W * ld, W * ld, ld, 1)


def planar_strides(C, D, H, W):
    """element strides (n,d,h,w,c) of a contiguous NCDHW tensor"""
    return (C * D * H * W, H * W, W, 1, D * H * W)


def ld_of(t2d):
    assert t2d.dim() == 2 and t2d.stride(1) == 1
    return t2d.stride(0)


def pack_conv_weight(w, act_dtype, want_wb=True):
    co, ci, kd, kh, kw = w.shape
    wf = torch.empty((kd * kh * kw, co, ci), dtype=act_dtype, device=w.device)
    wb = torch.empty((kd * kh * kw, ci, co), dtype=act_dtype, device=w.device) if want_wb else None
    call("fplx_pack_conv_weight", ptr(w), ptr(wf), ptr(wb), co, ci, kd, kh, kw, _DT[act_dtype], stream())
    return wf, wb


def pack_stamp_floats(cout, cin):
    """floats of a layer's stamp buffer (include/fplx.h, fplx_pack_conv_weights_batched): 32 per 16 x 32-channel pack tile"""
    return (cout // 16) * (cin // 32) * 32


def pack_conv_weights_batched(ws, act_dtype, want_wb, into=None, stamps=None, verify=False):
    """pack_conv_weight for a list of 3x3x3 weights in one launch -> list of (wf, wb).
    into: optional list of existing (wf, wb) destinations (None entries: allocate) - the engine's persistent pack buffers
    stamps: optional list of per-layer fp32 stamp buffers (pack_stamp_floats; None entries: none) that record which master values
    each pack tile was made from; verify=True: `into` and `stamps` hold an earlier pack - only the tiles whose masters changed
    since are packed again (one cheap launch when nothing changed)"""
    import ctypes
    n = len(ws)
    outs = []
    for i, (w, wantb) in enumerate(zip(ws, want_wb)):
        co, ci = w.shape[0], w.shape[1]
        assert tuple(w.shape[2:]) == (3, 3, 3) and w.dtype == torch.float32 and w.is_contiguous()
        if into is not None and into[i] is not None:
            outs.append(into[i])
            continue
        assert not verify, "verify needs the earlier pack (into)"
        wf = torch.empty((27, co, ci), dtype=act_dtype, device=w.device)
        wb = torch.empty((27, ci, co), dtype=act_dtype, device=w.device) if wantb else None
        outs.append((wf, wb))
    vp = ctypes.c_void_p * n
    ip = ctypes.c_int * n
    st = None
    if stamps is not None:
        for w, s_ in zip(ws, stamps):
            assert s_ is None or (s_.dtype == torch.float32 and s_.numel() >= pack_stamp_floats(w.shape[0], w.shape[1]))
        st = vp(*[ptr(s_) or None for s_ in stamps])
    call("fplx_pack_conv_weights_batched", n, vp(*[ptr(w) for w in ws]), vp(*[ptr(o[0]) for o in outs]),
         vp(*[ptr(o[1]) or None for o in outs]), ip(*[w.shape[0] for w in ws]), ip(*[w.shape[1] for w in ws]),
         _DT[act_dtype], st, 1 if verify else 0, stream())
    return outs


def pack_weights_multi(jobs):
    """jobs: [(kind, w fp32 contiguous, wf or None, wb or None, a, b, taps)] (kind 0: conv weight [a = Cout][b = Cin][taps];
    1: transposed conv [a = Cin][b = Cout][taps]) - all of them in ONE launch (fplx_pack_weights_multi)"""
    import ctypes
    n = len(jobs)
    vp, ip = ctypes.c_void_p * n, ctypes.c_int * n
    dts = [_DT[(j[2] if j[2] is not None else j[3]).dtype] for j in jobs]
    call("fplx_pack_weights_multi", n, ip(*[j[0] for j in jobs]), vp(*[ptr(j[1]) for j in jobs]), vp(*[ptr(j[2]) or None for j in jobs]),
         vp(*[ptr(j[3]) or None for j in jobs]), ip(*[j[4] for j in jobs]), ip(*[j[5] for j in jobs]), ip(*[j[6] for j in jobs]),
         ip(*dts), stream())


def pack_deconv_weight(w, act_dtype):
    """ConvTranspose3d weight [Cin,Cout,2,2,2] (8 taps) or ConvTranspose2d weight [Cin,Cout,2,2] (4 taps)"""
    ci, co = w.shape[0], w.shape[1]
    taps = 8 if w.dim() == 5 else 4
    wf = torch.empty((taps, co, ci), dtype=act_dtype, device=w.device)
    wb = torch.empty((taps, ci, co), dtype=act_dtype, device=w.device)
    call("fplx_pack_deconv_weight" if taps == 8 else "fplx_pack_deconv122_weight", ptr(w), ptr(wf), ptr(wb), ci, co,
         _DT[act_dtype], stream())
    return wf, wb


def pack_conv2d_weight(w, act_dtype, want_wb=True):
    """Conv2d weight [Cout,Cin,3,3] -> the 27-tap layouts of pack_conv_weight with the 9 taps in the middle depth plane"""
    co, ci = w.shape[0], w.shape[1]
    wf = torch.empty((27, co, ci), dtype=act_dtype, device=w.device)
    wb = torch.empty((27, ci, co), dtype=act_dtype, device=w.device) if want_wb else None
    call("fplx_pack_conv2d_weight", ptr(w), ptr(wf), ptr(wb), co, ci, _DT[act_dtype], stream())
    return wf, wb


def conv3d_stats_rows(dims, cin, cout, k, x_dt, y_dt, mid=False):
    """mid: the pack is a Conv2d in the middle depth plane (pack_conv2d_weight) - the fplx_conv2d_* form"""
    n, d, h, w = dims
    if mid:
        return _lib.lib().fplx_conv2d_stats_rows(n, d, h, w, cin, cout, x_dt, y_dt)
    return _lib.lib().fplx_conv3d_stats_rows(n, d, h, w, cin, cout, k[0], k[1], k[2], x_dt, y_dt)


_fwd_ws = {}


def conv3d_fwd_ws_bytes(dims, cin, cout, k, x_dt, y_dt, mid=False):
    n, d, h, w = dims
    if mid:
        return _lib.lib().fplx_conv2d_fwd_ws_bytes(n, d, h, w, cin, cout, x_dt, y_dt)
    return _lib.lib().fplx_conv3d_fwd_ws_bytes(n, d, h, w, cin, cout, k[0], k[1], k[2], x_dt, y_dt)


def conv3d_fwd(x, xs, x_dt, wp, bias, y, ys, y_dt, dims, cin, cout, k, stats=None, ws=None, mid=False):
    n, d, h, w = dims
    if ws is None:
        need = conv3d_fwd_ws_bytes(dims, cin, cout, k, x_dt, y_dt, mid)
        if need:
            # scratch for the split-K kernels, per (device, stream): launches on one stream are ordered and may share it,
            # launches on different streams (the engine runs two) may not
            key = (y.device, torch.cuda.current_stream(y.device).cuda_stream)
            if key not in _fwd_ws or _fwd_ws[key].numel() < need:
                _fwd_ws[key] = torch.empty(int(need), dtype=torch.uint8, device=y.device)
            ws = _fwd_ws[key]
    nws = 0 if ws is None else ws.numel() * ws.element_size()
    if mid:
        assert tuple(k) == (3, 3, 3)
        call("fplx_conv2d_fwd", ptr(x), x_dt, xs[0], xs[1], xs[2], xs[3], xs[4], ptr(wp), ptr(bias),
             ptr(y), y_dt, ys[0], ys[1], ys[2], ys[3], ys[4], n, d, h, w, cin, cout, ptr(stats), ptr(ws), nws, stream())
        return
    call("fplx_conv3d_fwd", ptr(x), x_dt, xs[0], xs[1], xs[2], xs[3], xs[4], ptr(wp), ptr(bias),
         ptr(y), y_dt, ys[0], ys[1], ys[2], ys[3], ys[4], n, d, h, w, cin, cout, k[0], k[1], k[2],
         ptr(stats), ptr(ws), nws, stream())


def conv3d_wgrad_ws_bytes(dims, cin, cout, k):
    n, d, h, w = dims
    return _lib.lib().fplx_conv3d_wgrad_ws_bytes(n, d, h, w, cin, cout, k[0], k[1], k[2])


def conv3d_wgrad(x, xs, x_dt, dy, ys, dy_dt, dw, db, dims, cin, cout, k, ws):
    n, d, h, w = dims
    call("fplx_conv3d_wgrad", ptr(x), x_dt, xs[0], xs[1], xs[2], xs[3], xs[4],
         ptr(dy), dy_dt, ys[0], ys[1], ys[2], ys[3], ys[4], ptr(dw), ptr(db), n, d, h, w, cin, cout,
         k[0], k[1], k[2], ptr(ws), ws.numel() * ws.element_size(), stream())


def conv2d_wgrad_ws_bytes(dims, cin, cout):
    n, d, h, w = dims
    return _lib.lib().fplx_conv2d_wgrad_ws_bytes(n, d, h, w, cin, cout)


def conv2d_wgrad(x, xs, x_dt, dy, ys, dy_dt, dw9, db, dims, cin, cout, ws):
    """weight gradient of a Conv2d(3x3) per depth slice: dw9 fp32 [Cout, Cin, 3, 3] (9 of 27 taps on the MFMA path)"""
    n, d, h, w = dims
    call("fplx_conv2d_wgrad", ptr(x), x_dt, xs[0], xs[1], xs[2], xs[3], xs[4], ptr(dy), dy_dt, ys[0], ys[1], ys[2], ys[3],
         ys[4], ptr(dw9), ptr(db), n, d, h, w, cin, cout, ptr(ws), ws.numel() * ws.element_size(), stream())


def conv3d_cat2_ok(dims, cin, cout):
    """True if the 3x3x3 convolution on cat([x0, x1], channel) of two cin/2-channel bf16 tensors has the split fast
    path (level 0 of the 32-base network): the concatenation is then never materialised."""
    n, d, h, w = dims
    return _lib.lib().fplx_conv3d_cat2_ok(n, d, h, w, cin, cout) == 1


def conv3d_fwd_cat2(x0, x1, wp, bias, y, dims, cin, cout, stats=None, mid=False):
    n, d, h, w = dims
    assert ld_of(x0) == ld_of(x1)
    call("fplx_conv2d_fwd_cat2" if mid else "fplx_conv3d_fwd_cat2", ptr(x0), ptr(x1), ld_of(x0), ptr(wp), ptr(bias), ptr(y), ld_of(y), n, d, h, w, cin,
         cout, ptr(stats), stream())


def conv3d_dgrad_split2(dy, wb, dx0, dx1, dims, cin, cout, mid=False):
    n, d, h, w = dims
    assert ld_of(dx0) == ld_of(dx1)
    call("fplx_conv2d_dgrad_split2" if mid else "fplx_conv3d_dgrad_split2", ptr(dy), ld_of(dy), ptr(wb), ptr(dx0), ptr(dx1), ld_of(dx0), n, d, h, w, cin,
         cout, stream())


def conv3d_wgrad_cat2(x0, x1, dy, dw, dims, cin, cout, ws, mid=False):
    """mid: dw is the 9-tap Conv2d gradient [Cout, Cin, 3, 3]"""
    n, d, h, w = dims
    assert ld_of(x0) == ld_of(x1)
    call("fplx_conv2d_wgrad_cat2" if mid else "fplx_conv3d_wgrad_cat2", ptr(x0), ptr(x1), ld_of(x0), ptr(dy), ld_of(dy),
         ptr(dw), n, d, h, w, cin, cout, ptr(ws), ws.numel() * ws.element_size(), stream())


def conv3d_fwd_act_ok(dims, cin, cout, mid=False, cat2=False):
    """True if the layer's forward kernel has the fused (folded eval-mode BatchNorm) + PReLU write-out"""
    n, d, h, w = dims
    return _lib.lib().fplx_conv3d_fwd_act_ok(n, d, h, w, cin, cout, 1 if mid else 0, 1 if cat2 else 0) == 1


def conv3d_fwd_act(x0, x1, wp, bias, slope, y, dims, cin, cout, mid=False, n_x0=0):
    """inference: y = PReLU(conv(x; wp) + bias) with the eval-mode BatchNorm already folded into wp / bias; x1: the second half
    of a channel concatenation (or None); n_x0: x0 holds that many samples, read modulo (0 = all).  bf16 NDHWC 2-D views."""
    n, d, h, w = dims
    ws = None
    need = conv3d_fwd_ws_bytes(dims, cin, cout, (3, 3, 3), BF16, BF16, mid)
    if need:
        key = (y.device, torch.cuda.current_stream(y.device).cuda_stream)
        if key not in _fwd_ws or _fwd_ws[key].numel() < need:
            _fwd_ws[key] = torch.empty(int(need), dtype=torch.uint8, device=y.device)
        ws = _fwd_ws[key]
    if x1 is not None:
        assert ld_of(x0) == ld_of(x1)
    call("fplx_conv3d_fwd_act", ptr(x0), ptr(x1), ld_of(x0), ptr(wp), ptr(bias), ptr(slope), ptr(y), ld_of(y), n, d, h, w, cin,
         cout, 1 if mid else 0, int(n_x0), ptr(ws), 0 if ws is None else ws.numel(), stream())


def _dc(sd):
    return "fplx_deconv2_" if sd == 2 else "fplx_deconv122_"


def deconv2_fwd(x, wf, bias, y, dims, cin, cout, sd=2):
    """sd = 2: ConvTranspose3d(2,2); sd = 1: ConvTranspose2d(2,2) on every depth slice (dims = INPUT dims)"""
    n, d, h, w = dims
    call(_dc(sd) + "fwd", ptr(x), ld_of(x), ptr(wf), ptr(bias), ptr(y), ld_of(y), n, d, h, w, cin, cout,
         dt_of(x), stream())


def deconv2_dgrad(dy, wb, dx, dims, cin, cout, sd=2):
    n, d, h, w = dims
    call(_dc(sd) + "dgrad", ptr(dy), ld_of(dy), ptr(wb), ptr(dx), ld_of(dx), n, d, h, w, cin, cout,
         dt_of(dy), stream())


def deconv2_wgrad_ws_bytes(dims, cin, cout, sd=2):
    n, d, h, w = dims
    return getattr(_lib.lib(), _dc(sd) + "wgrad_ws_bytes")(n, d, h, w, cin, cout)


def deconv2_wgrad(x, dy, dw, db, dims, cin, cout, ws, sd=2):
    n, d, h, w = dims
    call(_dc(sd) + "wgrad", ptr(x), ld_of(x), ptr(dy), ld_of(dy), ptr(dw), ptr(db), n, d, h, w, cin, cout,
         dt_of(x), ptr(ws), ws.numel() * ws.element_size(), stream())


def upsample2_fwd(x, y, dims, c, sd=2):
    """(tri / bi)linear x2 upsampling, align_corners = True; dims = INPUT dims, sd = 1: H and W only (2.5D levels)"""
    n, d, h, w = dims
    call("fplx_upsample2_fwd", ptr(x), ld_of(x), ptr(y), ld_of(y), n, d, h, w, c, dt_of(x), sd, stream())


def upsample2_bwd(dy, dx, dims, c, sd=2):
    n, d, h, w = dims
    call("fplx_upsample2_bwd", ptr(dy), ld_of(dy), ptr(dx), ld_of(dx), n, d, h, w, c, dt_of(dy), sd, stream())


def bn_train_finalize(stats, rows, c, count, gamma, beta, rm, rv, nbt, bnbuf, momentum=0.1, eps=1e-5):
    """bnbuf: fp32 [4, C] -> rows mean, rstd, scale, shift"""
    call("fplx_bn_train_finalize", ptr(stats), rows, c, count, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nbt),
         momentum, eps, ptr(bnbuf[0]), ptr(bnbuf[1]), ptr(bnbuf[2]), ptr(bnbuf[3]), stream())


def bn_eval_prepare(gamma, beta, rm, rv, bnbuf, eps=1e-5):
    c = gamma.numel()
    call("fplx_bn_eval_prepare", ptr(gamma), ptr(beta), ptr(rm), ptr(rv), eps, c, ptr(bnbuf[2]), ptr(bnbuf[3]),
         stream())


def bn_act_fwd(y, out, bnbuf, slope, p, seed, sid, c):
    call("fplx_bn_act_fwd", ptr(y), ld_of(y), ptr(out), ld_of(out), ptr(bnbuf[2]), ptr(bnbuf[3]), ptr(slope),
         float(p), int(seed), int(sid), y.shape[0], c, dt_of(y), stream())


def num_partials(voxels):
    return _lib.lib().fplx_num_partials(voxels)


def outconv_bn_rows(dims, c0=32, ncls=2):
    """partial rows of the fused out_conv backward (fplx_outconv_bn_rows); 0 where the fused forms do not apply"""
    n, d, h, w = dims
    return _lib.lib().fplx_outconv_bn_rows(n, d, h, w, int(c0), int(ncls))


def outconv_bn_ok(dims, c0, ncls):
    """True if out_conv can be fused with the BatchNorm + PReLU passes of the site in front of it (fplx_outconv_fwd_bn,
    fplx_outconv_dgrad_bn_reduce / _apply: bf16, C0 = 32, classes <= 4)"""
    return outconv_bn_rows(dims, c0, ncls) > 0


def outconv_fwd_bn(y, bnbuf, slope, a, wf, bias, logits, dims, c0, ncls):
    """a = PReLU(BN(y)) (bnbuf rows 2 / 3 = scale / shift) and logits = out_conv(a) in one pass over y.  a = None: the
    activation is not written (allowed where outconv_wgrad_bn_ws_bytes(...) > 0: backward then takes everything from y)"""
    n, d, h, w = dims
    call("fplx_outconv_fwd_bn", ptr(y), ld_of(y), ptr(bnbuf[2]), ptr(bnbuf[3]), ptr(slope), ptr(a), int(c0) if a is None else ld_of(a),
         ptr(wf), ptr(bias), ptr(logits), n, d, h, w, int(c0), int(ncls), stream())


def outconv_wgrad_bn_ws_bytes(dims, c0, ncls):
    """workspace of outconv_wgrad_bn; 0 where that form is not available (the stored activation + conv3d_wgrad are needed)"""
    n, d, h, w = dims
    return int(_lib.lib().fplx_outconv_wgrad_bn_ws_bytes(n, d, h, w, int(c0), int(ncls)))


def outconv_wgrad_bn(y, bnbuf, slope, dlogits, dw, db, dims, c0, ncls, ws):
    """out_conv's weight (and bias, db may be None) gradient from the PRE-BatchNorm tensor y of the site in front of it: the
    activation is formed on the way in (fplx_outconv_wgrad_bn), so the forward never has to store it"""
    n, d, h, w = dims
    call("fplx_outconv_wgrad_bn", ptr(y), ld_of(y), ptr(bnbuf[2]), ptr(bnbuf[3]), ptr(slope), ptr(dlogits), ptr(dw), ptr(db),
         n, d, h, w, int(c0), int(ncls), ptr(ws), ws.numel() * ws.element_size(), stream())


def outconv_dgrad_bn_bwd(dlogits, wb, y, bnbuf, slope, train, dgamma, dbeta, dslope, part, coef, dy, dims, c0, ncls):
    """backward of [BN + PReLU -> out_conv] without ever storing out_conv's data gradient: reduction (recomputing it from
    dlogits), the usual finalize, apply (recomputing it again) -> dy.  part must hold outconv_bn_rows(dims, c0, ncls) x (2 c0 + 1) floats."""
    n, d, h, w = dims
    rows = outconv_bn_rows(dims, c0, ncls)
    if part.numel() < rows * (2 * c0 + 1):
        raise ValueError("fplx: partial-row buffer too small for the fused out_conv backward (%d < %d floats)" % (part.numel(), rows * (2 * c0 + 1)))
    bn = [ptr(bnbuf[0]), ptr(bnbuf[1]), ptr(bnbuf[2]), ptr(bnbuf[3])]
    call("fplx_outconv_dgrad_bn_reduce", ptr(dlogits), ptr(wb), ptr(y), ld_of(y), bn[0], bn[1], bn[2], bn[3], ptr(slope), ptr(part),
         n, d, h, w, int(c0), int(ncls), stream())
    call("fplx_bn_act_bwd_finalize", ptr(part), rows, int(c0), y.shape[0], 1 if train else 0, ptr(dgamma), ptr(dbeta), ptr(dslope),
         ptr(coef), stream())
    call("fplx_outconv_dgrad_bn_apply", ptr(dlogits), ptr(wb), ptr(y), ld_of(y), bn[0], bn[1], bn[2], bn[3], ptr(slope), ptr(coef),
         ptr(dy), ld_of(dy), n, d, h, w, int(c0), int(ncls), stream())


def bn_pool_fused_ok(c, dtype):
    return _lib.lib().fplx_bn_pool_fused_ok(int(c), _DT[dtype]) == 1


def bn_act_pool_fwd(y, out, pooled, bnbuf, slope, dims, c, pd=2):
    """tail of a DownBlock in one pass: out = PReLU(BN(y)) (the skip tensor) and pooled = MaxPool(out)"""
    n, d, h, w = dims
    call("fplx_bn_act_pool_fwd", ptr(y), ld_of(y), ptr(out), ld_of(out), ptr(pooled), ld_of(pooled), ptr(bnbuf[2]), ptr(bnbuf[3]),
         ptr(slope), n, d, h, w, c, dt_of(y), pd, stream())


def pool_bwd_bn_reduce(y, dy, dskip, dx, bnbuf, slope, dims, c, part, pd=2):
    """dx = dskip + unpooled(dy) (= d of the DownBlock's output) and the partial rows of bn_act_bwd's reduction over it"""
    n, d, h, w = dims
    call("fplx_pool_bwd_bn_reduce", ptr(y), ld_of(y), ptr(dy), ld_of(dy), ptr(dskip), 0 if dskip is None else ld_of(dskip), ptr(dx),
         ld_of(dx), ptr(bnbuf[0]), ptr(bnbuf[1]), ptr(bnbuf[2]), ptr(bnbuf[3]), ptr(slope), n, d, h, w, c, dt_of(y), pd, ptr(part),
         stream())


def bn_act_bwd(y, dout, dy, bnbuf, slope, p, seed, sid, c, train, dgamma, dbeta, dslope, part, coef, reduced=False, apply=True):
    """three-stage backward of the fused BN-apply + PReLU + dropout pass; dy may alias dout.
    reduced: `part` already holds the partial rows (pool_bwd_bn_reduce wrote them while it formed dout)
    apply=False: reduction + finalize only - `coef` is left for a consumer that forms dy itself (stem_wgrad_bn)"""
    v = y.shape[0]
    rows = num_partials(v)
    if not reduced:
        call("fplx_bn_act_bwd_reduce", ptr(y), ld_of(y), ptr(dout), ld_of(dout), ptr(bnbuf[0]), ptr(bnbuf[1]),
             ptr(bnbuf[2]), ptr(bnbuf[3]), ptr(slope), float(p), int(seed), int(sid), v, c, dt_of(y), ptr(part), stream())
    call("fplx_bn_act_bwd_finalize", ptr(part), rows, c, v, 1 if train else 0, ptr(dgamma), ptr(dbeta), ptr(dslope),
         ptr(coef), stream())
    if not apply:
        return
    call("fplx_bn_act_bwd_apply", ptr(y), ld_of(y), ptr(dout), ld_of(dout), ptr(dy), ld_of(dy), ptr(bnbuf[0]),
         ptr(bnbuf[1]), ptr(bnbuf[2]), ptr(bnbuf[3]), ptr(slope), ptr(coef), float(p), int(seed), int(sid), v, c,
         dt_of(y), stream())


def stem_wgrad_bn_ok(dims, cin, cout):
    """True if the stem site's weight gradient can take y and d(a) and form dy itself (fplx_stem_wgrad_bn): the layers
    fplx_conv3d_plan_query gives to the MFMA stem kernels (fp32 NCDHW input of 1 | 4 channels, C0 % 32 == 0)"""
    import ctypes
    n, d, h, w = dims
    kern, g, ks, r = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(_lib.lib().fplx_conv3d_plan_query(n, d, h, w, int(cin), int(cout), 3, 3, 3, F32, BF16, ctypes.byref(kern),
                                                 ctypes.byref(g), ctypes.byref(ks), ctypes.byref(r)))
    return kern.value == 6                               # FPLX_KERNEL_STEM


def stem_wgrad_bn(x, y, dout, bnbuf, slope, coef, dw, dims, cin, cout, ws):
    """weight gradient of the stem convolution from y (its stored output) and dout = gradient w.r.t. the site's OUTPUT: the apply
    pass of the site's BatchNorm + PReLU backward runs on the pieces the kernel stages; dy is never stored"""
    n, d, h, w = dims
    call("fplx_stem_wgrad_bn", ptr(x), ptr(y), ld_of(y), ptr(dout), ld_of(dout), ptr(bnbuf[0]), ptr(bnbuf[1]), ptr(bnbuf[2]),
         ptr(bnbuf[3]), ptr(slope), ptr(coef), ptr(dw), n, d, h, w, int(cin), int(cout), ptr(ws), ws.numel() * ws.element_size(),
         stream())


def maxpool2_fwd(x, y, dims, c, pd=2):
    """pd = 2: MaxPool3d(2); pd = 1: MaxPool2d(2) on every depth slice"""
    n, d, h, w = dims
    call("fplx_maxpool2_fwd" if pd == 2 else "fplx_maxpool122_fwd", ptr(x), ld_of(x), ptr(y), ld_of(y), n, d, h, w, c, dt_of(x), stream())


def maxpool2_bwd(x, dy, dskip, dx, dims, c, pd=2):
    n, d, h, w = dims
    call("fplx_maxpool2_bwd" if pd == 2 else "fplx_maxpool122_bwd", ptr(x), ld_of(x), ptr(dy), ld_of(dy), ptr(dskip), 0 if dskip is None else ld_of(dskip),
         ptr(dx), ld_of(dx), n, d, h, w, c, dt_of(x), stream())


def loss_rows(v):
    return _lib.lib().fplx_loss_rows(v)


def loss_k(c):
    return 6 * c + 3


def seg_loss_fwd(logits, label, pw, iw, weights, softmax, part, out, coef):
    n, c = logits.shape[0], logits.shape[1]
    v = logits[0, 0].numel()
    call("fplx_seg_loss_fwd", ptr(logits), ptr(label), ptr(pw), ptr(iw), n, c, v, weights[0], weights[1], weights[2],
         weights[3], 1 if softmax else 0, ptr(part), ptr(out), ptr(coef), stream())


def seg_loss_fwd_dist(logits, label, pw, iw, weights, softmax, part, out, coef, group):
    """seg_loss_fwd under data parallelism: the sums over the local samples are all-reduced over `group` between the
    reduction and the evaluation, so loss, metric and backward coefficients are those of the FULL batch (what the
    reference's nn.DataParallel computes on its gathered logits, agent_seg.py:692-698).  All ranks must hold the same
    number of samples."""
    import torch.distributed as dist
    n, c = logits.shape[0], logits.shape[1]
    v = logits[0, 0].numel()
    k = loss_k(c)
    sums = torch.empty((n + 1, k), dtype=torch.float64, device=logits.device)
    call("fplx_seg_loss_sums", ptr(logits), ptr(label), ptr(pw), n, c, v, 1 if softmax else 0, ptr(part), ptr(sums[:n]),
         ptr(sums[n]), stream())
    dist.all_reduce(sums[n], op=dist.ReduceOp.SUM, group=group)
    world = dist.get_world_size(group)
    call("fplx_seg_loss_from_sums", ptr(sums[:n]), ptr(sums[n]), ptr(iw), n, n * world, c, v, 0 if pw is None else 1,
         weights[0], weights[1], weights[2], weights[3], ptr(out), ptr(coef), stream())


def seg_loss_bwd(logits, label, pw, coef, gscale, weights, softmax, dlogits):
    n, c = logits.shape[0], logits.shape[1]
    v = logits[0, 0].numel()
    call("fplx_seg_loss_bwd", ptr(logits), ptr(label), ptr(pw), ptr(coef), ptr(gscale), n, c, v, weights[0],
         weights[1], weights[2], weights[3], 1 if softmax else 0, ptr(dlogits), stream())


def adam_step(p, g, m, v, lr, step, weight_decay, grad_scale=1.0, betas=(0.9, 0.999), eps=1e-8):
    call("fplx_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, betas[0], betas[1], eps, weight_decay,
         int(step), grad_scale, stream())


def adam_pack_ok(cout, cin):
    return _lib.lib().fplx_adam_pack_ok(int(cout), int(cin)) == 1


def adam_pack_step(p, g, m, v, lr, step, weight_decay, grad_scale, betas, eps, layers):
    """fplx_adam_step over the flat segment p AND the bf16 packs of the 3x3x3 weights inside it, one launch.
    layers: [(element offset in p, cout, cin, wf, wb[, stamp])] ascending by offset (stamp: pack_conv_weights_batched)"""
    import ctypes
    n = len(layers)
    vp, ip, lp = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_int64 * n
    st = vp(*[(ptr(l[5]) or None) if len(l) > 5 else None for l in layers])
    call("fplx_adam_pack_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, betas[0], betas[1], eps, weight_decay, int(step),
         grad_scale, n, lp(*[int(l[0]) for l in layers]), ip(*[int(l[1]) for l in layers]), ip(*[int(l[2]) for l in layers]),
         vp(*[ptr(l[3]) for l in layers]), vp(*[ptr(l[4]) or None for l in layers]), st, stream())


def mc_filter(logits_tcv, thr=0.01, want_hards=True, want_maps=False):
    """logits_tcv: fp32 [T, C, ...volume...] on the GPU -> dict (device tensors, no sync)"""
    require_gpu(logits_tcv)
    t, c = logits_tcv.shape[0], logits_tcv.shape[1]
    vol = tuple(logits_tcv.shape[2:])
    v = logits_tcv[0, 0].numel()
    dev = logits_tcv.device
    lg = logits_tcv.contiguous()
    hards = torch.empty((t,) + vol, dtype=torch.uint8, device=dev) if want_hards else None
    mean = torch.empty(vol, dtype=torch.float32, device=dev) if want_maps else None
    unc = torch.empty(vol, dtype=torch.float32, device=dev) if want_maps else None
    part = torch.empty((num_partials(v), 2), dtype=torch.float64, device=dev)
    out = torch.empty(4, dtype=torch.float64, device=dev)
    call("fplx_mc_filter", ptr(lg), t, c, v, thr, ptr(hards), ptr(mean), ptr(unc), ptr(part), ptr(out), stream())
    return dict(hards=hards, means=mean, uncertainty=unc, stats=out)


def hard_label(logits):
    """save_outputs (reference agent_seg.py:1049-1050): [N,C,...] fp32 -> uint8 [N,...]"""
    require_gpu(logits)
    n, c = logits.shape[0], logits.shape[1]
    lg = logits.contiguous()
    out = torch.empty((n,) + tuple(logits.shape[2:]), dtype=torch.uint8, device=logits.device)
    call("fplx_hard_label", ptr(lg), n, c, lg[0, 0].numel(), ptr(out), stream())
    return out


def pixel_weight(mask_a, mask_b, image_weight=None):
    """reference data/get_pixel_weight.py:21-26 (+ NiftyDataset.set_weight_ when image_weight is given)"""
    require_gpu(mask_a, mask_b)
    if mask_a.shape != mask_b.shape:
        raise ValueError("fplx: mask shapes differ")        # get_pixel_weight.py:19 asserts the same
    a = mask_a.contiguous()
    b = mask_b.contiguous()
    out = torch.empty(a.shape, dtype=torch.float32, device=a.device)
    call("fplx_pixel_weight", ptr(a), ptr(b), a.numel(), 0 if image_weight is None else 1,
         0.0 if image_weight is None else float(image_weight), ptr(out), stream())
    return out


# ---------------------------------------------------------------- training-sample transforms (csrc/sample.hip)
_ELEM = {torch.float32: 4, torch.uint8: 1}


def _elem_bytes(t):
    try:
        return _ELEM[t.dtype]
    except KeyError:
        raise ValueError("fplx: transforms take float32 or uint8 volumes, got {0:}".format(t.dtype))


def normalize_mean_std(x, mean_std=None, out=None, want_moments=False):
    """(x - mean) / std of one channel volume (fp32, contiguous); mean_std None -> x's own float32 mean / population std"""
    require_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty_like(x) if out is None else out
    ms = None if mean_std is None else torch.tensor(list(mean_std), dtype=torch.float32, device=x.device)
    nb = _lib.lib().fplx_normalize_ws_bytes()
    ws = torch.empty(nb // 8, dtype=torch.float64, device=x.device)
    got = torch.empty(2, dtype=torch.float32, device=x.device) if want_moments else None
    call("fplx_normalize_mean_std", ptr(x), ptr(out), x.numel(), ptr(ms), ptr(ws), nb, ptr(got), stream())
    return (out, got) if want_moments else out


def normalize_positive(x, noise, out=None, want_moments=False):
    """NormalizeWithMeanStd_ignore_non_positive: (x - mean) / std with the moments of the voxels > 0; noise elsewhere"""
    require_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and noise.dtype == torch.float32 and noise.is_contiguous()
    assert noise.numel() == x.numel()
    out = torch.empty_like(x) if out is None else out
    nb = _lib.lib().fplx_normalize_ws_bytes()
    ws = torch.empty(nb // 8, dtype=torch.float64, device=x.device)
    got = torch.empty(2, dtype=torch.float32, device=x.device) if want_moments else None
    call("fplx_normalize_positive", ptr(x), ptr(noise), ptr(out), x.numel(), ptr(ws), nb, ptr(got), stream())
    return (out, got) if want_moments else out


def pad_reflect(x, lower, out_size):
    """numpy.pad(x, mode='reflect') of a [C,D,H,W] volume to out_size (D,H,W) with lower margins `lower`"""
    require_gpu(x)
    assert x.dim() == 4 and x.is_contiguous()
    c, d, h, w = x.shape
    y = torch.empty((c,) + tuple(out_size), dtype=x.dtype, device=x.device)
    call("fplx_pad_reflect", ptr(x), ptr(y), _elem_bytes(x), c, d, h, w, lower[0], lower[1], lower[2], out_size[0],
         out_size[1], out_size[2], stream())
    return y


def crop_flip(x, crop_min, out_size, flip_mask=0):
    """crop box [crop_min, crop_min + out_size) of a [C,D,H,W] volume, then flip (bit0 W, bit1 H, bit2 D)"""
    require_gpu(x)
    assert x.dim() == 4 and x.is_contiguous()
    c, d, h, w = x.shape
    y = torch.empty((c,) + tuple(out_size), dtype=x.dtype, device=x.device)
    call("fplx_crop_flip", ptr(x), ptr(y), _elem_bytes(x), c, d, h, w, crop_min[0], crop_min[1], crop_min[2], out_size[0],
         out_size[1], out_size[2], int(flip_mask), stream())
    return y


def label_bbox(label, mask_labels):
    """-> (count, bb_min[4], bb_max[4]) of {label in mask_labels} on a uint8 [C,D,H,W] label (one device->host copy)"""
    require_gpu(label)
    assert label.dtype == torch.uint8 and label.dim() == 4 and label.is_contiguous()
    c, d, h, w = label.shape
    ml = torch.tensor([int(v) for v in mask_labels], dtype=torch.int32, device=label.device)
    out = torch.empty(9, dtype=torch.int32, device=label.device)
    call("fplx_label_bbox", ptr(label), c, d, h, w, ptr(ml), ml.numel(), ptr(out), stream())
    o = out.tolist()
    return o[0], o[1:5], o[5:9]


def label_to_probability(label, class_num):
    """uint8 label [D,H,W] (or any shape) -> fp32 one-hot [class_num, *shape]"""
    require_gpu(label)
    assert label.dtype == torch.uint8 and label.is_contiguous()
    prob = torch.empty((class_num,) + tuple(label.shape), dtype=torch.float32, device=label.device)
    call("fplx_label_to_probability", ptr(label), ptr(prob), class_num, label.numel(), stream())
    return prob


def set_weight_(pixel_weight, image_weight):
    """NiftyDataset.set_weight_ in place on a fp32 device volume"""
    require_gpu(pixel_weight)
    assert pixel_weight.dtype == torch.float32 and pixel_weight.is_contiguous()
    call("fplx_set_weight", ptr(pixel_weight), pixel_weight.numel(), float(image_weight), stream())
    return pixel_weight


def overlap_counts(seg, gt, labels, fuse=False):
    """exact voxel counts [rows, 3] = (|seg==l & gt==l|, |seg==l|, |gt==l|) as python ints (one device->host copy)"""
    require_gpu(seg, gt)
    if seg.shape != gt.shape:
        raise ValueError("fplx: segmentation and ground truth shapes differ")
    assert seg.dtype == torch.uint8 and gt.dtype == torch.uint8
    s, g = seg.contiguous(), gt.contiguous()
    lab = torch.tensor([int(v) for v in labels], dtype=torch.int32, device=s.device)
    rows = 1 if fuse else lab.numel()
    out = torch.empty((rows, 3), dtype=torch.int64, device=s.device)
    call("fplx_overlap_counts", ptr(s), ptr(g), s.numel(), ptr(lab), lab.numel(), 1 if fuse else 0, ptr(out), stream())
    return out.tolist()


def edge_points(mask):
    """get_edge_points of a binary uint8 device volume [D, H, W] ([H, W]: the 2D form) -> uint8 edge map"""
    require_gpu(mask)
    assert mask.dtype == torch.uint8 and mask.dim() in (2, 3)
    m = mask.contiguous()
    d, h, w = (1,) + tuple(m.shape) if m.dim() == 2 else tuple(m.shape)
    edge = torch.empty_like(m)
    call("fplx_surface_edge_points", ptr(m), d, h, w, ptr(edge), stream())
    return edge


def surface_min_dist(query_zyx, seed_zyx, spacing):
    """distance (GeodisTK raster-scan metric on a constant image) from each int32 (z, y, x) query row to the nearest
    seed row -> fp32 [nq]; 1e10 everywhere when there is no seed"""
    require_gpu(query_zyx)
    assert query_zyx.dtype == torch.int32 and query_zyx.dim() == 2 and query_zyx.shape[1] == 3
    assert seed_zyx.dtype == torch.int32 and seed_zyx.dim() == 2 and seed_zyx.shape[1] == 3
    q, sd = query_zyx.contiguous(), seed_zyx.contiguous()
    out = torch.empty((q.shape[0],), dtype=torch.float32, device=q.device)
    if q.shape[0] == 0:
        return out
    call("fplx_surface_min_dist", ptr(q), q.shape[0], ptr(sd) if sd.shape[0] else 0, sd.shape[0], float(spacing[0]),
         float(spacing[1]), float(spacing[2]), ptr(out), stream())
    return out

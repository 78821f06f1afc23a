"""Times the five HBM-bound edge kernels at the level-0 shape (2 x 80 x 160 x 160): stem fwd / wgrad (fp32 NCDHW input
-> 32 bf16 channels) and out_conv fwd / dgrad / wgrad (32 bf16 channels <-> 2 fp32 logit planes, kernel (1,3,3));
prints us and GB/s against each kernel's algorithmic bytes."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fpl-plus_amd"))
import torch  # noqa: E402

from fplx import ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    n, d, h, w, c0, ncls = 2, 80, 160, 160, 32, 2
    dims, v = (n, d, h, w), n * d * h * w
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    bf = torch.bfloat16
    feat = torch.randn(v, c0, device=dev, generator=g).to(bf)
    dfeat = torch.empty(v, c0, device=dev, dtype=bf)
    logits = torch.empty(n, ncls, d, h, w, device=dev)
    dl = torch.randn(n, ncls, d, h, w, device=dev, generator=g)
    wo = torch.randn(ncls, c0, 1, 3, 3, device=dev, generator=g) * 0.1
    wof, _ = ops.pack_conv_weight(wo, torch.float32, False)
    _, wob = ops.pack_conv_weight(wo, bf, True)
    bo = torch.zeros(ncls, device=dev)
    img = torch.randn(n, 1, d, h, w, device=dev, generator=g)
    ws_ = torch.randn(c0, 1, 3, 3, 3, device=dev, generator=g) * 0.1
    wsf, _ = ops.pack_conv_weight(ws_, bf, False)
    bs = torch.zeros(c0, device=dev)
    rows = ops.conv3d_stats_rows(dims, 1, c0, (3, 3, 3), ops.F32, ops.BF16)
    stats = torch.zeros((rows, 2, c0), device=dev)
    cl, pl = ops.cl_strides, ops.planar_strides
    tb = v * c0 * 2
    res = []
    res.append(("stem fwd", timeit(lambda: ops.conv3d_fwd(img, pl(1, d, h, w), ops.F32, wsf, bs, feat, cl(d, h, w, c0), ops.BF16,
                                                          dims, 1, c0, (3, 3, 3), stats)), tb + v * 4))
    wsb = torch.empty(max(ops.conv3d_wgrad_ws_bytes(dims, 1, c0, (3, 3, 3)), 16), dtype=torch.uint8, device=dev)
    dws, dbs = torch.empty_like(ws_), torch.empty(c0, device=dev)
    res.append(("stem wgrad", timeit(lambda: ops.conv3d_wgrad(img, pl(1, d, h, w), ops.F32, feat, cl(d, h, w, c0), ops.BF16, dws,
                                                              None, dims, 1, c0, (3, 3, 3), wsb)), tb + v * 4))
    res.append(("out_conv fwd", timeit(lambda: ops.conv3d_fwd(feat, cl(d, h, w, c0), ops.BF16, wof, bo, logits, pl(ncls, d, h, w),
                                                              ops.F32, dims, c0, ncls, (1, 3, 3), None)), tb + v * ncls * 4))
    res.append(("out_conv dgrad", timeit(lambda: ops.conv3d_fwd(dl, pl(ncls, d, h, w), ops.F32, wob, None, dfeat, cl(d, h, w, c0),
                                                                ops.BF16, dims, ncls, c0, (1, 3, 3), None)), tb + v * ncls * 4))
    wsb2 = torch.empty(max(ops.conv3d_wgrad_ws_bytes(dims, c0, ncls, (1, 3, 3)), 16), dtype=torch.uint8, device=dev)
    dwo, dbo = torch.empty_like(wo), torch.empty(ncls, device=dev)
    res.append(("out_conv wgrad", timeit(lambda: ops.conv3d_wgrad(feat, cl(d, h, w, c0), ops.BF16, dl, pl(ncls, d, h, w), ops.F32,
                                                                  dwo, None, dims, c0, ncls, (1, 3, 3), wsb2)), tb + v * ncls * 4))
    # ---- the level-0 BatchNorm passes of a 32-channel site beside them (what a fused edge kernel would have to beat)
    gamma, beta = torch.ones(c0, device=dev), torch.zeros(c0, device=dev)
    rm, rv, nbt = torch.zeros(c0, device=dev), torch.ones(c0, device=dev), torch.zeros(1, dtype=torch.long, device=dev)
    bnbuf = torch.empty((4, c0), device=dev)
    ops.bn_train_finalize(stats, rows, c0, v, gamma, beta, rm, rv, nbt, bnbuf)
    slope = torch.full((1,), 0.25, device=dev)
    act = torch.empty_like(feat)
    dout = torch.randn(v, c0, device=dev, generator=g).to(bf)
    dyb = torch.empty_like(feat)
    part = torch.empty(ops.num_partials(v) * (2 * c0 + 1), device=dev)
    coef = torch.empty((2, c0), device=dev)
    gg, gb, gs = torch.zeros(c0, device=dev), torch.zeros(c0, device=dev), torch.zeros(1, device=dev)
    res.append(("bn_act_fwd 32ch", timeit(lambda: ops.bn_act_fwd(feat, act, bnbuf, slope, 0.0, 0, 0, c0)), 2 * tb))
    res.append(("bn_act_bwd 32ch", timeit(lambda: ops.bn_act_bwd(feat, dout, dyb, bnbuf, slope, 0.0, 0, 0, c0, True, gg, gb, gs, part,
                                                                  coef)), 5 * tb))
    # ---- the stem site's backward tail: apply pass + weight gradient against the weight gradient that forms dy itself
    ops.bn_act_bwd(feat, dout, dyb, bnbuf, slope, 0.0, 0, 0, c0, True, gg, gb, gs, part, coef)
    res.append(("bn bwd reduce+fin", timeit(lambda: ops.bn_act_bwd(feat, dout, dyb, bnbuf, slope, 0.0, 0, 0, c0, True, gg, gb, gs, part,
                                                                    coef, apply=False)), 2 * tb))
    res.append(("stem wgrad + bn", timeit(lambda: ops.stem_wgrad_bn(img, feat, dout, bnbuf, slope, coef, dws, dims, 1, c0, wsb)),
                2 * tb + v * 4))
    # ---- out_conv fused with the last site's BatchNorm passes: forward, and the backward pair (reduce + finalize + apply) with
    # the row-segment kernel (default) and the tile kernel
    from fplx import _lib
    lg2 = torch.empty(n, ncls, d, h, w, device=dev)
    for knob in (1, 0):
        _lib.set_tuning("outconv_fwd_rows", knob)
        res.append(("outconv_fwd_bn %s" % ("rows" if knob else "tiles"),
                    timeit(lambda: ops.outconv_fwd_bn(feat, bnbuf, slope, act, wof, bo, lg2, dims, c0, ncls)), 2 * tb + v * ncls * 4))
    _lib.set_tuning("outconv_fwd_rows", 1)
    # the forward without the activation store, and out_conv's weight + bias gradient from the pre-BatchNorm tensor
    res.append(("outconv_fwd_bn rows, logits only",
                timeit(lambda: ops.outconv_fwd_bn(feat, bnbuf, slope, None, wof, bo, lg2, dims, c0, ncls)), tb + v * ncls * 4))
    wsb3 = torch.empty(max(1, ops.outconv_wgrad_bn_ws_bytes(dims, c0, ncls)), dtype=torch.uint8, device=dev)
    res.append(("out_conv wgrad from y", timeit(lambda: ops.outconv_wgrad_bn(feat, bnbuf, slope, dl, dwo, dbo, dims, c0, ncls, wsb3)),
                tb + v * ncls * 4))
    for knob in (1, 0):
        _lib.set_tuning("outconv_dgrad_rows", knob)
        prt = torch.empty(max(ops.num_partials(v), ops.outconv_bn_rows(dims, c0, ncls)) * (2 * c0 + 1), device=dev)
        res.append(("outconv bwd fused %s" % ("rows" if knob else "tiles"),
                    timeit(lambda: ops.outconv_dgrad_bn_bwd(dl, wob, feat, bnbuf, slope, True, gg, gb, gs, prt, coef, dyb, dims, c0, ncls)),
                    3 * tb + 2 * v * ncls * 4))
    _lib.set_tuning("outconv_dgrad_rows", 1)
    env = {k: v_ for k, v_ in os.environ.items() if k.startswith("FPLX_")}
    for name, us, nbytes in res:
        print("%-34s %8.1f us  %6.0f GB/s  %s" % (name, us, nbytes / us / 1e3, env))


if __name__ == "__main__":
    main()

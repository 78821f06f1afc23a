"""Round-3 A/B of the rewritten edge / transposed-convolution kernels at the benchmark's shapes, interleaved rounds in one
process (tuning knobs through fplx_set_tuning): stem forward (stem_rows), out_conv forward (outconv_t), transposed-
convolution data gradients of the two shallow levels (deconv_dgrad_rows).  Prints the median of 5 rounds x 20 launches."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fpl-plus_amd"))
import torch  # noqa: E402

from fplx import ops, _lib  # noqa: E402


def ab(name, knob, values, fn, nbytes):
    res = {v: [] for v in values}
    for rnd in range(5):
        for v in values:
            _lib.set_tuning(knob, v)
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) * 50.0)
    _lib.set_tuning(knob, values[0])
    print("%-34s" % name + "  ".join("%s=%d: %7.1f us %5.0f GB/s" % (knob, v, sorted(res[v])[2], nbytes / sorted(res[v])[2] / 1e3) for v in values),
          flush=True)


def main():
    dev, bf = torch.device("cuda:0"), torch.bfloat16
    n, d, h, w, c0, ncls = 2, 80, 160, 160, 32, 2
    dims, v = (n, d, h, w), n * d * h * w
    cl, pl = ops.cl_strides, ops.planar_strides
    feat = torch.randn(v, c0, device=dev).to(bf)
    img = torch.randn(n, 1, d, h, w, device=dev)
    wsf, _ = ops.pack_conv_weight(torch.randn(c0, 1, 3, 3, 3, device=dev) * 0.1, bf, False)
    bs = torch.zeros(c0, device=dev)
    rows = ops.conv3d_stats_rows(dims, 1, c0, (3, 3, 3), ops.F32, ops.BF16)
    stats = torch.zeros((rows, 2, c0), device=dev)
    out = torch.empty(v, c0, device=dev, dtype=bf)
    ab("stem fwd 1->32", "stem_rows", (1, 0), lambda: ops.conv3d_fwd(img, pl(1, d, h, w), ops.F32, wsf, bs, out, cl(d, h, w, c0), ops.BF16,
                                                                     dims, 1, c0, (3, 3, 3), stats), v * c0 * 2 + v * 4)
    wof, _ = ops.pack_conv_weight(torch.randn(ncls, c0, 1, 3, 3, device=dev) * 0.1, torch.float32, False)
    logits = torch.empty(n, ncls, d, h, w, device=dev)
    bo = torch.zeros(ncls, device=dev)
    ab("out_conv fwd 32->2", "outconv_t", (1, 0), lambda: ops.conv3d_fwd(feat, cl(d, h, w, c0), ops.BF16, wof, bo, logits, pl(ncls, d, h, w),
                                                                          ops.F32, dims, c0, ncls, (1, 3, 3), None), v * c0 * 2 + v * ncls * 4)
    for (dd, cin, cout) in (((2, 40, 80, 80), 64, 32), ((2, 20, 40, 40), 128, 64), ((2, 10, 20, 20), 256, 128), ((2, 5, 10, 10), 512, 256)):
        nn, d1, h1, w1 = dd
        v1 = nn * d1 * h1 * w1
        dy = torch.randn(v1 * 8, cout, device=dev).to(bf)
        _, wb = ops.pack_deconv_weight(torch.randn(cin, cout, 2, 2, 2, device=dev) * 0.1, bf)
        dx = torch.empty(v1, cin, device=dev, dtype=bf)
        ab("deconv dgrad %s %d<-%d" % (dd, cin, cout), "deconv_dgrad_rows", (1, 0), lambda: ops.deconv2_dgrad(dy, wb, dx, dd, cin, cout),
           v1 * 8 * cout * 2 + v1 * cin * 2)


def pool_ab():
    dev, bf = torch.device("cuda:0"), torch.bfloat16
    for (dims, c) in (((2, 80, 160, 160), 32), ((2, 40, 80, 80), 64), ((2, 20, 40, 40), 128)):
        n, d, h, w = dims
        v, vo = n * d * h * w, n * d * h * w // 8
        y = torch.randn(v, c, device=dev).to(bf)
        dsk = (torch.randn(v, c, device=dev) * 0.01).to(bf)
        dyp = (torch.randn(vo, c, device=dev) * 0.01).to(bf)
        bnbuf = torch.stack([torch.zeros(c), torch.ones(c), torch.ones(c), torch.zeros(c)]).to(dev)
        slope = torch.full((1,), 0.25, device=dev)
        a2, pooled, dx = torch.empty_like(y), torch.empty(vo, c, device=dev, dtype=bf), torch.empty_like(y)
        part = torch.empty((ops.num_partials(v), 2 * c + 1), device=dev)
        ab("bn_act_pool_fwd %s C=%d" % (dims, c), "pool_col", (1, 0), lambda: ops.bn_act_pool_fwd(y, a2, pooled, bnbuf, slope, dims, c, 2),
           v * c * 4 + vo * c * 2)
        ab("pool_bwd_bn_reduce %s C=%d" % (dims, c), "pool_col", (1, 0),
           lambda: ops.pool_bwd_bn_reduce(y, dyp, dsk, dx, bnbuf, slope, dims, c, part, 2), v * c * 6 + vo * c * 2)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pool":
        pool_ab()
        sys.exit(0)
    main()

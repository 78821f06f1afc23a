"""A/B of the one-launch deep-level sites (fplx_conv3d_site_fwd / _bwd, fplx_bn_act_bwd_site) against the launches they replace,
alone on the device: us per site at the benchmark's level-3 / level-4 shapes.  python tools/deep_site_bench.py"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fpl-plus_amd"))
from fplx import ops  # noqa: E402


def timed(fn, reps=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    bf, dt = torch.bfloat16, ops._DT[torch.bfloat16]
    for (n, d, h, w, cin, cout, p) in [(2, 10, 20, 20, 128, 256, 0.4), (2, 10, 20, 20, 256, 256, 0.4), (2, 10, 20, 20, 512, 256, 0.4),
                                       (2, 5, 10, 10, 256, 512, 0.5), (2, 5, 10, 10, 512, 512, 0.5)]:
        dims, v = (n, d, h, w), n * d * h * w
        x = torch.randn(v, cin, device="cuda").to(bf)
        wt = (torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05)
        wf, wb = ops.pack_conv_weight(wt, bf, want_wb=True)                # wb: [27][cin][cout] -> data gradient cout -> cin
        bias = torch.randn(cout, device="cuda")
        bn = torch.nn.BatchNorm3d(cout).cuda()
        slope = torch.full((1,), 0.25, device="cuda")
        y, a = torch.empty((v, cout), dtype=bf, device="cuda"), torch.empty((v, cout), dtype=bf, device="cuda")
        buf = torch.empty((4, cout), dtype=torch.float32, device="cuda")
        rows = ops.conv3d_stats_rows(dims, cin, cout, (3, 3, 3), dt, dt)
        stats = torch.empty((rows, 2, cout), dtype=torch.float32, device="cuda")
        xs, ys = ops.cl_strides(d, h, w, cin), ops.cl_strides(d, h, w, cout)

        def fwd_sep():
            ops.conv3d_fwd(x, xs, dt, wf, bias, y, ys, dt, dims, cin, cout, (3, 3, 3), stats)
            ops.bn_train_finalize(stats, rows, cout, v, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, buf)
            ops.bn_act_fwd(y, a, buf, slope, p, 1, 2, cout)

        def conv_only():
            ops.conv3d_fwd(x, xs, dt, wf, bias, y, ys, dt, dims, cin, cout, (3, 3, 3), stats)

        def fwd_one():
            ops.conv3d_site_fwd(x, wf, bias, dims, cin, cout, False, bn, buf, slope, p, 1, 2, y, a)
        t_conv, t_sep, t_one = timed(conv_only), timed(fwd_sep), timed(fwd_one)
        # backward of the site: BatchNorm stages alone, and behind the data gradient of a cout -> cout convolution
        dout = (torch.randn(v, cout, device="cuda") * 0.01).to(bf)
        dy = torch.empty_like(dout)
        g, b_, s_ = torch.zeros(cout, device="cuda"), torch.zeros(cout, device="cuda"), torch.zeros(1, device="cuda")
        part = torch.empty((ops.num_partials(v), 2 * cout + 1), dtype=torch.float32, device="cuda")
        coef = torch.empty((2, cout), dtype=torch.float32, device="cuda")
        t_b3 = timed(lambda: ops.bn_act_bwd(y, dout, dy, buf, slope, p, 1, 2, cout, True, g, b_, s_, part, coef))
        t_b1 = timed(lambda: ops.bn_act_bwd_site(y, dout, dy, buf, slope, p, 1, 2, cout, True, g, b_, s_))
        w2 = torch.randn(cout, cout, 3, 3, 3, device="cuda") * 0.05
        _, wb2 = ops.pack_conv_weight(w2, bf, want_wb=True)
        da = torch.empty_like(dout)

        def bwd_sep():
            ops.conv3d_fwd(dout, ys, dt, wb2, None, da, ys, dt, dims, cout, cout, (3, 3, 3), None)
            ops.bn_act_bwd(y, da, da, buf, slope, p, 1, 2, cout, True, g, b_, s_, part, coef)

        def dgrad_only():
            ops.conv3d_fwd(dout, ys, dt, wb2, None, da, ys, dt, dims, cout, cout, (3, 3, 3), None)
        t_d, t_ds, t_d1 = timed(dgrad_only), timed(bwd_sep), timed(
            lambda: ops.conv3d_site_bwd(dout, wb2, dims, cout, cout, False, y, buf, slope, p, 1, 2, True, g, b_, s_, da))
        print("%dx%dx%dx%d %3d->%3d  forward: conv+finish %.1f | + finalize + apply %.1f | one-launch site %.1f us    "
              "BN backward: 3 stages %.1f | one launch %.1f    dgrad(%d->%d) %.1f | + 3 stages %.1f | site_bwd %.1f"
              % (n, d, h, w, cin, cout, t_conv, t_sep, t_one, t_b3, t_b1, cout, cout, t_d, t_ds, t_d1))


if __name__ == "__main__":
    main()

"""conv_fwd_brick: correctness against torch (CPU fp32 on bf16-rounded operands) and timing, forward (+statistics) and
data gradient, through fplx.ops.  FPLX_BRICK=0 in the environment times the tile kernel on the same shapes.

    python tools/brick_check.py [check]"""
import os
import sys
import ctypes

sys.path.insert(0, os.path.join(os.getcwd(), "fpl-plus_amd"))
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from fplx import ops  # noqa: E402

bf = torch.bfloat16
dt = ops._DT[bf]
lib = ops._lib.lib()


def cl(t):
    return t.permute(0, 2, 3, 4, 1).contiguous().view(-1, t.shape[1])


def plan(n, d, h, w, cin, cout):
    """(kernel family, geometry, ksplit, rows) from the declared plan query (5 = FPLX_KERNEL_BRICK)"""
    kern, g, k, r = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib.fplx_conv3d_plan_query(n, d, h, w, cin, cout, 3, 3, 3, dt, dt, ctypes.byref(kern), ctypes.byref(g), ctypes.byref(k),
                               ctypes.byref(r))
    return kern.value, g.value, k.value, r.value


def check(shape, direct):
    n, cin, cout, d, h, w = shape
    g = torch.Generator().manual_seed(5)
    q = lambda t: t.bfloat16().float()
    x = q(torch.randn(n, cin, d, h, w, generator=g))
    wt = q(torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.05)
    b = torch.randn(cout, generator=g)
    yr = F.conv3d(x, wt, b, padding=1)
    dims = (n, d, h, w)
    xg = cl(x).to(bf).cuda()
    wf, _ = ops.pack_conv_weight(wt.cuda(), bf, want_wb=False)
    y = torch.full((xg.shape[0], cout), 7.0, dtype=bf, device="cuda")
    if direct:                 # geometry 0 forced on shapes the plan would leave to the tile kernel
        ops._lib.set_tuning("brick_geo", 0)
    try:
        assert plan(n, d, h, w, cin, cout)[0] == 5
        rows = ops.conv3d_stats_rows(dims, cin, cout, (3, 3, 3), dt, dt)
        stats = torch.zeros((rows, 2, cout), dtype=torch.float32, device="cuda")
        ops.conv3d_fwd(xg, ops.cl_strides(d, h, w, cin), dt, wf, b.cuda(), y, ops.cl_strides(d, h, w, cout), dt, dims, cin,
                       cout, (3, 3, 3), stats)
    finally:
        ops._lib.set_tuning("brick_geo", -1)
    torch.cuda.synchronize()
    got = y.float().cpu().view(n, d, h, w, cout).permute(0, 4, 1, 2, 3)
    scale = float(yr.abs().max())
    err = float((got - yr).abs().max()) / scale
    s = stats.sum(0).cpu()
    yf = cl(yr)
    e1 = float((s[0] - yf.sum(0)).abs().max()) / (scale * yf.shape[0] ** 0.5)
    e2 = float(((s[1] - (yf * yf).sum(0)) / (yf * yf).sum(0)).abs().max())
    print("shape %s direct=%d: max err %.4f of max, stats sum %.4f sumsq rel %.4f" % (shape, direct, err, e1, e2), flush=True)
    assert err < 2e-2 and e1 < 2e-2 and e2 < 8e-2


def timeit(shape, stats_on, reps=30):
    n, cin, cout, d, h, w = shape
    dims = (n, d, h, w)
    V = n * d * h * w
    xg = torch.randn(V, cin, device="cuda").to(bf)
    wf = (torch.randn(27, cout, cin, device="cuda") * 0.05).to(bf)
    y = torch.empty(V, cout, dtype=bf, device="cuda")
    rows = ops.conv3d_stats_rows(dims, cin, cout, (3, 3, 3), dt, dt)
    stats = torch.zeros((rows, 2, cout), dtype=torch.float32, device="cuda") if stats_on else None
    if "check" not in sys.argv and not stats_on:
        print("plan %s: kernel %d geo %d ksplit %d rows %d" % ((shape,) + plan(n, d, h, w, cin, cout)))
    f = lambda: ops.conv3d_fwd(xg, ops.cl_strides(d, h, w, cin), dt, wf, None, y, ops.cl_strides(d, h, w, cout), dt, dims,
                               cin, cout, (3, 3, 3), stats)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    fl = 2.0 * V * 27 * cin * cout
    print("time %s stats=%d brick_ok=%d: %.1f us  %.3f PFLOP/s (%.3f of 2.5)" %
          (shape, stats_on, int(plan(n, d, h, w, cin, cout)[0] == 5), us, fl / us / 1e9, fl / us / 1e9 / 2.5), flush=True)


if "check" in sys.argv:
    check((1, 64, 128, 4, 8, 8), True)            # one brick
    check((1, 64, 128, 6, 20, 20), True)          # ragged in d, h and w
    check((2, 96, 256, 5, 9, 17), True)           # two couts tiles, three chunks, ragged, n = 2
    check((2, 128, 128, 20, 40, 40), False)       # the benchmark's level-2 layer through the dispatcher
    check((1, 64, 64, 5, 10, 17), True)           # 64 output channels per block
    check((2, 96, 192, 4, 8, 16), True)
L1 = ((2, 64, 64, 40, 80, 80), (2, 128, 64, 40, 80, 80), (2, 64, 128, 40, 80, 80))
L2 = ((2, 128, 128, 20, 40, 40), (2, 128, 256, 20, 40, 40), (2, 256, 128, 20, 40, 40))
L3 = ((2, 256, 256, 10, 20, 20), (2, 256, 512, 10, 20, 20), (2, 512, 256, 10, 20, 20), (2, 512, 512, 5, 10, 10))
for shp in (L1 if "l1" in sys.argv else (L3 if "l3" in sys.argv else L2)):
    timeit(shp, True)
    timeit(shp, False)

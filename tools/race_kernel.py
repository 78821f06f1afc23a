"""Which main-stream kernel of the 2.5D backward changes its result when weight-gradient kernels run beside it on a
second stream?  (tools/race25.py narrowed the hazard to the level-1 up-block.)  Each candidate op runs alone (reference
bits) and then `reps` times while a side stream loops over level-0 weight-gradient launches; outputs compared bitwise."""
import os
import sys

sys.path.insert(0, os.path.join(os.getcwd(), "fpl-plus_amd"))
import torch  # noqa: E402
import fplx  # noqa: E402
from fplx import ops  # noqa: E402
from fplx._lib import BF16  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
bf = torch.bfloat16
g = torch.Generator(device="cpu").manual_seed(0)


def rnd(*shape, dtype=bf, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev)


# ---- side load: level-0 2D weight gradients of the shipped configuration (4 x 28 x 128 x 128)
d0 = (4, 28, 128, 128)
v0 = d0[0] * d0[1] * d0[2] * d0[3]
x0a, x0b, dy0 = rnd(v0, 32), rnd(v0, 32), rnd(v0, 32, scale=1e-3)
dw_cat = torch.empty((32, 64, 3, 3), device=dev)
dw_32 = torch.empty((32, 32, 3, 3), device=dev)
ws_need = max(ops.conv2d_wgrad_ws_bytes(d0, 64, 32), ops.conv3d_wgrad_ws_bytes(d0, 64, 32, (3, 3, 3)))
ws_side = torch.empty(int(ws_need), dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()


def side_load(n):
    with torch.cuda.stream(side):
        for _ in range(n):
            ops.conv3d_wgrad_cat2(x0a, x0b, dy0, dw_cat, d0, 64, 32, ws_side, True)
            ops.conv2d_wgrad(x0a, ops.cl_strides(*d0[1:], 32), BF16, dy0, ops.cl_strides(*d0[1:], 32), BF16, dw_32, None, d0, 32, 32,
                             ws_side)


# ---- candidates at level 1 (4 x 28 x 64 x 64)
d1 = (4, 28, 64, 64)
v1 = d1[0] * d1[1] * d1[2] * d1[3]
w64 = torch.randn(64, 64, 3, 3, generator=g).to(dev) * 0.05
w128 = torch.randn(64, 128, 3, 3, generator=g).to(dev) * 0.05        # Conv2d(128 -> 64): its dgrad maps 64 -> 128
_, wb64 = ops.pack_conv2d_weight(w64, bf, True)
_, wb128 = ops.pack_conv2d_weight(w128, bf, True)
wf64, _ = ops.pack_conv2d_weight(w64, bf, False)
dy1 = rnd(v1, 64, scale=1e-2)
y1 = rnd(v1, 64)
bnbuf = torch.stack([torch.zeros(64), torch.ones(64), torch.ones(64), torch.zeros(64)]).to(dev)
slope = torch.full((1,), 0.25, device=dev)
part = torch.empty((ops.num_partials(v1), 2 * 128 + 1), device=dev)
coef = torch.empty((2, 128), device=dev)
bias64 = torch.zeros(64, device=dev)


def op_dgrad64():
    out = torch.empty((v1, 64), dtype=bf, device=dev)
    ops.conv3d_fwd(dy1, ops.cl_strides(*d1[1:], 64), BF16, wb64, None, out, ops.cl_strides(*d1[1:], 64), BF16, d1, 64, 64, (3, 3, 3),
                   None, mid=True)
    return out


def op_dgrad128():
    out = torch.empty((v1, 128), dtype=bf, device=dev)
    ops.conv3d_fwd(dy1, ops.cl_strides(*d1[1:], 64), BF16, wb128, None, out, ops.cl_strides(*d1[1:], 128), BF16, d1, 64, 128,
                   (3, 3, 3), None, mid=True)
    return out


def op_fwd64_stats():
    out = torch.empty((v1, 64), dtype=bf, device=dev)
    rows = ops.conv3d_stats_rows(d1, 64, 64, (3, 3, 3), BF16, BF16, True)
    stats = torch.empty((rows, 2, 64), device=dev)
    ops.conv3d_fwd(y1, ops.cl_strides(*d1[1:], 64), BF16, wf64, bias64, out, ops.cl_strides(*d1[1:], 64), BF16, d1, 64, 64, (3, 3, 3),
                   stats, mid=True)
    return torch.cat([out.float().reshape(-1), stats.reshape(-1)])


def op_bn_bwd():
    d = dy1.clone()
    dg, db, ds = torch.zeros(64, device=dev), torch.zeros(64, device=dev), torch.zeros(1, device=dev)
    ops.bn_act_bwd(y1, d, d, bnbuf, slope, 0.0, 1, 0, 64, True, dg, db, ds, part, coef)
    return torch.cat([d.float().reshape(-1), dg, db, ds])


def op_pool_bwd():
    dp = rnd(v1 // 4, 64, scale=1e-2) if not hasattr(op_pool_bwd, "dp") else op_pool_bwd.dp
    op_pool_bwd.dp = dp
    dx = torch.empty((v1, 64), dtype=bf, device=dev)
    ops.maxpool2_bwd(y1, dp, dy1, dx, d1, 64, 1)
    return dx


def op_wgrad64():
    dw = torch.empty((64, 64, 3, 3), device=dev)
    need = ops.conv2d_wgrad_ws_bytes(d1, 64, 64)
    ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
    ops.conv2d_wgrad(y1, ops.cl_strides(*d1[1:], 64), BF16, dy1, ops.cl_strides(*d1[1:], 64), BF16, dw, None, d1, 64, 64, ws)
    return dw


for name, fn in (("dgrad 64->64 (march64 2D)", op_dgrad64), ("dgrad 64->128 (march64 2D)", op_dgrad128),
                 ("fwd 64->64 + stats (march64 2D)", op_fwd64_stats), ("bn_act_bwd C=64", op_bn_bwd),
                 ("maxpool122_bwd", op_pool_bwd), ("conv2d_wgrad 64->64", op_wgrad64)):
    ref = fn()
    torch.cuda.synchronize()
    solo_bad = 0
    for _ in range(5):
        solo_bad += int(not torch.equal(fn(), ref))
    torch.cuda.synchronize()
    bad, worst, nel = 0, 0.0, 0
    for _ in range(reps):
        side_load(3)
        out = fn()
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            dd = (out.float() - ref.float()).abs()
            worst = max(worst, float(dd.max()))
            nel = max(nel, int((dd > 0).sum()))
    print("%-34s alone: %d/5 differ; beside the side stream: %d/%d differ (max |diff| %.3e, up to %d elements)" % (
        name, solo_bad, bad, reps, worst, nel))

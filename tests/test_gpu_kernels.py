"""Kernel-level GPU tests through the C ABI: generic convolution / transposed convolution /
pooling / Adam against plain PyTorch fp32 on the CPU, incl. ragged and strided (channel-slice) cases."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import detdata
from util import plan_kernel

pytestmark = pytest.mark.gpu


def cl(t):       # NCDHW -> [V, C] channels-last 2-D
    return t.permute(0, 2, 3, 4, 1).contiguous().view(-1, t.shape[1])


def uncl(t2, n, d, h, w):
    return t2.view(n, d, h, w, -1).permute(0, 4, 1, 2, 3)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("shape", [(1, 5, 7, 3, 9, 6), (2, 16, 8, 4, 6, 10), (1, 3, 11, 2, 2, 2),
                                   # Cin % 32 == 0 and Cout % 32 == 0: the bf16 runs take the MFMA kernels
                                   (1, 32, 32, 5, 9, 35), (2, 64, 32, 3, 8, 33), (1, 32, 64, 4, 11, 17),
                                   (1, 128, 64, 2, 5, 9), (1, 32, 32, 9, 16, 64),
                                   # large H x W with Cin in {32, 64}: the streaming LDS kernel (ragged tiles included)
                                   (1, 32, 32, 20, 24, 96), (2, 64, 64, 9, 18, 70), (1, 64, 32, 4, 16, 64),
                                   (2, 32, 64, 7, 20, 70),
                                   # depth-marching kernels: several depth segments per tile, ragged footprints, n = 3,
                                   # Cout = 96, short volumes; (1, 64, 64, 20, 40, 40) also takes the two-ci-tile wgrad
                                   (1, 32, 32, 40, 32, 64), (3, 32, 96, 6, 17, 65), (1, 64, 32, 37, 24, 64),
                                   (2, 64, 96, 5, 9, 70), (1, 64, 64, 20, 40, 40),
                                   # tiny volumes, wide channels: split-K over the taps (27 / 9 / 3-way)
                                   (2, 128, 128, 2, 5, 5), (2, 64, 256, 5, 10, 10), (1, 128, 64, 10, 20, 20),
                                   # a larger tile-kernel case with a ragged last tile
                                   (1, 64, 256, 8, 60, 61)])
def test_conv3d_fwd_wgrad_dgrad_generic(dtype, tol, shape):
    from fplx import ops
    n, cin, cout, d, h, w = shape
    x = torch.from_numpy(detdata.normal("k.x%s" % (shape,), (n, cin, d, h, w)))
    wt = torch.from_numpy(detdata.normal("k.w%s" % (shape,), (cout, cin, 3, 3, 3), 0.2))
    b = torch.from_numpy(detdata.normal("k.b%s" % (shape,), (cout,)))
    dy = torch.from_numpy(detdata.normal("k.dy%s" % (shape,), (n, cout, d, h, w)))
    if dtype == torch.bfloat16:
        x, wt, dy = x.bfloat16().float(), wt.bfloat16().float(), dy.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, br, padding=1)
    yr.backward(dy)
    dt = ops._DT[dtype]
    dims = (n, d, h, w)
    xg, dyg = cl(x).to(dtype).cuda(), cl(dy).to(dtype).cuda()
    wf, wb = ops.pack_conv_weight(wt.cuda(), dtype)
    # output into a channel slice of a wider buffer (ld = cout + 3)
    ybuf = torch.zeros((xg.shape[0], cout + 3), dtype=dtype, device="cuda")
    yv = ybuf[:, 2:2 + cout]
    rows = ops.conv3d_stats_rows(dims, cin, cout, (3, 3, 3), dt, dt)
    stats = torch.zeros((rows, 2, cout), dtype=torch.float32, device="cuda")
    ops.conv3d_fwd(xg, ops.cl_strides(d, h, w, cin), dt, wf, b.cuda(), yv, ops.cl_strides(d, h, w, cout + 3), dt,
                   dims, cin, cout, (3, 3, 3), stats)
    got = uncl(yv.float().cpu(), n, d, h, w)
    scale = float(yr.abs().max())
    assert float((got - yr.detach()).abs().max()) < tol * scale
    assert float(ybuf[:, :2].abs().max()) == 0 and float(ybuf[:, 2 + cout:].abs().max()) == 0
    s = stats.sum(0).cpu()
    yf = cl(yr.detach())
    np.testing.assert_allclose(s[0].numpy(), yf.sum(0).numpy(), atol=tol * scale * yf.shape[0] ** 0.5 + 1e-3)
    np.testing.assert_allclose(s[1].numpy(), (yf * yf).sum(0).numpy(), rtol=max(tol, 1e-4) * 4)
    # data gradient = same kernel with the mirrored pack
    dx = torch.empty((xg.shape[0], cin), dtype=dtype, device="cuda")
    ops.conv3d_fwd(dyg, ops.cl_strides(d, h, w, cout), dt, wb, None, dx, ops.cl_strides(d, h, w, cin), dt, dims,
                   cout, cin, (3, 3, 3), None)
    assert float((uncl(dx.float().cpu(), n, d, h, w) - xr.grad).abs().max()) < tol * float(xr.grad.abs().max())
    # weight / bias gradient
    ws = torch.empty(ops.conv3d_wgrad_ws_bytes(dims, cin, cout, (3, 3, 3)), dtype=torch.uint8, device="cuda")
    dw = torch.empty((cout, cin, 3, 3, 3), dtype=torch.float32, device="cuda")
    db = torch.empty(cout, dtype=torch.float32, device="cuda")
    ops.conv3d_wgrad(xg, ops.cl_strides(d, h, w, cin), dt, dyg, ops.cl_strides(d, h, w, cout), dt, dw, db, dims,
                     cin, cout, (3, 3, 3), ws)
    assert float((dw.cpu() - wr.grad).abs().max()) < max(tol, 1e-4) * float(wr.grad.abs().max())
    assert float((db.cpu() - br.grad).abs().max()) < max(tol, 1e-4) * float(br.grad.abs().max())
    with pytest.raises(RuntimeError):
        ops.conv3d_wgrad(xg, ops.cl_strides(d, h, w, cin), dt, dyg, ops.cl_strides(d, h, w, cout), dt, dw, db, dims,
                         cin, cout, (3, 3, 3), ws[:16])
    with pytest.raises(ValueError):
        ops.conv3d_fwd(xg, ops.cl_strides(d, h, w, cin), dt, wf, None, yv, ops.cl_strides(d, h, w, cout + 3), dt,
                       (0, d, h, w), cin, cout, (3, 3, 3), None)


@pytest.mark.parametrize("shape", [
    # Cin = 64, 16 x 16 footprint (less padded area than 8 x 32): exact fit, ragged in h and w, several depth segments
    (1, 64, 64, 6, 16, 80), (2, 64, 32, 5, 9, 70), (1, 64, 96, 21, 48, 80), (1, 64, 32, 4, 17, 65),
    # Cin = 64, 8 x 32 footprint; Cin = 32
    (1, 64, 32, 7, 24, 64), (1, 32, 64, 9, 20, 96),
    # Cin = 128: the streamed-weight form (four channel quarters per slab), both footprints, ragged, depth segments
    (1, 128, 64, 6, 16, 80), (2, 128, 32, 5, 9, 70), (1, 128, 96, 21, 24, 64), (1, 128, 64, 4, 17, 65),
    # Cin = 32, footprints inside the volume (H % 16 = W % 32 = 0): conv_fwd_march32v2<true>, the forward + statistics kernel
    # of the benchmark's level 0 (2 of its 4 dominant launches per step); one / two co blocks, several depth segments
    (1, 32, 32, 9, 16, 64), (2, 32, 64, 12, 32, 96), (2, 32, 32, 40, 48, 64)])
def test_conv3d_march_kernels_aligned_output(shape):
    """the depth-marching kernels need 16-byte aligned rows (the generic test writes into an odd channel slice and so
    exercises them only through the data gradient): forward + BN statistics on a dense output, bf16, against torch"""
    from fplx import ops
    n, cin, cout, d, h, w = shape
    q = lambda t: t.bfloat16().float()
    x = q(torch.from_numpy(detdata.normal("m.x%s" % (shape,), (n, cin, d, h, w))))
    wt = q(torch.from_numpy(detdata.normal("m.w%s" % (shape,), (cout, cin, 3, 3, 3), 0.2)))
    b = torch.from_numpy(detdata.normal("m.b%s" % (shape,), (cout,)))
    yr = F.conv3d(x, wt, b, padding=1)
    bf, dt, dims = torch.bfloat16, ops._DT[torch.bfloat16], (n, d, h, w)
    assert plan_kernel(n, d, h, w, cin, cout) == 4                # FPLX_KERNEL_MARCH
    if cin == 32:       # the one-wave-per-SIMD kernel (v2 with statistics) exactly on the footprints inside the volume
        assert plan_kernel(n, d, h, w, cin, cout, full=True)[1] == (2 if h % 16 == 0 and w % 32 == 0 else 0)
    xg = cl(x).to(bf).cuda()
    wf, _ = ops.pack_conv_weight(wt.cuda(), bf, want_wb=False)
    y = torch.full((xg.shape[0], cout), 7.0, dtype=bf, device="cuda")
    rows = ops.conv3d_stats_rows(dims, cin, cout, (3, 3, 3), dt, dt)
    stats = torch.zeros((rows, 2, cout), dtype=torch.float32, device="cuda")
    ops.conv3d_fwd(xg, ops.cl_strides(d, h, w, cin), dt, wf, b.cuda(), y, ops.cl_strides(d, h, w, cout), dt, dims, cin,
                   cout, (3, 3, 3), stats)
    scale = float(yr.abs().max())
    assert float((uncl(y.float().cpu(), n, d, h, w) - yr).abs().max()) < 2e-2 * scale
    s = stats.sum(0).cpu()
    yf = cl(yr)
    np.testing.assert_allclose(s[0].numpy(), yf.sum(0).numpy(), atol=2e-2 * scale * yf.shape[0] ** 0.5 + 1e-3)
    np.testing.assert_allclose(s[1].numpy(), (yf * yf).sum(0).numpy(), rtol=8e-2)
    if cin == 32 and h % 16 == 0 and w % 32 == 0:
        # the statistics-free form of the same layer (conv_fwd_march32v3<false>, what a data gradient runs) against torch too
        y3 = torch.full((xg.shape[0], cout), 7.0, dtype=bf, device="cuda")
        ops.conv3d_fwd(xg, ops.cl_strides(d, h, w, cin), dt, wf, b.cuda(), y3, ops.cl_strides(d, h, w, cout), dt, dims, cin, cout,
                       (3, 3, 3), None)
        assert float((uncl(y3.float().cpu(), n, d, h, w) - yr).abs().max()) < 2e-2 * scale


@pytest.mark.parametrize("shape,geo,ks", [
    ((1, 64, 128, 4, 8, 8), 0, 1),            # one 4 x 8 x 8 brick
    ((1, 64, 128, 6, 20, 20), 0, 1),          # ragged in d, h and w
    ((2, 96, 256, 5, 9, 17), 0, 1),           # two 128-channel blocks, three channel chunks, ragged, n = 2
    ((1, 64, 64, 5, 10, 17), 0, 1),           # 64 output channels per block
    ((2, 96, 192, 4, 8, 16), 0, 1),
    ((1, 64, 128, 10, 20, 20), 1, 1),         # 5 x 4 x 8 bricks: level 3's geometry, ragged in w only
    ((2, 96, 256, 7, 9, 13), 1, 1),           # ... ragged everywhere
    ((1, 128, 128, 10, 20, 20), 1, 2),        # Cin split over blockIdx.z: fp32 partials
    ((1, 160, 64, 6, 12, 16), 0, 3),          # ... 5 chunks over 3 splits (2 / 1 / 2), 4 x 8 x 8 bricks
    ((2, 128, 128, 20, 40, 40), -1, 0),       # through the dispatcher: a level-2 layer of the benchmark
    ((2, 128, 64, 12, 32, 64), -1, 0),        # ... a layer a march kernel would take as well
    ((2, 256, 256, 10, 20, 20), -1, 0),       # ... a level-3 layer: 5 x 4 x 8 bricks, split-K + finish
    ((2, 512, 512, 5, 10, 10), -1, 0),        # ... level 4 (round 4: bricks padded 1.9x, 4-way Cin split) and its data gradient's shape
    ((2, 512, 256, 5, 10, 10), -1, 0)])
def test_conv3d_brick_kernel(shape, geo, ks):
    """conv_fwd_brick (input-stationary bricks, conv_brick.hip): forward + BN statistics against torch, bf16;
    geo >= 0: the tuning knobs "brick_geo" / "brick_ksplit" force that geometry / Cin split (+ its split-K finish) on shapes
    the dispatcher leaves to other kernels (few / ragged bricks)"""
    from fplx import ops
    _lib = ops._lib
    n, cin, cout, d, h, w = shape
    q = lambda t: t.bfloat16().float()
    x = q(torch.from_numpy(detdata.normal("b.x%s" % (shape,), (n, cin, d, h, w))))
    wt = q(torch.from_numpy(detdata.normal("b.w%s" % (shape,), (cout, cin, 3, 3, 3), 0.05)))
    b = torch.from_numpy(detdata.normal("b.b%s" % (shape,), (cout,)))
    yr = F.conv3d(x, wt, b, padding=1)
    bf, dt, dims = torch.bfloat16, ops._DT[torch.bfloat16], (n, d, h, w)
    xg = cl(x).to(bf).cuda()
    wf, _ = ops.pack_conv_weight(wt.cuda(), bf, want_wb=False)
    y = torch.full((xg.shape[0], cout), 7.0, dtype=bf, device="cuda")
    bg = b.cuda()
    scale = float(yr.abs().max())
    if geo >= 0:
        # operands as channel slices of wider buffers (a concat half as input, a concat half as output): ld > channels
        xw = torch.full((xg.shape[0], cin + 64), 9.0, dtype=bf, device="cuda")
        xw[:, 32:32 + cin] = xg
        xg = xw[:, 32:32 + cin]
        yw = torch.full((xg.shape[0], cout + 32), 7.0, dtype=bf, device="cuda")
        y = yw[:, 8:8 + cout]
        _lib.set_tuning("brick_geo", geo)
        _lib.set_tuning("brick_ksplit", ks)
        try:
            assert plan_kernel(n, d, h, w, cin, cout, full=True)[:3] == (5, geo, ks)      # FPLX_KERNEL_BRICK, as forced
            rows = ops.conv3d_stats_rows(dims, cin, cout, (3, 3, 3), dt, dt)
            stats = torch.zeros((rows, 2, cout), dtype=torch.float32, device="cuda")
            ops.conv3d_fwd(xg, ops.cl_strides(d, h, w, cin + 64), dt, wf, bg, y, ops.cl_strides(d, h, w, cout + 32), dt, dims,
                           cin, cout, (3, 3, 3), stats)
        finally:
            _lib.set_tuning("brick_geo", -1)
            _lib.set_tuning("brick_ksplit", 0)
        assert float(yw[:, :8].float().abs().max()) == 7.0 and float(yw[:, 8 + cout:].float().min()) == 7.0   # nothing beside the slice
    else:
        assert plan_kernel(n, d, h, w, cin, cout) == 5
        rows = ops.conv3d_stats_rows(dims, cin, cout, (3, 3, 3), dt, dt)
        stats = torch.zeros((rows, 2, cout), dtype=torch.float32, device="cuda")
        ops.conv3d_fwd(xg, ops.cl_strides(d, h, w, cin), dt, wf, bg, y, ops.cl_strides(d, h, w, cout), dt, dims, cin, cout,
                       (3, 3, 3), stats)
    assert float((uncl(y.float().contiguous().cpu(), n, d, h, w) - yr).abs().max()) < 2e-2 * scale
    s = stats.sum(0).cpu()
    yf = cl(yr)
    np.testing.assert_allclose(s[0].numpy(), yf.sum(0).numpy(), atol=2e-2 * scale * yf.shape[0] ** 0.5 + 1e-3)
    np.testing.assert_allclose(s[1].numpy(), (yf * yf).sum(0).numpy(), rtol=8e-2)
    if geo < 0 and plan_kernel(n, d, h, w, cout, cin) == 5:
        # the data gradient is the same kernel on the mirrored pack (no bias, no statistics)
        dy = q(torch.from_numpy(detdata.normal("b.dy%s" % (shape,), (n, cout, d, h, w))))
        xr = x.clone().requires_grad_(True)
        F.conv3d(xr, wt, None, padding=1).backward(dy)
        _, wb = ops.pack_conv_weight(wt.cuda(), bf, want_wb=True)
        dx = torch.full((xg.shape[0], cin), 7.0, dtype=bf, device="cuda")
        ops.conv3d_fwd(cl(dy).to(bf).cuda(), ops.cl_strides(d, h, w, cout), dt, wb, None, dx, ops.cl_strides(d, h, w, cin), dt,
                       dims, cout, cin, (3, 3, 3), None)
        assert float((uncl(dx.float().cpu(), n, d, h, w) - xr.grad).abs().max()) < 2e-2 * float(xr.grad.abs().max())


@pytest.mark.parametrize("shape", [(1, 20, 40, 64), (2, 21, 24, 70),
                                   # H % 16 = W % 32 = 0: the split data gradient is conv_fwd_march32v3 with two output tensors (y1)
                                   (1, 16, 32, 64), (2, 12, 48, 96)])
def test_conv3d_cat2_split_concat(shape):
    """conv3x3x3 on cat([x0, x1], channel) with the concatenation never built (reference unet2d5_dsbn.py:182-183):
    forward + statistics, data gradient into two tensors, weight gradient - against torch on the explicit cat"""
    from fplx import ops
    n, d, h, w = shape
    cin, cout, tol = 64, 32, 2e-2
    dims = (n, d, h, w)
    assert ops.conv3d_cat2_ok(dims, cin, cout)
    assert not ops.conv3d_cat2_ok((1, 4, 16, 32), cin, cout) and not ops.conv3d_cat2_ok(dims, 128, 64)
    q = lambda t: t.bfloat16().float()
    x0 = q(torch.from_numpy(detdata.normal("c2.x0%s" % (shape,), (n, 32, d, h, w))))
    x1 = q(torch.from_numpy(detdata.normal("c2.x1%s" % (shape,), (n, 32, d, h, w))))
    wt = q(torch.from_numpy(detdata.normal("c2.w%s" % (shape,), (cout, cin, 3, 3, 3), 0.2)))
    b = torch.from_numpy(detdata.normal("c2.b%s" % (shape,), (cout,)))
    dy = q(torch.from_numpy(detdata.normal("c2.dy%s" % (shape,), (n, cout, d, h, w))))
    xr = torch.cat([x0, x1], 1).requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, b, padding=1)
    yr.backward(dy)
    bf, dt = torch.bfloat16, ops._DT[torch.bfloat16]
    g0, g1, dyg = cl(x0).to(bf).cuda(), cl(x1).to(bf).cuda(), cl(dy).to(bf).cuda()
    wf, wb = ops.pack_conv_weight(wt.cuda(), bf)
    y = torch.empty((g0.shape[0], cout), dtype=bf, device="cuda")
    rows = ops.conv3d_stats_rows(dims, cin, cout, (3, 3, 3), dt, dt)
    stats = torch.zeros((rows, 2, cout), dtype=torch.float32, device="cuda")
    ops.conv3d_fwd_cat2(g0, g1, wf, b.cuda(), y, dims, cin, cout, stats)
    scale = float(yr.detach().abs().max())
    assert float((uncl(y.float().cpu(), n, d, h, w) - yr.detach()).abs().max()) < tol * scale
    yf = cl(yr.detach())
    sm = stats.sum(0).cpu()
    np.testing.assert_allclose(sm[0].numpy(), yf.sum(0).numpy(), atol=tol * scale * yf.shape[0] ** 0.5 + 1e-3)
    np.testing.assert_allclose(sm[1].numpy(), (yf * yf).sum(0).numpy(), rtol=tol * 4)
    # the same kernel on the materialised concatenation gives the same bits
    cat = torch.cat([g0, g1], 1).contiguous()
    y2 = torch.empty_like(y)
    ops.conv3d_fwd(cat, ops.cl_strides(d, h, w, cin), dt, wf, b.cuda(), y2, ops.cl_strides(d, h, w, cout), dt, dims, cin,
                   cout, (3, 3, 3), None)
    assert torch.equal(y, y2)
    dx0 = torch.empty_like(g0)
    dx1 = torch.empty_like(g1)
    assert plan_kernel(n, d, h, w, cout, cin, full=True)[:2] == (4, 2 if h % 16 == 0 and w % 32 == 0 else 0)
    ops.conv3d_dgrad_split2(dyg, wb, dx0, dx1, dims, cin, cout)
    gmax = float(xr.grad.abs().max())
    assert float((uncl(dx0.float().cpu(), n, d, h, w) - xr.grad[:, :32]).abs().max()) < tol * gmax
    assert float((uncl(dx1.float().cpu(), n, d, h, w) - xr.grad[:, 32:]).abs().max()) < tol * gmax
    ws = torch.empty(ops.conv3d_wgrad_ws_bytes(dims, cin, cout, (3, 3, 3)), dtype=torch.uint8, device="cuda")
    dw = torch.empty((cout, cin, 3, 3, 3), dtype=torch.float32, device="cuda")
    ops.conv3d_wgrad_cat2(g0, g1, dyg, dw, dims, cin, cout, ws)
    assert float((dw.cpu() - wr.grad).abs().max()) < tol * float(wr.grad.abs().max())
    with pytest.raises(ValueError):
        ops.conv3d_fwd_cat2(g0, g1, wf, None, y, (1, 4, 16, 32), cin, cout, None)
    with pytest.raises(RuntimeError):
        ops.conv3d_wgrad_cat2(g0, g1, dyg, dw, dims, cin, cout, ws[:16])


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("shape", [(2, 12, 7, 3, 4, 5), (2, 64, 32, 3, 4, 5), (1, 128, 64, 2, 5, 9), (1, 32, 32, 4, 4, 7),
                                   (1, 96, 32, 2, 3, 67),
                                   # >= 32768 input voxels, Cin 64 | 128, Cout 32 | 64: the streaming forward
                                   # (deconv_fwd_rows), ragged last tile, two samples; the pipelined data gradient
                                   (2, 64, 32, 9, 37, 50), (1, 128, 64, 8, 65, 63)])
def test_deconv2_fwd_bwd(dtype, tol, shape):
    from fplx import ops
    n, cin, cout, d, h, w = shape
    x = torch.from_numpy(detdata.normal("dc.x%s" % (shape,), (n, cin, d, h, w)))
    wt = torch.from_numpy(detdata.normal("dc.w%s" % (shape,), (cin, cout, 2, 2, 2), 0.3))
    b = torch.from_numpy(detdata.normal("dc.b%s" % (shape,), (cout,)))
    dy = torch.from_numpy(detdata.normal("dc.dy%s" % (shape,), (n, cout, 2 * d, 2 * h, 2 * w)))
    if dtype == torch.bfloat16:
        x, wt, dy = x.bfloat16().float(), wt.bfloat16().float(), dy.bfloat16().float()
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv_transpose3d(xr, wr, br, stride=2)
    yr.backward(dy)
    dims = (n, d, h, w)
    xg, dyg = cl(x).to(dtype).cuda(), cl(dy).to(dtype).cuda()
    wf, wb = ops.pack_deconv_weight(wt.cuda(), dtype)
    cat = torch.zeros((dyg.shape[0], 2 * cout), dtype=dtype, device="cuda")     # write the "up" half of a concat
    ops.deconv2_fwd(xg, wf, b.cuda(), cat[:, cout:], dims, cin, cout)
    got = uncl(cat[:, cout:].float().cpu(), n, 2 * d, 2 * h, 2 * w)
    assert float((got - yr.detach()).abs().max()) < tol * float(yr.abs().max())
    assert float(cat[:, :cout].abs().max()) == 0
    dx = torch.empty_like(xg)
    ops.deconv2_dgrad(dyg, wb, dx, dims, cin, cout)
    assert float((uncl(dx.float().cpu(), n, d, h, w) - xr.grad).abs().max()) < tol * float(xr.grad.abs().max())
    ws = torch.empty(ops.deconv2_wgrad_ws_bytes(dims, cin, cout), dtype=torch.uint8, device="cuda")
    dw = torch.empty((cin, cout, 2, 2, 2), dtype=torch.float32, device="cuda")
    db = torch.empty(cout, dtype=torch.float32, device="cuda")
    ops.deconv2_wgrad(xg, dyg, dw, db, dims, cin, cout, ws)
    assert float((dw.cpu() - wr.grad).abs().max()) < max(tol, 1e-4) * float(wr.grad.abs().max())
    assert float((db.cpu() - br.grad).abs().max()) < max(tol, 1e-4) * float(br.grad.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c", [8, 5])
def test_maxpool_fwd_bwd_first_max_wins(dtype, c):
    from fplx import ops
    n, d, h, w = 2, 4, 6, 8
    x = torch.from_numpy(detdata.normal("mp.x%d" % c, (n, c, d, h, w)))
    x = (x * 2).round() / 2            # many exact ties
    dy = torch.from_numpy(detdata.normal("mp.dy%d" % c, (n, c, d // 2, h // 2, w // 2)))
    skip = torch.from_numpy(detdata.normal("mp.sk%d" % c, (n, c, d, h, w)))
    if dtype == torch.bfloat16:
        dy, skip = dy.bfloat16().float(), skip.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool3d(xr, 2, 2)
    yr.backward(dy)
    xg = cl(x).to(dtype).cuda()
    yg = torch.empty((n * d * h * w // 8, c), dtype=dtype, device="cuda")
    ops.maxpool2_fwd(xg, yg, (n, d, h, w), c)
    assert torch.equal(uncl(yg.float().cpu(), n, d // 2, h // 2, w // 2), yr.detach())
    dx = torch.empty_like(xg)
    ops.maxpool2_bwd(xg, cl(dy).to(dtype).cuda(), cl(skip).to(dtype).cuda(), dx, (n, d, h, w), c)
    ref = (xr.grad + skip)
    if dtype == torch.bfloat16:
        ref = ref.bfloat16().float()
    assert float((uncl(dx.float().cpu(), n, d, h, w) - ref).abs().max()) <= (0 if dtype == torch.float32 else 1e-2)
    with pytest.raises(ValueError):
        ops.maxpool2_fwd(xg, yg, (n, 3, h, w), c)


def test_fused_adam_matches_torch():
    from fplx import ops
    torch.manual_seed(3)
    p0 = torch.randn(10007)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], 1e-3, weight_decay=1e-5)
    p, m, v = p0.clone().cuda(), torch.zeros(10007).cuda(), torch.zeros(10007).cuda()
    for step in range(1, 6):
        g = torch.randn(10007)
        p_ref.grad = g.clone()
        opt.step()
        ops.adam_step(p, g.cuda(), m, v, 1e-3, step, 1e-5)
        assert float((p.cpu() - p_ref.detach()).abs().max()) < 2e-6
    with pytest.raises(ValueError):
        ops.adam_step(p, g.cuda(), m, v, 1e-3, 0, 1e-5)


def test_pack_weights_multi_equals_the_single_packs():
    """fplx_pack_weights_multi (the transposed-convolution and out_conv packs of a step in one launch) against
    fplx_pack_deconv_weight / fplx_pack_deconv122_weight / fplx_pack_conv_weight: bit for bit, incl. the fp32 out_conv pack, a
    job with only the data-gradient layout, a ConvTranspose2d weight (4 taps) and odd channel counts."""
    from fplx import ops
    g = torch.Generator().manual_seed(4)
    bf = torch.bfloat16
    wd3 = torch.randn(24, 40, 2, 2, 2, generator=g).cuda()
    wd2 = torch.randn(16, 8, 2, 2, generator=g).cuda()
    wo = torch.randn(3, 32, 1, 3, 3, generator=g).cuda()
    ws = torch.randn(32, 1, 3, 3, 3, generator=g).cuda()
    ref3, ref2 = ops.pack_deconv_weight(wd3, bf), ops.pack_deconv_weight(wd2, bf)
    ref_of, _ = ops.pack_conv_weight(wo, torch.float32, False)
    _, ref_ob = ops.pack_conv_weight(wo, bf, True)
    ref_s, _ = ops.pack_conv_weight(ws, bf, False)
    e = lambda *shape, dt=bf: torch.full(shape, 7.0, dtype=dt, device="cuda")
    o3, o2 = (e(8, 40, 24), e(8, 24, 40)), (e(4, 8, 16), e(4, 16, 8))
    of, ob, os_ = e(9, 3, 32, dt=torch.float32), e(9, 32, 3), e(27, 32, 1)
    ops.pack_weights_multi([(1, wd3, o3[0], o3[1], 24, 40, 8), (1, wd2, o2[0], o2[1], 16, 8, 4), (0, wo, of, None, 3, 32, 9),
                            (0, wo, None, ob, 3, 32, 9), (0, ws, os_, None, 32, 1, 27)])
    assert torch.equal(o3[0], ref3[0]) and torch.equal(o3[1], ref3[1]) and torch.equal(o2[0], ref2[0]) and torch.equal(o2[1], ref2[1])
    assert torch.equal(of, ref_of) and torch.equal(ob, ref_ob) and torch.equal(os_, ref_s)
    with pytest.raises(ValueError):
        ops.pack_weights_multi([(2, wo, of, None, 3, 32, 9)])


def test_fused_adam_pack_step_kernel():
    """fplx_adam_pack_step (Adam over a flat segment + the bf16 packs of the 3x3x3 weights inside it, one launch) against
    fplx_adam_step followed by fplx_pack_conv_weights_batched: parameters, both moments and both packs bit for bit, over three
    steps; gaps in front of, between and behind the layers (odd lengths), a layer without a data-gradient pack."""
    from fplx import ops
    g = torch.Generator().manual_seed(21)
    shapes = [(32, 32), (16, 64), (64, 96)]
    gaps = [8, 1028, 36, 4097]
    offs, pos = [], 0
    for (co, ci), gp in zip(shapes, gaps):
        pos += (gp + 3) // 4 * 4
        offs.append(pos)
        pos += co * ci * 27
    n = pos + gaps[-1]
    p0 = torch.randn(n, generator=g) * 0.1
    res = {}
    for fused in (True, False):
        p, m, v = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
        packs = [(torch.empty((27, co, ci), dtype=torch.bfloat16, device="cuda"),
                  None if k == 1 else torch.empty((27, ci, co), dtype=torch.bfloat16, device="cuda")) for k, (co, ci) in enumerate(shapes)]
        gg = torch.Generator().manual_seed(5)
        for step in range(1, 4):
            grad = (torch.randn(n, generator=gg) * 0.01).cuda()
            if fused:
                ops.adam_pack_step(p, grad, m, v, 1e-3, step, 1e-5, 0.5, (0.9, 0.999), 1e-8,
                                   [(o, co, ci, wf, wb) for o, (co, ci), (wf, wb) in zip(offs, shapes, packs)])
            else:
                ops.adam_step(p, grad, m, v, 1e-3, step, 1e-5, 0.5)
                ws = [p[o:o + co * ci * 27].view(co, ci, 3, 3, 3) for o, (co, ci) in zip(offs, shapes)]
                ops.pack_conv_weights_batched(ws, torch.bfloat16, [True, False, True], packs)
        torch.cuda.synchronize()
        res[fused] = (p.clone(), m.clone(), v.clone(), packs)
    for a, b in zip(res[True][:3], res[False][:3]):
        assert torch.equal(a, b)
    assert float((res[True][0].cpu() - p0).abs().max()) > 1e-4
    for (wf_a, wb_a), (wf_b, wb_b) in zip(res[True][3], res[False][3]):
        assert torch.equal(wf_a, wf_b) and (wb_a is None or torch.equal(wb_a, wb_b))
    with pytest.raises(ValueError):          # a layer the tiled pack does not take
        ops.adam_pack_step(res[True][0], res[True][1], res[True][1], res[True][2], 1e-3, 1, 0.0, 1.0, (0.9, 0.999), 1e-8,
                           [(0, 8, 32, res[True][3][0][0], None)])


def test_train_step_with_the_fused_adam_pack_launch():
    """TrainStep with the optimiser launch writing the 3x3x3 packs (Engine.use_adam_pack, the default) against the step that
    repacks at the head of every forward: parameters bit-identical after four steps (both domains), no 3x3x3 layer is repacked by
    a forward after the first one, and a torch-side in-place edit of a weight between two steps IS seen (version counters)."""
    import fplx
    from fplx import ops
    p = dict(in_chns=1, feature_chns=[32, 64, 64, 128, 128], dropout=[0, 0, 0.3, 0, 0], conv_dims=[3] * 5, class_num=2,
             bilinear=False, num_domains=2, net_type="UNet2D5_dsbn", precision="bf16")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 1, 16, 32, 64, generator=g).cuda()
    lab = torch.zeros(2, 2, 16, 32, 64)
    lab[:, 0] = 1.0
    lab[:, 0, 4:10, 8:20, 16:40] = 0.0
    lab[:, 1, 4:10, 8:20, 16:40] = 1.0
    lab = lab.cuda()
    res = []
    for fuse in (True, False):
        torch.manual_seed(3)
        net = fplx.UNet2D5_dsbn(dict(p)).cuda()
        net.engine.use_adam_pack = fuse
        ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-3, weight_decay=1e-5)
        packed = []
        inner = ops.pack_conv_weights_batched

        def counting(ws, act_dtype, want_wb, into=None, stamps=None, verify=False, _inner=inner, _packed=packed):
            _packed.append((len(ws), bool(verify)))
            return _inner(ws, act_dtype, want_wb, into, stamps, verify)
        ops.pack_conv_weights_batched = counting
        try:
            for it in range(4):
                ts.step(x, lab, it % 2)
                if it == 1:
                    with torch.no_grad():
                        net.block1.conv.conv3d_2.weight.mul_(0.5)       # a torch-side edit: the next forward must repack
        finally:
            ops.pack_conv_weights_batched = inner
        torch.cuda.synchronize()
        res.append((net.flat_params.detach().clone(), list(packed)))
    assert torch.equal(res[0][0], res[1][0])
    n3 = res[1][1][0][0]
    assert res[1][1] == [(n3, False)] * 4                # unfused: every forward packs every 3x3x3 layer
    # fused: the first forward packs everything (nothing written yet), the forward behind the in-place edit again; the others
    # VERIFY the kept packs' stamps (one launch that repacks nothing here) and pack only the stem (1 input channel: not a layer
    # of the tiled pack)
    assert res[0][1] == [(n3, False), (n3 - 1, True), (1, False), (n3, False), (n3 - 1, True), (1, False)], res[0][1]


def test_pack_stamps_catch_writes_that_bypass_the_version_counters():
    """with the optimiser launch writing the packs, a write through `.data` (the reference's EMA / init paths,
    agent_seg.py) bumps no version counter - the next train step must still run on the edited weights.  Kernel level: a verify
    pass over unchanged masters rewrites nothing, after a `.data` edit of one layer it repacks exactly that layer's tiles, and
    the result equals a full pack.  Step level: p.data.mul_() between two TrainStep.step calls changes the result, to the bits
    of the same run with an explicit net.parameters_changed()."""
    import fplx
    from fplx import ops
    g = torch.Generator().manual_seed(4)
    ws = [(torch.randn(co, ci, 3, 3, 3, generator=g) * 0.1).cuda() for co, ci in ((32, 64), (64, 32), (16, 96))]
    bf = torch.bfloat16
    packs = [(torch.empty((27, w.shape[0], w.shape[1]), dtype=bf, device="cuda"),
              torch.empty((27, w.shape[1], w.shape[0]), dtype=bf, device="cuda")) for w in ws]
    stamps = [torch.zeros(ops.pack_stamp_floats(w.shape[0], w.shape[1]), device="cuda") for w in ws]
    ops.pack_conv_weights_batched(ws, bf, [True] * 3, packs, stamps, False)
    ref = ops.pack_conv_weights_batched(ws, bf, [True] * 3)
    for (a, b), (c, d) in zip(packs, ref):
        assert torch.equal(a, c) and torch.equal(b, d)
    assert all(float(s_.abs().sum()) > 0 for s_ in stamps)
    # poison the packs: a verify pass over unchanged masters must not touch them
    for a, b in packs:
        a.fill_(7.0)
        b.fill_(7.0)
    ops.pack_conv_weights_batched(ws, bf, [True] * 3, packs, stamps, True)
    assert all(float((a.float() - 7.0).abs().max()) == 0 and float((b.float() - 7.0).abs().max()) == 0 for a, b in packs)
    # a `.data` write to ONE layer: that layer is repacked (equal to a fresh pack), the others keep the poison
    ws[1].data.mul_(0.5)
    ops.pack_conv_weights_batched(ws, bf, [True] * 3, packs, stamps, True)
    ref = ops.pack_conv_weights_batched(ws, bf, [True] * 3)
    assert torch.equal(packs[1][0], ref[1][0]) and torch.equal(packs[1][1], ref[1][1])
    assert float((packs[0][0].float() - 7.0).abs().max()) == 0 and float((packs[2][1].float() - 7.0).abs().max()) == 0
    # ... and its stamps follow: a second verify is a no-op again
    packs[1][0].fill_(7.0)
    ops.pack_conv_weights_batched(ws, bf, [True] * 3, packs, stamps, True)
    assert float((packs[1][0].float() - 7.0).abs().max()) == 0
    # one output-channel row of one tile (what `w.data[3] = 0` does): only the tiles of that row
    ws[0].data[3].zero_()
    ops.pack_conv_weights_batched(ws, bf, [True] * 3, packs, stamps, True)
    ref = ops.pack_conv_weights_batched(ws, bf, [True] * 3)
    assert torch.equal(packs[0][0][:, :16], ref[0][0][:, :16]) and float((packs[0][0][:, 16:].float() - 7.0).abs().max()) == 0
    with pytest.raises(ValueError):
        ops.pack_conv_weights_batched(ws, bf, [True] * 3, packs, None, True)          # verify without stamps

    # ---- step level
    p = dict(in_chns=1, feature_chns=[32, 64, 64, 128, 128], dropout=[0, 0, 0.3, 0, 0], conv_dims=[3] * 5, class_num=2,
             bilinear=False, num_domains=2, net_type="UNet2D5_dsbn", precision="bf16")
    x = torch.randn(2, 1, 16, 32, 64, generator=g).cuda()
    lab = torch.zeros(2, 2, 16, 32, 64)
    lab[:, 0] = 1.0
    lab[:, 0, 4:10, 8:20, 16:40] = 0.0
    lab[:, 1, 4:10, 8:20, 16:40] = 1.0
    lab = lab.cuda()
    res = {}
    for mode in ("data", "data+changed", "none"):
        torch.manual_seed(3)
        net = fplx.UNet2D5_dsbn(dict(p)).cuda()
        assert net.engine.use_adam_pack
        ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-3, weight_decay=1e-5)
        outs = []
        for it in range(4):
            outs.append(ts.step(x, lab, it % 2).clone())
            if it == 1 and mode != "none":
                v0 = net.block1.conv.conv3d_2.weight._version
                net.block1.conv.conv3d_2.weight.data.mul_(0.5)          # no version counter moves
                net.up3.conv.conv3d_1.weight.data.add_(0.01)
                assert net.block1.conv.conv3d_2.weight._version == v0
                if mode == "data+changed":
                    net.parameters_changed()
        torch.cuda.synchronize()
        res[mode] = (torch.stack(outs), net.flat_params.detach().clone())
    assert torch.equal(res["data"][0], res["data+changed"][0]) and torch.equal(res["data"][1], res["data+changed"][1])
    assert torch.equal(res["data"][0][:2], res["none"][0][:2])
    assert float((res["data"][0][2:, 0] - res["none"][0][2:, 0]).abs().max()) > 1e-4          # the edit IS seen by the next step


def test_backward_behind_an_optimiser_step_is_refused():
    """a saved forward aliases the persistent weight packs; forward A, optimiser step, backward A would take A's data
    gradients with the updated weights - the engine raises instead"""
    import fplx
    p = dict(in_chns=1, feature_chns=[32, 64, 64, 128, 128], dropout=[0, 0, 0, 0, 0], conv_dims=[3] * 5, class_num=2,
             bilinear=False, num_domains=2, net_type="UNet2D5_dsbn", precision="bf16")
    torch.manual_seed(2)
    net = fplx.UNet2D5_dsbn(dict(p)).cuda().train()
    opt = fplx.FusedAdam(net, 1e-3)
    x = torch.randn(1, 1, 16, 32, 64).cuda()
    dom = torch.zeros(1, dtype=torch.long)
    la = net(x, domain_label=dom).sum()
    lb = net(x, domain_label=dom).sum()
    lb.backward()                       # same generation of packs: fine
    opt.step()
    net(x, domain_label=dom)            # the forward behind the step packs the updated weights into the same buffers
    with pytest.raises(RuntimeError, match="overwritten"):
        la.backward()
    net.zero_grad()
    net(x, domain_label=dom).sum().backward()          # and the engine carries on
    opt.step()
    torch.cuda.synchronize()


@pytest.mark.parametrize("shape", [(2, 1, 32, 3, 9, 35), (1, 4, 32, 2, 16, 64), (1, 1, 64, 4, 8, 32),
                                   # in_chns = 1 runs the LDS-free row kernel (stem_fwd_rows): a volume of one row, a width
                                   # one past a segment, a single voxel, several samples
                                   (1, 1, 32, 1, 1, 40), (3, 1, 32, 5, 7, 161), (1, 1, 32, 1, 1, 1), (2, 1, 64, 6, 20, 96),
                                   # in_chns = 4 (config 5) on the same kernel: the channels' planes sit back to back, a halo
                                   # offset of channel ci > 0 lands in channel ci - 1 and must be cleared by the padding selects
                                   (2, 4, 32, 3, 9, 35), (3, 4, 64, 2, 5, 70), (1, 4, 32, 1, 1, 1)])
def test_stem_kernels_bf16(shape):
    """fp32 NCDHW network input -> bf16 NDHWC (MFMA stem kernels): forward + statistics, weight gradient"""
    from fplx import ops
    n, cin, cout, d, h, w = shape
    x = torch.from_numpy(detdata.normal("st.x%s" % (shape,), (n, cin, d, h, w)))
    wt = torch.from_numpy(detdata.normal("st.w%s" % (shape,), (cout, cin, 3, 3, 3), 0.3)).bfloat16().float()
    b = torch.from_numpy(detdata.normal("st.b%s" % (shape,), (cout,)))
    dy = torch.from_numpy(detdata.normal("st.dy%s" % (shape,), (n, cout, d, h, w))).bfloat16().float()
    xq = x.bfloat16().float()              # the kernel rounds the input to bf16 when staging it
    xr, wr = xq.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, b, padding=1)
    yr.backward(dy)
    dims = (n, d, h, w)
    wf, _ = ops.pack_conv_weight(wt.cuda(), torch.bfloat16, False)
    rows = ops.conv3d_stats_rows(dims, cin, cout, (3, 3, 3), ops.F32, ops.BF16)
    stats = torch.zeros((rows, 2, cout), dtype=torch.float32, device="cuda")
    y = torch.zeros((n * d * h * w, cout), dtype=torch.bfloat16, device="cuda")
    ops.conv3d_fwd(x.cuda(), ops.planar_strides(cin, d, h, w), ops.F32, wf, b.cuda(), y, ops.cl_strides(d, h, w, cout),
                   ops.BF16, dims, cin, cout, (3, 3, 3), stats)
    got = uncl(y.float().cpu(), n, d, h, w)
    scale = float(yr.detach().abs().max())
    assert float((got - yr.detach()).abs().max()) < 1e-2 * scale
    yf = cl(yr.detach())
    np.testing.assert_allclose(stats.sum(0)[0].cpu().numpy(), yf.sum(0).numpy(), atol=2e-3 * scale * yf.shape[0] ** 0.5)
    np.testing.assert_allclose(stats.sum(0)[1].cpu().numpy(), (yf * yf).sum(0).numpy(), rtol=1e-2)
    ws = torch.empty(ops.conv3d_wgrad_ws_bytes(dims, cin, cout, (3, 3, 3)), dtype=torch.uint8, device="cuda")
    dw = torch.zeros((cout, cin, 3, 3, 3), dtype=torch.float32, device="cuda")
    ops.conv3d_wgrad(x.cuda(), ops.planar_strides(cin, d, h, w), ops.F32, cl(dy).bfloat16().cuda(),
                     ops.cl_strides(d, h, w, cout), ops.BF16, dw, None, dims, cin, cout, (3, 3, 3), ws)
    assert float((dw.cpu() - wr.grad).abs().max()) < 1e-2 * float(wr.grad.abs().max())


@pytest.mark.parametrize("shape", [(2, 1, 32, 3, 9, 35), (1, 1, 32, 5, 16, 64), (1, 4, 32, 2, 16, 64), (3, 1, 32, 5, 7, 161),
                                   (1, 1, 32, 1, 1, 1), (2, 1, 64, 6, 20, 96), (2, 4, 32, 3, 10, 48)])
def test_stem_weight_gradient_with_the_batchnorm_apply_inside(shape):
    """fplx_stem_wgrad_bn (round 6): the stem site's dy has one consumer - the stem's weight gradient - which forms it from the
    stored convolution output y and the gradient w.r.t. the site's output while it stages them, so the apply pass of the site's
    BatchNorm + PReLU backward is never run and dy never stored.  Same arithmetic on the same values, rounded where the stored
    tensor was: dw bit for bit the result of fplx_bn_act_bwd (all three stages) + fplx_conv3d_wgrad; ragged tiles, single voxels,
    several samples, 4 input channels, 64 output channels (two 32-channel launches)."""
    from fplx import ops
    n, cin, cout, d, h, w = shape
    dims, v = (n, d, h, w), n * d * h * w
    assert ops.stem_wgrad_bn_ok(dims, cin, cout) and not ops.stem_wgrad_bn_ok(dims, 2, cout) and not ops.stem_wgrad_bn_ok(dims, cin, 48)
    x = torch.from_numpy(detdata.normal("sw.x%s" % (shape,), (n, cin, d, h, w))).cuda()
    y = torch.from_numpy(detdata.normal("sw.y%s" % (shape,), (v, cout))).bfloat16().cuda()
    dout = torch.from_numpy(detdata.normal("sw.d%s" % (shape,), (v, cout))).bfloat16().cuda()
    g = torch.Generator().manual_seed(7)
    mean, var = torch.randn(cout, generator=g) * 0.3, torch.rand(cout, generator=g) + 0.5
    gamma, beta = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    rstd = torch.rsqrt(var + 1e-5)
    bnbuf = torch.stack([mean, rstd, gamma * rstd, beta - mean * gamma * rstd]).cuda()
    slope = torch.tensor([0.25]).cuda()
    part = torch.empty(ops.num_partials(v) * (2 * cout + 1), device="cuda")
    coef = torch.empty((2, 2 * cout), device="cuda")           # (the engine's buffer is wider than the site: the finalize writes it flat)
    ws = torch.empty(ops.conv3d_wgrad_ws_bytes(dims, cin, cout, (3, 3, 3)), dtype=torch.uint8, device="cuda")
    # ---- the separate passes
    gr = [torch.zeros(cout, device="cuda"), torch.zeros(cout, device="cuda"), torch.zeros(1, device="cuda")]
    dy = torch.empty_like(dout)
    ops.bn_act_bwd(y, dout, dy, bnbuf, slope, 0.0, 0, 0, cout, True, gr[0], gr[1], gr[2], part, coef)
    dw_ref = torch.zeros((cout, cin, 3, 3, 3), device="cuda")
    ops.conv3d_wgrad(x, ops.planar_strides(cin, d, h, w), ops.F32, dy, ops.cl_strides(d, h, w, cout), ops.BF16, dw_ref, None, dims,
                     cin, cout, (3, 3, 3), ws)
    # ---- reduction + finalize only, then the fused weight gradient
    gf = [torch.zeros(cout, device="cuda"), torch.zeros(cout, device="cuda"), torch.zeros(1, device="cuda")]
    coef2 = torch.full_like(coef, float("nan"))
    d2 = dout.clone()
    ops.bn_act_bwd(y, d2, d2, bnbuf, slope, 0.0, 0, 0, cout, True, gf[0], gf[1], gf[2], part, coef2, apply=False)
    assert torch.equal(d2, dout)                                  # no apply pass ran
    for a_, b_ in zip(gf, gr):
        assert torch.equal(a_, b_)
    dw = torch.full((cout, cin, 3, 3, 3), 7.0, device="cuda")
    ops.stem_wgrad_bn(x, y, d2, bnbuf, slope, coef2, dw, dims, cin, cout, ws)
    assert torch.equal(dw, dw_ref)
    assert float(dw_ref.abs().max()) > 0 or v == 1
    from fplx._lib import FplxError
    with pytest.raises(FplxError):
        ops.stem_wgrad_bn(x, y, d2, bnbuf, slope, coef2, dw, dims, cin, cout, ws[:16])


def test_network_step_with_and_without_the_fused_stem_weight_gradient():
    """the engine with the stem's weight gradient forming dy itself (Engine.use_stem_wgrad_bn, the default) against the separate
    apply pass: the same arithmetic on the same values - parameters, losses and the Adam moments bit for bit after three steps; a
    dropout at the stem site and a 2.5D stem (Conv2d) keep the separate pass"""
    import fplx
    from fplx import ops
    p = dict(in_chns=1, feature_chns=[32, 64, 128, 256, 512], dropout=[0, 0, 0.3, 0.4, 0.5], conv_dims=[3] * 5, class_num=2,
             bilinear=False, num_domains=2, net_type="UNet2D5_dsbn", precision="bf16")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 1, 16, 32, 64, generator=g).cuda()
    lab = torch.zeros(2, 2, 16, 32, 64)
    lab[:, 0] = 1.0
    lab[:, 0, 4:10, 8:20, 16:40] = 0.0
    lab[:, 1, 4:10, 8:20, 16:40] = 1.0
    lab = lab.cuda()
    res, calls = [], []
    inner = ops.stem_wgrad_bn

    def counting(*a, **kw):
        calls.append(1)
        return inner(*a, **kw)
    ops.stem_wgrad_bn = counting
    try:
        for fuse, extra in ((True, {}), (False, {}), (True, dict(dropout=[0.2, 0, 0.3, 0.4, 0.5])), (True, dict(conv_dims=[2, 2, 3, 3, 3]))):
            torch.manual_seed(3)
            net = fplx.UNet2D5_dsbn(dict(p, **extra)).cuda()
            net.engine.use_stem_wgrad_bn = fuse
            with torch.no_grad():
                net.dropout_seed, net._fwd_counter = 9, 0
            ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-3, weight_decay=1e-5)
            n0 = len(calls)
            outs = [ts.step(x, lab, it % 2).clone() for it in range(3)]
            torch.cuda.synchronize()
            res.append((net.flat_params.detach().clone(), torch.stack(outs), ts.opt.exp_avg.clone(), len(calls) - n0))
    finally:
        ops.stem_wgrad_bn = inner
    assert res[0][3] == 3 and res[1][3] == 0 and res[2][3] == 0 and res[3][3] == 0
    for k in range(3):
        assert torch.equal(res[0][k], res[1][k]), k


@pytest.mark.parametrize("shape", [(2, 32, 2, 3, 9, 35), (1, 32, 3, 2, 16, 64), (1, 64, 2, 2, 8, 40)])
def test_outconv_kernels_bf16(shape):
    """bf16 NDHWC features <-> fp32 NCDHW logits, kernel (1,3,3): forward, data gradient, weight gradient"""
    from fplx import ops
    n, c0, ncls, d, h, w = shape
    x = torch.from_numpy(detdata.normal("oc.x%s" % (shape,), (n, c0, d, h, w))).bfloat16().float()
    wt = torch.from_numpy(detdata.normal("oc.w%s" % (shape,), (ncls, c0, 1, 3, 3), 0.2))
    b = torch.from_numpy(detdata.normal("oc.b%s" % (shape,), (ncls,)))
    dl = torch.from_numpy(detdata.normal("oc.dl%s" % (shape,), (n, ncls, d, h, w)))
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, br, padding=(0, 1, 1))
    yr.backward(dl)
    dims = (n, d, h, w)
    xg = cl(x).bfloat16().cuda()
    wf, _ = ops.pack_conv_weight(wt.cuda(), torch.float32, False)
    _, wb = ops.pack_conv_weight(wt.cuda(), torch.bfloat16, True)
    out = torch.zeros((n, ncls, d, h, w), dtype=torch.float32, device="cuda")
    ops.conv3d_fwd(xg, ops.cl_strides(d, h, w, c0), ops.BF16, wf, b.cuda(), out, ops.planar_strides(ncls, d, h, w),
                   ops.F32, dims, c0, ncls, (1, 3, 3), None)
    # the MFMA path rounds the fp32 weights to bf16 operands: 2^-9 relative per product
    assert float((out.cpu() - yr.detach()).abs().max()) < 1e-2 * float(yr.detach().abs().max())
    dx = torch.zeros((xg.shape[0], c0), dtype=torch.bfloat16, device="cuda")
    ops.conv3d_fwd(dl.cuda(), ops.planar_strides(ncls, d, h, w), ops.F32, wb, None, dx, ops.cl_strides(d, h, w, c0),
                   ops.BF16, dims, ncls, c0, (1, 3, 3), None)
    assert float((uncl(dx.float().cpu(), n, d, h, w) - xr.grad).abs().max()) < 2e-2 * float(xr.grad.abs().max())
    ws = torch.empty(ops.conv3d_wgrad_ws_bytes(dims, c0, ncls, (1, 3, 3)), dtype=torch.uint8, device="cuda")
    dw = torch.zeros((ncls, c0, 1, 3, 3), dtype=torch.float32, device="cuda")
    db = torch.zeros(ncls, dtype=torch.float32, device="cuda")
    ops.conv3d_wgrad(xg, ops.cl_strides(d, h, w, c0), ops.BF16, dl.cuda(), ops.planar_strides(ncls, d, h, w), ops.F32,
                     dw, db, dims, c0, ncls, (1, 3, 3), ws)
    assert float((dw.cpu() - wr.grad).abs().max()) < 1e-2 * float(wr.grad.abs().max())
    assert float((db.cpu() - br.grad).abs().max()) < 1e-4 * float(br.grad.abs().max())


def test_pack_conv_weights_batched_equals_per_layer_packs():
    """fplx_pack_conv_weights_batched (every 3x3x3 layer in one launch) against fplx_pack_conv_weight layer by layer:
    bit-identical packs, including a layer the tiled kernel does not take (Cin = 1), a missing data-gradient pack and
    the fp32 mode (which falls back to the per-layer kernel inside)."""
    from fplx import ops
    shapes = [(32, 1), (32, 32), (64, 32), (64, 64), (128, 64), (48, 96), (512, 256)]
    ws = [torch.from_numpy(detdata.normal("pk.%d.%d" % s, s + (3, 3, 3), 0.3)).cuda() for s in shapes]
    want = [i != 0 for i in range(len(ws))]
    for dtype in (torch.bfloat16, torch.float32):
        got = ops.pack_conv_weights_batched(ws, dtype, want)
        for w, wb_wanted, (wf, wb) in zip(ws, want, got):
            rf, rb = ops.pack_conv_weight(w, dtype, wb_wanted)
            assert torch.equal(wf, rf)
            assert (wb is None and rb is None) or torch.equal(wb, rb)
    with pytest.raises(ValueError):
        ops.pack_conv_weights_batched(ws * 5, torch.bfloat16, want * 5)          # more than 32 layers in one call


def test_fused_downblock_tail_equals_the_separate_passes():
    """fplx_bn_act_pool_fwd == bn_act_fwd then maxpool, and fplx_pool_bwd_bn_reduce == maxpool_bwd then the BatchNorm
    reduction, bit for bit (same arithmetic, same bf16 rounding points, same first-maximum rule)"""
    from fplx import ops
    g = torch.Generator().manual_seed(5)
    for (n, d, h, w, c, pd) in ((2, 4, 8, 12, 32, 2), (1, 3, 6, 8, 64, 1), (1, 2, 4, 4, 16, 2)):
        v, vo = n * d * h * w, n * (d // pd) * (h // 2) * (w // 2)
        y = torch.randn(v, c, generator=g).bfloat16().cuda()
        # repeated values make ties: the first maximum must win in both forms
        y[::3] = y[1::3][: y[::3].shape[0]]
        bnbuf = torch.stack([torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5, torch.rand(c, generator=g) + 0.5,
                             torch.randn(c, generator=g) * 0.1]).cuda()
        slope = torch.full((1,), 0.25, device="cuda")
        a_ref = torch.empty_like(y)
        ops.bn_act_fwd(y, a_ref, bnbuf, slope, 0.0, 1, 0, c)
        p_ref = torch.empty((vo, c), dtype=torch.bfloat16, device="cuda")
        ops.maxpool2_fwd(a_ref, p_ref, (n, d, h, w), c, pd)
        a_f, p_f = torch.empty_like(y), torch.empty_like(p_ref)
        ops.bn_act_pool_fwd(y, a_f, p_f, bnbuf, slope, (n, d, h, w), c, pd)
        assert torch.equal(a_f, a_ref) and torch.equal(p_f, p_ref)
        dy = (torch.randn(vo, c, generator=g) * 0.01).bfloat16().cuda()
        dskip = (torch.randn(v, c, generator=g) * 0.01).bfloat16().cuda()
        dx_ref = torch.empty_like(y)
        ops.maxpool2_bwd(a_ref, dy, dskip, dx_ref, (n, d, h, w), c, pd)
        rows = ops.num_partials(v)
        part_ref = torch.zeros((rows, 2 * c + 1), device="cuda")
        ops.call("fplx_bn_act_bwd_reduce", ops.ptr(y), c, ops.ptr(dx_ref), c, ops.ptr(bnbuf[0]), ops.ptr(bnbuf[1]), ops.ptr(bnbuf[2]),
                 ops.ptr(bnbuf[3]), ops.ptr(slope), 0.0, 1, 0, v, c, ops.dt_of(y), ops.ptr(part_ref), ops.stream())
        dx_f = torch.empty_like(y)
        part_f = torch.zeros((rows, 2 * c + 1), device="cuda")
        ops.pool_bwd_bn_reduce(y, dy, dskip, dx_f, bnbuf, slope, (n, d, h, w), c, part_f, pd)
        assert torch.equal(dx_f, dx_ref)
        tot_ref, tot_f = part_ref.double().sum(0), part_f.double().sum(0)          # the rows partition the voxels differently
        assert float((tot_ref - tot_f).abs().max()) <= 1e-5 * float(tot_ref.abs().max()) + 1e-9


def _bn_site_setup(n, d, h, w, c, tag):
    """a BatchNorm site in bf16: pre-BN tensor y [V, C] with per-channel offsets / scales, affine parameters, PReLU slope,
    batch statistics through the library (fplx_channel_stats -> fplx_bn_train_finalize)"""
    from fplx import ops
    v = n * d * h * w
    g = torch.Generator().manual_seed(11 + c)
    y = (torch.randn(v, c, generator=g) * (0.5 + torch.rand(c, generator=g)) + torch.randn(c, generator=g)).bfloat16()
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    slope = torch.full((1,), 0.25)
    yg = y.cuda()
    rows = ops.num_partials(v)
    stats = torch.empty((rows, 2, c), dtype=torch.float32, device="cuda")
    ops.call("fplx_channel_stats", ops.ptr(yg), c, v, c, ops.BF16, ops.ptr(stats), ops.stream())
    bnbuf = torch.empty((4, c), dtype=torch.float32, device="cuda")
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    nbt = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.bn_train_finalize(stats, rows, c, v, gamma.cuda(), beta.cuda(), rm, rv, nbt, bnbuf)
    return v, y, yg, gamma, beta, slope, bnbuf


def _bn_site_torch(y, gamma, beta, slope, keep, p):
    """float64 autograd of BatchNorm(train) -> PReLU -> dropout with a SUPPLIED keep mask, on the bf16 values of y"""
    yd = y.double().requires_grad_(True)
    gd, bd, sd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True), slope.double().requires_grad_(True)
    mean, var = yd.mean(0), yd.var(0, unbiased=False)
    z = (yd - mean) * torch.rsqrt(var + 1e-5) * gd + bd
    a = torch.where(z > 0, z, z * sd)
    if p > 0:
        a = a * keep.double() * float(np.float32(1.0 / (1.0 - float(np.float32(p)))))
    return yd, gd, bd, sd, a


@pytest.mark.parametrize("shape,p", [((2, 8, 32, 32, 32), 0.0),      # level-0-like: 32 channels, the group-stationary kernels
                                     ((1, 6, 20, 24, 128), 0.3),     # level-2-like with dropout (Philox masks from the oracle)
                                     ((2, 4, 16, 16, 64), 0.5),
                                     ((1, 5, 10, 10, 512), 0.0),     # level 4: 64 channel groups, few voxels
                                     ((1, 3, 7, 9, 16), 0.4)])       # ragged voxel count, 2 groups
def test_bn_act_bf16_kernels_against_torch_autograd(shape, p):
    """bf16 bn_act_fwd_g_k / bn_act_bwd_reduce_k / bn_act_bwd_finalize_k / bn_act_bwd_apply_g_k against float64 torch
    autograd of BatchNorm3d(train) + PReLU + dropout with the supplied mask (these kernels produce the d gamma /
    d beta / d slope the end-to-end rule is loose on; so far they were only compared with each other).
    dx <= 2e-2 of its max (bf16 storage), d gamma / d beta / d slope <= 1e-2 relative."""
    from fplx import ops
    from oracle import np_ref as N
    n, d, h, w, c = shape
    v, y, yg, gamma, beta, slope, bnbuf = _bn_site_setup(n, d, h, w, c, "a")
    seed, sid = 1234, 7
    keep = torch.from_numpy(N.philox_keep_mask(seed, sid, v * c, p).reshape(v, c)) if p > 0 else None
    yd, gd, bd, sd, a_ref = _bn_site_torch(y, gamma, beta, slope, keep, p)
    a = torch.empty_like(yg)
    ops.bn_act_fwd(yg, a, bnbuf, slope.cuda(), p, seed, sid, c)
    amax = float(a_ref.abs().max())
    assert float((a.float().cpu().double() - a_ref.detach()).abs().max()) < 1e-2 * amax
    if p > 0:                                    # the mask itself: exactly the oracle's Philox stream
        assert bool(((a.float().cpu() == 0) | keep).all()) and float((a.float().cpu()[~keep]).abs().max()) == 0.0
    g = torch.Generator().manual_seed(5)
    dout = (torch.randn(v, c, generator=g) * 0.01).bfloat16()
    a_ref.backward(dout.double())
    dgamma, dbeta, dslope = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda"), torch.zeros(1, device="cuda")
    part = torch.empty((ops.num_partials(v), 2 * c + 1), dtype=torch.float32, device="cuda")
    coef = torch.empty((2, c), dtype=torch.float32, device="cuda")
    dy = torch.empty_like(yg)
    ops.bn_act_bwd(yg, dout.cuda(), dy, bnbuf, slope.cuda(), p, seed, sid, c, True, dgamma, dbeta, dslope, part, coef)
    dx_ref = yd.grad
    assert float((dy.float().cpu().double() - dx_ref).abs().max()) < 2e-2 * float(dx_ref.abs().max())
    for got, ref in ((dgamma, gd.grad), (dbeta, bd.grad), (dslope, sd.grad)):
        assert float((got.cpu().double() - ref).abs().max()) < 1e-2 * float(ref.abs().max()), (got, ref)
    # a sign error anywhere would show: none of these sums cancels on this data
    assert float(sd.grad.abs()) > 10 * float((dslope.cpu().double() - sd.grad).abs())


@pytest.mark.parametrize("shape,pd", [((2, 8, 32, 32, 32), 2), ((1, 6, 16, 24, 64), 1), ((1, 4, 8, 8, 128), 2)])
def test_pool_bwd_bn_reduce_bf16_against_torch_autograd(shape, pd):
    """the fused DownBlock tail - fplx_bn_act_pool_fwd, then fplx_pool_bwd_bn_reduce -> bn_act_bwd(reduced) - against float64
    autograd of BatchNorm(train) -> PReLU -> {skip, MaxPool}; the pooling sees the bf16-stored a2 in both (straight-through
    rounding in the reference), so the arg-max agrees except where fp32 / fp64 round a2 differently (a handful of voxels)."""
    from fplx import ops
    n, d, h, w, c = shape
    v, y, yg, gamma, beta, slope, bnbuf = _bn_site_setup(n, d, h, w, c, "p")
    yd, gd, bd, sd, a_ref = _bn_site_torch(y, gamma, beta, slope, None, 0.0)
    a_q = a_ref + (a_ref.detach().float().bfloat16().double() - a_ref.detach())        # value rounded, gradient straight through
    a5 = a_q.view(n, d, h, w, c).permute(0, 4, 1, 2, 3)
    k = (pd, 2, 2)
    pooled_ref = F.max_pool3d(a5, k, k)
    do, ho, wo = d // pd, h // 2, w // 2
    vo = n * do * ho * wo
    g = torch.Generator().manual_seed(9)
    dyp = (torch.randn(vo, c, generator=g) * 0.01).bfloat16()
    dskip = (torch.randn(v, c, generator=g) * 0.01).bfloat16()
    loss = (pooled_ref.permute(0, 2, 3, 4, 1).reshape(vo, c) * dyp.double()).sum() + (a_q * dskip.double()).sum()
    loss.backward()
    a2, pooled = torch.empty_like(yg), torch.empty((vo, c), dtype=torch.bfloat16, device="cuda")
    ops.bn_act_pool_fwd(yg, a2, pooled, bnbuf, slope.cuda(), (n, d, h, w), c, pd)
    assert float((pooled.float().cpu().double() - pooled_ref.detach().permute(0, 2, 3, 4, 1).reshape(vo, c)).abs().max()) < \
        1e-2 * float(pooled_ref.abs().max())
    part = torch.empty((ops.num_partials(v), 2 * c + 1), dtype=torch.float32, device="cuda")
    coef = torch.empty((2, c), dtype=torch.float32, device="cuda")
    d_a2 = torch.empty_like(yg)
    ops.pool_bwd_bn_reduce(yg, dyp.cuda(), dskip.cuda(), d_a2, bnbuf, slope.cuda(), (n, d, h, w), c, part, pd)
    dgamma, dbeta, dslope = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda"), torch.zeros(1, device="cuda")
    ops.bn_act_bwd(yg, d_a2, d_a2, bnbuf, slope.cuda(), 0.0, 0, 0, c, True, dgamma, dbeta, dslope, part, coef, reduced=True)
    err = (d_a2.float().cpu().double() - yd.grad).abs()
    lim = 2e-2 * float(yd.grad.abs().max())
    assert int((err > lim).sum()) <= 8, (int((err > lim).sum()), float(err.max()), lim)      # arg-max flips at rounding ties
    for got, ref in ((dgamma, gd.grad), (dbeta, bd.grad), (dslope, sd.grad)):
        assert float((got.cpu().double() - ref).abs().max()) < 1e-2 * float(ref.abs().max()), (got, ref)


@pytest.mark.parametrize("shape", [
    (2, 256, 256, 10, 20, 20),       # level 3 of the benchmark (5 x 4 x 8 = 8000 voxels: 63 chunks, 8 voxel ranges)
    (2, 512, 512, 5, 10, 10),        # level 4: 1000 voxels, the last chunk ragged, HW = 100 < chunk
    (2, 64, 128, 20, 40, 40),        # level 2: W at the kernel's limit (40)
    (3, 32, 96, 3, 7, 9),            # odd everything, n = 3, Cout % 64 != 0 (one co tile per block), tiny volume (567 voxels)
    (1, 96, 64, 2, 2, 2),            # 8 voxels: every pair crosses a border
    (1, 64, 64, 9, 33, 17)])         # K split whose ranges cut through rows and slices
@pytest.mark.parametrize("lw", [2, 0])
def test_conv3d_wgrad_vox_kernel(shape, lw):
    """conv_wgrad_vox / conv_wgrad_vox_lw (voxel-GEMM weight gradient of the deep levels, conv_mfma.hip; lw = 2: the loader-wave
    form of round 5 on every shape - by default it takes the volumes of at most 2048 voxels): against torch autograd, bf16 operands, forced on every shape by the tuning knob
    wg_vox = 2; and the same numbers as the footprint march (wg_vox = 0) up to the order of the fp32 additions."""
    from fplx import ops
    _lib = ops._lib
    n, cin, cout, d, h, w = shape
    q = lambda t: t.bfloat16().float()
    x = q(torch.from_numpy(detdata.normal("vx.x%s" % (shape,), (n, cin, d, h, w))))
    dy = q(torch.from_numpy(detdata.normal("vx.dy%s" % (shape,), (n, cout, d, h, w))))
    wr = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    F.conv3d(x, wr, None, padding=1).backward(dy)
    bf, dt, dims = torch.bfloat16, ops._DT[torch.bfloat16], (n, d, h, w)
    # operands as channel slices of wider buffers (ld > channels)
    xw = torch.full((n * d * h * w, cin + 32), 3.0, dtype=bf, device="cuda")
    xw[:, 8:8 + cin] = cl(x).to(bf).cuda()
    dyw = torch.full((n * d * h * w, cout + 16), 5.0, dtype=bf, device="cuda")
    dyw[:, 16:] = cl(dy).to(bf).cuda()
    xg, dyg = xw[:, 8:8 + cin], dyw[:, 16:]
    got = {}
    for mode in (2, 0):
        _lib.set_tuning("wg_vox", mode)
        _lib.set_tuning("wg_vox_lw", lw)
        try:
            ws = torch.empty(ops.conv3d_wgrad_ws_bytes(dims, cin, cout, (3, 3, 3)), dtype=torch.uint8, device="cuda")
            dw = torch.full((cout, cin, 3, 3, 3), 7.0, dtype=torch.float32, device="cuda")
            ops.conv3d_wgrad(xg, ops.cl_strides(d, h, w, cin + 32), dt, dyg, ops.cl_strides(d, h, w, cout + 16), dt, dw, None,
                             dims, cin, cout, (3, 3, 3), ws)
            got[mode] = dw.cpu()
        finally:
            _lib.set_tuning("wg_vox", 1)
            _lib.set_tuning("wg_vox_lw", 1)
    scale = float(wr.grad.abs().max())
    assert float((got[2] - wr.grad).abs().max()) < 1e-4 * scale + 1e-6, float((got[2] - wr.grad).abs().max()) / scale
    assert float((got[2] - got[0]).abs().max()) < 1e-4 * scale + 1e-6


@pytest.mark.parametrize("shape,geo", [
    ((1, 32, 32, 9, 16, 64), 1),         # 8 x 32 footprints, exact fit, one depth segment
    ((2, 32, 64, 7, 20, 70), 1),         # ... ragged in h and w, n = 2, two co tiles
    ((1, 64, 32, 37, 24, 64), 1),        # ... several depth segments, two ci tiles
    ((1, 32, 32, 12, 32, 48), 2),        # 16 x 16 footprints, exact fit
    ((3, 32, 96, 6, 17, 65), 2),         # ... ragged, n = 3, three co tiles
    ((1, 64, 64, 20, 40, 40), 2),        # ... level-2-like
    ((2, 64, 128, 20, 40, 40), 3),       # 8 x 16 footprints: level 2 of the benchmark
    ((1, 96, 32, 5, 9, 17), 3),          # ... ragged, short volume
    ((2, 32, 32, 40, 48, 64), 0),        # the dispatcher's own choice
    ((1, 128, 64, 10, 20, 20), 0)])
@pytest.mark.parametrize("m16,mb", [(1, 1), (0, 1), (1, 0), (0, 0), (1, 2)])
def test_conv3d_wgrad_roll_kernel(shape, geo, m16, mb):
    """conv_wgrad_roll (rolling-window weight gradient, conv_wgrad.hip): against torch autograd, bf16 operands given as channel
    slices of wider buffers, every footprint forced by the tuning knob wg_roll_geo; and the same numbers as the footprint
    march it replaces (wg_roll = 0) up to the order of the fp32 additions.  Reference: autograd of nn.Conv3d,
    unet2d5_dsbn.py:54-55.  m16 / mb (VERDICT r04 parity hole (c)): every shipped instantiation is reachable - the 8 x 32
    footprint on 32x32x16 (conv_wgrad_roll<8,32,.>, wg_roll_m16 = 0) instead of conv_wgrad_roll16, and the end-of-step barrier
    forms (MB = false, wg_roll_mb = 0) / the mid-step form for 8 x 16 too (wg_roll_mb = 2)."""
    from fplx import ops
    _lib = ops._lib
    n, cin, cout, d, h, w = shape
    q = lambda t: t.bfloat16().float()
    x = q(torch.from_numpy(detdata.normal("rl.x%s" % (shape,), (n, cin, d, h, w))))
    dy = q(torch.from_numpy(detdata.normal("rl.dy%s" % (shape,), (n, cout, d, h, w))))
    wr = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    F.conv3d(x, wr, None, padding=1).backward(dy)
    bf, dt, dims = torch.bfloat16, ops._DT[torch.bfloat16], (n, d, h, w)
    xw = torch.full((n * d * h * w, cin + 32), 3.0, dtype=bf, device="cuda")
    xw[:, 8:8 + cin] = cl(x).to(bf).cuda()
    dyw = torch.full((n * d * h * w, cout + 16), 5.0, dtype=bf, device="cuda")
    dyw[:, 16:] = cl(dy).to(bf).cuda()
    xg, dyg = xw[:, 8:8 + cin], dyw[:, 16:]
    got = {}
    for roll in (1, 0):
        _lib.set_tuning("wg_roll", roll)
        _lib.set_tuning("wg_roll_geo", geo)
        _lib.set_tuning("wg_roll_m16", m16)
        _lib.set_tuning("wg_roll_mb", mb)
        _lib.set_tuning("wg_vox", 0)
        try:
            ws = torch.empty(ops.conv3d_wgrad_ws_bytes(dims, cin, cout, (3, 3, 3)), dtype=torch.uint8, device="cuda")
            dw = torch.full((cout, cin, 3, 3, 3), 7.0, dtype=torch.float32, device="cuda")
            ops.conv3d_wgrad(xg, ops.cl_strides(d, h, w, cin + 32), dt, dyg, ops.cl_strides(d, h, w, cout + 16), dt, dw, None,
                             dims, cin, cout, (3, 3, 3), ws)
            got[roll] = dw.cpu()
        finally:
            _lib.set_tuning("wg_roll", 1)
            _lib.set_tuning("wg_roll_geo", 0)
            _lib.set_tuning("wg_roll_m16", 1)
            _lib.set_tuning("wg_roll_mb", 1)
            _lib.set_tuning("wg_vox", 1)
    scale = float(wr.grad.abs().max())
    assert float((got[1] - wr.grad).abs().max()) < 1e-4 * scale + 1e-6, float((got[1] - wr.grad).abs().max()) / scale
    assert float((got[1] - got[0]).abs().max()) < 1e-4 * scale + 1e-6


@pytest.mark.parametrize("shape", [
    (4, 64, 32, 16, 32, 64),       # the depth march on two 32-channel half-slabs (level 0), x0 shared by 2 passes
    (4, 128, 64, 16, 32, 44),      # the brick kernel's two-tensor form (level 1: 64 || 64 -> 64), ragged bricks in W
    (2, 256, 128, 16, 32, 48),     # level 2: 128 || 128 -> 128 (128-wide output tile)
    (3, 128, 64, 12, 40, 40)])     # n = 3 (x0 shared by all three)
def test_conv3d_fwd_act_two_tensor_form(shape):
    """fplx_conv3d_fwd_act with x1 != NULL (the decoder's first convolution on skip || up without the concatenation) and
    n_x0 (x0 = the ONE skip tensor all Monte-Carlo passes share, read modulo the batch): bit-identical to the same kernel
    family on the materialised concatenation, and within bf16 of torch's conv3d + PReLU."""
    from fplx import ops
    n, cin, cout, d, h, w = shape
    dims, half, bf = (n, d, h, w), cin // 2, torch.bfloat16
    assert ops.conv3d_fwd_act_ok(dims, cin, cout, False, True)
    q = lambda t: t.bfloat16().float()
    wt = q(torch.from_numpy(detdata.normal("act2.w%s" % (shape,), (cout, cin, 3, 3, 3))) * (2.0 / (27 * cin)) ** 0.5)
    bias = torch.from_numpy(detdata.normal("act2.b%s" % (shape,), (cout,))).float()
    slope = torch.tensor([0.2])
    wp, _ = ops.pack_conv_weight(wt.cuda(), bf, False)
    for n0 in (n, 1 if n % 2 else n // 2):
        x0 = q(torch.from_numpy(detdata.normal("act2.x0%s%d" % (shape, n0), (n0, half, d, h, w))))
        x1 = q(torch.from_numpy(detdata.normal("act2.x1%s" % (shape,), (n, half, d, h, w))))
        full = torch.cat([x0.repeat(n // n0, 1, 1, 1, 1), x1], 1)
        ref = F.prelu(F.conv3d(full, wt, bias, padding=1), slope)
        y2 = torch.full((n * d * h * w, cout + 8), 7.0, dtype=bf, device="cuda")
        ops.conv3d_fwd_act(cl(x0).to(bf).cuda(), cl(x1).to(bf).cuda(), wp, bias.cuda(), slope.cuda(), y2[:, :cout], dims, cin, cout,
                           False, 0 if n0 == n else n0)
        assert bool((y2[:, cout:] == 7.0).all())
        err = float((y2[:, :cout].float().cpu() - cl(ref)).abs().max())
        assert err < 2e-2 * float(ref.abs().max()), (n0, err)
        if ops.conv3d_fwd_act_ok(dims, cin, cout, False, False):
            y1 = torch.empty((n * d * h * w, cout), dtype=bf, device="cuda")
            ops.conv3d_fwd_act(cl(full).to(bf).cuda(), None, wp, bias.cuda(), slope.cuda(), y1, dims, cin, cout, False, 0)
            if plan_kernel(n, d, h, w, cin, cout) == 5:       # the brick kernel both ways: same order of additions
                assert torch.equal(y1, y2[:, :cout]), n0
            else:
                assert float((y1.float() - y2[:, :cout].float()).abs().max()) < 2e-2 * float(ref.abs().max())


@pytest.fixture(params=[1, 0], ids=["rows", "tiles"])
def outconv_dgrad_rows_knob(request):
    """the fused out_conv backward has two kernels for two classes: the stream of row segments (outconv_dgrad_rows, round 6, the
    default) and the tile kernel (outconv_dgrad_mfma<1, MODE>: knob outconv_dgrad_rows = 0; what 3 and 4 classes always take)"""
    from fplx import _lib
    _lib.set_tuning("outconv_dgrad_rows", request.param)
    _lib.set_tuning("outconv_fwd_rows", request.param)          # (and the fused forward: outconv_fwd_rows | outconv_fwd_t<1, true>)
    yield request.param
    _lib.set_tuning("outconv_dgrad_rows", 1)
    _lib.set_tuning("outconv_fwd_rows", 1)


@pytest.mark.parametrize("shape", [(2, 3, 9, 35, 2), (1, 2, 16, 64, 2), (1, 4, 24, 40, 3), (2, 5, 40, 96, 2), (1, 1, 1, 1, 2),
                                   (3, 2, 7, 161, 2), (1, 2, 70, 33, 4), (2, 1, 32, 32, 2), (1, 1, 33, 64, 1)])
def test_outconv_fused_with_the_last_sites_batchnorm_passes(shape, outconv_dgrad_rows_knob):
    """fplx_outconv_fwd_bn / fplx_outconv_dgrad_bn_reduce / _apply (out_conv fused with the BatchNorm + PReLU passes of the
    convolution site in front of it; reference unet2d5_dsbn.py:79-81 + 293-294, 307) against the calls they replace: forward -
    fplx_bn_act_fwd then fplx_conv3d_fwd (1,3,3): the activation and the logits bit for bit; backward - fplx_conv3d_fwd with the
    mirrored pack (out_conv's data gradient), then fplx_bn_act_bwd: the BatchNorm / PReLU gradients to 1e-4 (another order of
    additions), dy equal on >= 99 % of the elements and within bf16 rounding elsewhere.  Ragged tiles, n = 2, 2 and 3 classes."""
    from fplx import ops
    n, d, h, w, ncls = shape
    c0, v, bf = 32, n * d * h * w, torch.bfloat16
    dims = (n, d, h, w)
    assert ops.outconv_bn_ok(dims, c0, ncls) and not ops.outconv_bn_ok(dims, 64, ncls)
    g = torch.Generator().manual_seed(11)
    y = torch.randn(v, c0, generator=g).to(bf).cuda()
    gamma, beta = torch.rand(c0, generator=g) + 0.5, torch.randn(c0, generator=g) * 0.2
    mean, var = torch.randn(c0, generator=g) * 0.3, torch.rand(c0, generator=g) + 0.5
    rstd = torch.rsqrt(var + 1e-5)
    bnbuf = torch.stack([mean, rstd, gamma * rstd, beta - mean * gamma * rstd]).cuda()
    slope = torch.tensor([0.25]).cuda()
    wo = torch.randn(ncls, c0, 1, 3, 3, generator=g) * 0.1
    wof, _ = ops.pack_conv_weight(wo.cuda(), torch.float32, False)
    _, wob = ops.pack_conv_weight(wo.cuda(), bf, True)
    bias = torch.randn(ncls, generator=g).cuda()
    cl, pl = ops.cl_strides, ops.planar_strides
    # ---- forward
    a_ref = torch.empty(v, c0, dtype=bf, device="cuda")
    ops.bn_act_fwd(y, a_ref, bnbuf, slope, 0.0, 0, 0, c0)
    lg_ref = torch.empty(n, ncls, d, h, w, device="cuda")
    ops.conv3d_fwd(a_ref, cl(d, h, w, c0), ops.BF16, wof, bias, lg_ref, pl(ncls, d, h, w), ops.F32, dims, c0, ncls, (1, 3, 3), None)
    a = torch.full((v, c0), 7.0, dtype=bf, device="cuda")
    lg = torch.full((n, ncls, d, h, w), 7.0, device="cuda")
    ops.outconv_fwd_bn(y, bnbuf, slope, a, wof, bias, lg, dims, c0, ncls)
    assert torch.equal(a, a_ref) and torch.equal(lg, lg_ref)
    # ---- backward
    dl = (torch.randn(n, ncls, d, h, w, generator=g) * 0.05).cuda()
    d_ref = torch.empty(v, c0, dtype=bf, device="cuda")
    ops.conv3d_fwd(dl, pl(ncls, d, h, w), ops.F32, wob, None, d_ref, cl(d, h, w, c0), ops.BF16, dims, ncls, c0, (1, 3, 3), None)
    maxc = 64
    part = torch.empty((max(ops.num_partials(v), ops.outconv_bn_rows(dims, c0, ncls)), 2 * maxc + 1), device="cuda")
    coef = torch.empty((2, maxc), device="cuda")
    gr = [torch.zeros(c0, device="cuda"), torch.zeros(c0, device="cuda"), torch.zeros(1, device="cuda")]
    ops.bn_act_bwd(y, d_ref, d_ref, bnbuf, slope, 0.0, 0, 0, c0, True, gr[0], gr[1], gr[2], part, coef)
    gf = [torch.zeros(c0, device="cuda"), torch.zeros(c0, device="cuda"), torch.zeros(1, device="cuda")]
    dy = torch.full((v, c0), 7.0, dtype=bf, device="cuda")
    ops.outconv_dgrad_bn_bwd(dl, wob, y, bnbuf, slope, True, gf[0], gf[1], gf[2], part, coef, dy, dims, c0, ncls)
    for a_, b_ in zip(gf, gr):
        assert float((a_ - b_).abs().max()) <= 1e-4 * float(b_.abs().max()) + 1e-7, (a_, b_)
    diff = (dy.float() - d_ref.float()).abs()
    assert float(diff.max()) <= 1e-2 * float(d_ref.float().abs().max())
    assert float((diff == 0).float().mean()) >= 0.99
    with pytest.raises(ValueError):
        ops.outconv_dgrad_bn_bwd(dl, wob, y, bnbuf, slope, True, gf[0], gf[1], gf[2], part.view(-1)[:8], coef, dy, dims, c0, ncls)


@pytest.mark.parametrize("shape", [(2, 3, 9, 35, 2), (1, 2, 16, 64, 2), (1, 4, 24, 40, 3), (2, 5, 40, 96, 2), (1, 1, 1, 1, 2),
                                   (3, 2, 7, 161, 2), (2, 1, 32, 32, 2), (1, 1, 33, 64, 1), (1, 3, 5, 31, 3), (2, 2, 3, 97, 2)])
def test_outconv_wgrad_from_the_pre_batchnorm_tensor(shape):
    """fplx_outconv_wgrad_bn (round 6: out_conv's weight + bias gradient from the PRE-BatchNorm tensor y of the site in front
    of it, so that site's activation is never stored; reference unet2d5_dsbn.py:79-81 + 293-294, 307) and fplx_outconv_fwd_bn
    with a = NULL: (a) the logits without the activation store are the bits of the call that stores it; (b) dw / db against
    fplx_conv3d_wgrad on the stored activation - the same bf16 operands, another order of fp32 additions: 1e-4 of the largest
    entry; (c) dw / db against float64 torch (bf16 activation, fp32 dlogits): 1e-2.  1-3 classes, ragged rows, W < 32, n > 1;
    where the form is not available (4 classes, the tile kernels) the query says 0 and a = NULL is refused."""
    from fplx import ops, _lib
    n, d, h, w, ncls = shape
    c0, v, bf, dims = 32, n * d * h * w, torch.bfloat16, (n, d, h, w)
    assert ops.outconv_wgrad_bn_ws_bytes(dims, c0, ncls) > 0
    assert ops.outconv_wgrad_bn_ws_bytes(dims, c0, 4) == 0 and ops.outconv_wgrad_bn_ws_bytes(dims, 64, ncls) == 0
    g = torch.Generator().manual_seed(23 + w)
    y = (torch.randn(v, c0, generator=g) * (0.5 + torch.rand(c0, generator=g)) + torch.randn(c0, generator=g) * 0.3).to(bf).cuda()
    gamma, beta = torch.rand(c0, generator=g) + 0.5, torch.randn(c0, generator=g) * 0.2
    mean, var = torch.randn(c0, generator=g) * 0.3, torch.rand(c0, generator=g) + 0.5
    rstd = torch.rsqrt(var + 1e-5)
    bnbuf = torch.stack([mean, rstd, gamma * rstd, beta - mean * gamma * rstd]).cuda()
    slope = torch.tensor([0.25]).cuda()
    wo = torch.randn(ncls, c0, 1, 3, 3, generator=g) * 0.1
    wof, _ = ops.pack_conv_weight(wo.cuda(), torch.float32, False)
    bias = torch.randn(ncls, generator=g).cuda()
    cl, pl = ops.cl_strides, ops.planar_strides
    # (a) forward without the activation store
    a_ref = torch.empty(v, c0, dtype=bf, device="cuda")
    lg_ref = torch.empty(n, ncls, d, h, w, device="cuda")
    ops.outconv_fwd_bn(y, bnbuf, slope, a_ref, wof, bias, lg_ref, dims, c0, ncls)
    lg = torch.full((n, ncls, d, h, w), 7.0, device="cuda")
    ops.outconv_fwd_bn(y, bnbuf, slope, None, wof, bias, lg, dims, c0, ncls)
    assert torch.equal(lg, lg_ref)
    # (b) against the weight gradient on the stored activation
    dl = (torch.randn(n, ncls, d, h, w, generator=g) * 0.05).cuda()
    ws = torch.empty(max(ops.conv3d_wgrad_ws_bytes(dims, c0, ncls, (1, 3, 3)), ops.outconv_wgrad_bn_ws_bytes(dims, c0, ncls)),
                     dtype=torch.uint8, device="cuda")
    dw_ref, db_ref = torch.zeros(ncls, c0, 1, 3, 3, device="cuda"), torch.zeros(ncls, device="cuda")
    ops.conv3d_wgrad(a_ref, cl(d, h, w, c0), ops.BF16, dl, pl(ncls, d, h, w), ops.F32, dw_ref, db_ref, dims, c0, ncls, (1, 3, 3), ws)
    dw, db = torch.full((ncls, c0, 1, 3, 3), 7.0, device="cuda"), torch.full((ncls,), 7.0, device="cuda")
    ops.outconv_wgrad_bn(y, bnbuf, slope, dl, dw, db, dims, c0, ncls, ws)
    assert float((dw - dw_ref).abs().max()) <= 1e-4 * float(dw_ref.abs().max()) + 1e-7
    assert float((db - db_ref).abs().max()) <= 1e-4 * float(db_ref.abs().max()) + 1e-6
    dw2 = torch.full((ncls, c0, 1, 3, 3), 7.0, device="cuda")
    ops.outconv_wgrad_bn(y, bnbuf, slope, dl, dw2, None, dims, c0, ncls, ws)          # no bias gradient wanted; same bits again
    assert torch.equal(dw2, dw)
    # (c) against float64 torch
    a5 = a_ref.double().cpu().view(n, d, h, w, c0).permute(0, 4, 1, 2, 3)
    w64 = wo.double().requires_grad_(True)
    b64 = bias.double().cpu().requires_grad_(True)
    (F.conv3d(a5, w64, b64, padding=(0, 1, 1)) * dl.double().cpu()).sum().backward()
    assert float((dw.cpu().double() - w64.grad).abs().max()) <= 1e-2 * float(w64.grad.abs().max())
    assert float((db.cpu().double() - b64.grad).abs().max()) <= 1e-2 * float(b64.grad.abs().max()) + 1e-6
    # errors: workspace, unavailable form
    with pytest.raises(RuntimeError):
        ops.outconv_wgrad_bn(y, bnbuf, slope, dl, dw, db, dims, c0, ncls, ws[:16])
    _lib.set_tuning("outconv_fwd_rows", 0)
    try:
        assert ops.outconv_wgrad_bn_ws_bytes(dims, c0, ncls) == 0
        with pytest.raises(ValueError):
            ops.outconv_fwd_bn(y, bnbuf, slope, None, wof, bias, lg, dims, c0, ncls)
    finally:
        _lib.set_tuning("outconv_fwd_rows", 1)


@pytest.mark.parametrize("shape", [(2, 3, 9, 35, 2), (2, 2, 17, 40, 3), (2, 4, 24, 33, 4), (1, 2, 16, 64, 2), (2, 3, 20, 96, 2)])
def test_outconv_fused_kernels_against_float64_autograd(shape, outconv_dgrad_rows_knob):
    """VERDICT r04 parity hole (a): fplx_outconv_fwd_bn / fplx_outconv_dgrad_bn_reduce / _apply per kernel against float64
    torch autograd of BatchNorm3d(train) -> PReLU -> Conv3d(C0 -> classes, (1,3,3)) (reference unet2d5_dsbn.py:79-81, 293-294,
    307), for 2, 3 and 4 classes, ragged tiles, n = 2 - not against other HIP kernels.  The oracle sees the bf16 values of y and
    the bf16-rounded activation on the way into out_conv (straight-through), as the kernel's LDS tile holds it.
    activation 1e-2 of its max, logits 1e-2 of their range, d gamma / d beta / d slope 1e-2 relative, dy to bf16 rounding."""
    from fplx import ops
    n, d, h, w, ncls = shape
    c0, bf, dims = 32, torch.bfloat16, (n, d, h, w)
    assert ops.outconv_bn_ok(dims, c0, ncls)
    v, y, yg, gamma, beta, slope, bnbuf = _bn_site_setup(n, d, h, w, c0, "oc")
    yd, gd, bd, sd, a_ref = _bn_site_torch(y, gamma, beta, slope, None, 0.0)
    g = torch.Generator().manual_seed(100 + ncls)
    wo = (torch.randn(ncls, c0, 1, 3, 3, generator=g) * 0.1)
    bias = torch.randn(ncls, generator=g)
    a_q = a_ref + (a_ref.detach().float().bfloat16().double() - a_ref.detach())          # value rounded, gradient straight through
    a5 = a_q.view(n, d, h, w, c0).permute(0, 4, 1, 2, 3)
    wo_b = wo.bfloat16().double()             # the data gradient runs on the bf16 pack; the forward on the fp32 one
    lg_ref = F.conv3d(a5, wo.double(), bias.double(), padding=(0, 1, 1))
    wof, _ = ops.pack_conv_weight(wo.cuda(), torch.float32, False)
    _, wob = ops.pack_conv_weight(wo.cuda(), bf, True)
    # ---- forward: activation + logits
    a = torch.full((v, c0), 7.0, dtype=bf, device="cuda")
    lg = torch.full((n, ncls, d, h, w), 7.0, device="cuda")
    ops.outconv_fwd_bn(yg, bnbuf, slope.cuda(), a, wof, bias.cuda(), lg, dims, c0, ncls)
    assert float((a.float().cpu().double() - a_ref.detach()).abs().max()) < 1e-2 * float(a_ref.detach().abs().max())
    lr = lg_ref.detach()
    assert float((lg.cpu().double() - lr).abs().max()) < 1e-2 * float(lr.max() - lr.min())
    # ---- backward: dlogits -> (recomputed data gradient) -> BatchNorm backward
    dl = (torch.randn(n, ncls, d, h, w, generator=g) * 0.05)
    # the kernel forms out_conv's data gradient with the bf16 pack: differentiate the same function
    lg_b = F.conv3d(a5, wo_b, None, padding=(0, 1, 1))
    a_ref.retain_grad()
    (lg_b * dl.double()).sum().backward()
    part = torch.empty(max(ops.num_partials(v) * (2 * c0 + 1), ops.outconv_bn_rows(dims, c0, ncls) * (2 * c0 + 1)), device="cuda")
    coef = torch.empty((2, c0), device="cuda")
    gf = [torch.zeros(c0, device="cuda"), torch.zeros(c0, device="cuda"), torch.zeros(1, device="cuda")]
    dy = torch.full((v, c0), 7.0, dtype=bf, device="cuda")
    ops.outconv_dgrad_bn_bwd(dl.cuda(), wob, yg, bnbuf, slope.cuda(), True, gf[0], gf[1], gf[2], part, coef, dy, dims, c0, ncls)
    dx_ref = yd.grad
    assert float((dy.float().cpu().double() - dx_ref).abs().max()) < 2e-2 * float(dx_ref.abs().max())
    for got, ref in ((gf[0], gd.grad), (gf[1], bd.grad)):
        assert float((got.cpu().double() - ref).abs().max()) < 1e-2 * float(ref.abs().max()), (got, ref)
    # the slope gradient is ONE number, sum_i d a_i min(z_i, 0), that cancels: bound on the cancellation scale S
    with torch.no_grad():
        z = (yd - yd.mean(0)) * torch.rsqrt(yd.var(0, unbiased=False) + 1e-5) * gd + bd
        S = float((a_ref.grad.abs() * z.clamp(max=0).abs()).sum())
    assert abs(float(gf[2].cpu().double() - sd.grad)) <= 1e-2 * S, (gf[2], sd.grad, S)


def test_fused_out_conv_backward_on_a_narrow_network_with_many_level0_tiles():
    """a network with ft[0] = 32 and max(feature_chns) <= 64 on a volume of >= 2025 level-0 tiles: the
    fused out_conv backward needs outconv_bn_rows x 65 floats of partial rows, more than num_partials x (4 max(ft) + 1) - the
    engine sizes the buffer for both; the step trains (finite loss, parameters move) and equals the unfused engine's step."""
    import fplx
    from fplx import ops
    p = dict(in_chns=1, feature_chns=[32, 32, 64, 64, 64], dropout=[0, 0, 0, 0, 0], conv_dims=[3] * 5, class_num=2,
             bilinear=False, num_domains=2, net_type="UNet2D5_dsbn", precision="bf16")
    shape = (1, 1, 16, 192, 224)
    dims = (shape[0],) + shape[2:]
    vox = dims[0] * dims[1] * dims[2] * dims[3]
    from fplx import _lib
    _lib.set_tuning("outconv_dgrad_rows", 0)          # the tile kernel: one partial row per tile block, up to 2048
    try:
        assert ops.outconv_bn_rows(dims) * 65 > ops.num_partials(vox) * (4 * 64 + 1)          # the case the old sizing missed
        _fused_out_conv_narrow_network_case(p, shape)
    finally:
        _lib.set_tuning("outconv_dgrad_rows", 1)
    _fused_out_conv_narrow_network_case(p, shape)         # and with the row kernel (at most 1024 rows)


def _fused_out_conv_narrow_network_case(p, shape):
    import fplx
    g = torch.Generator().manual_seed(5)
    x = torch.randn(shape, generator=g).cuda()
    lab = torch.zeros((1, 2) + shape[2:])
    lab[:, 0] = 1.0
    lab[:, 0, 4:10, 40:90, 60:140] = 0.0
    lab[:, 1, 4:10, 40:90, 60:140] = 1.0
    lab = lab.cuda()
    res = []
    for fuse in (True, False):
        torch.manual_seed(3)
        net = fplx.UNet2D5_dsbn(dict(p)).cuda()
        net.engine.use_outconv_fusion = fuse
        net.train()
        net._ensure_flat()
        before = net.flat_params.detach().clone()
        ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-3, weight_decay=1e-5)
        out = ts.step(x, lab, 0)
        assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(net.flat_params).all())
        assert float((net.flat_params - before).abs().max()) > 1e-4
        res.append((net.flat_params.detach().clone(), float(out[0])))
    assert abs(res[0][1] - res[1][1]) < 1e-6
    rel = float((res[0][0] - res[1][0]).abs().max()) / float(res[1][0].abs().max())
    assert rel < 2e-3, rel


def test_network_step_with_and_without_the_out_conv_fusion():
    """the engine with out_conv fused into the last site's BatchNorm passes (Engine.use_outconv_fusion, the default) against the
    separate passes: same logits bit for bit, the parameters after one Adam step within the noise of another order of
    additions in three BatchNorm sums"""
    import fplx
    p = dict(in_chns=1, feature_chns=[32, 64, 128, 256, 512], dropout=[0, 0, 0.3, 0.4, 0.5], conv_dims=[3] * 5, class_num=2,
             bilinear=False, num_domains=2, net_type="UNet2D5_dsbn", precision="bf16")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 1, 16, 32, 64, generator=g).cuda()
    lab = torch.zeros(2, 2, 16, 32, 64)
    lab[:, 0] = 1.0
    lab[:, 0, 4:10, 8:20, 16:40] = 0.0
    lab[:, 1, 4:10, 8:20, 16:40] = 1.0
    lab = lab.cuda()
    res = []
    # (fused, out_conv's weight gradient from the pre-BatchNorm tensor = the last activation never stored - the default) |
    # (fused, activation stored) | separate passes
    for fuse, wg in ((True, True), (True, False), (False, False)):
        torch.manual_seed(3)
        net = fplx.UNet2D5_dsbn(dict(p)).cuda()
        net.engine.use_outconv_fusion = fuse
        net.engine.use_outconv_wgrad_bn = wg
        net.train()
        with torch.no_grad():
            net.dropout_seed, net._fwd_counter = 9, 0
        ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-3, weight_decay=1e-5)
        logits, sv = net.engine.forward(x, 1, True, net.dropout_active(), 9, 0, keep=True)
        assert bool(sv.oc_fused) == fuse and bool(sv.oc_wg) == wg and (sv.blocks[8]["out"] is None) == wg
        out = ts.step(x, lab, 1)
        res.append((logits.clone(), net.flat_params.detach().clone(), float(out[0])))
    for k in (1, 2):
        assert torch.equal(res[0][0], res[k][0])
        assert abs(res[0][2] - res[k][2]) < 1e-6
        rel = float((res[0][1] - res[k][1]).abs().max()) / float(res[k][1].abs().max())
        assert rel < 2e-3, rel           # one Adam step of lr 1e-3: a sign flip of a near-zero gradient moves a parameter by 2 lr

"""-m gpu: the 2.5D variant of the shipped configs (conv_dims = [2, 2, 3, 3, 3], config_dual/data_vs/vs_t1s_g.cfg:58):
the (1,2,2) pooling / transposed-convolution kernels, the 2D weight pack / gradient extraction, and the whole network and
train step against fixtures produced by RUNNING the reference in that configuration (tests/golden/make_golden_25d.py)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import detdata
from make_golden_cfg import NETS, SHAPES, key_for
from util import load_det_weights

pytestmark = pytest.mark.gpu


def cl(t):          # [N,C,D,H,W] -> [V, C]
    return t.permute(0, 2, 3, 4, 1).reshape(-1, t.shape[1]).contiguous()


def uncl(t2, n, d, h, w):
    return t2.view(n, d, h, w, -1).permute(0, 4, 1, 2, 3)


@pytest.mark.parametrize("dtype,chans", [(torch.float32, (16, 24)), (torch.bfloat16, (16, 24)),
                                         (torch.bfloat16, (64, 32)), (torch.bfloat16, (32, 64))])   # the last two: MFMA paths
def test_maxpool122_and_deconv122_match_torch(dtype, chans):
    from fplx import ops
    n, d, h, w = 2, 5, 8, 12
    c, co = chans
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    x = torch.from_numpy(detdata.normal("p122.x", (n, c, d, h, w)))
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    fold = lambda t: t.transpose(1, 2).reshape(t.shape[0] * t.shape[2], t.shape[1], t.shape[3], t.shape[4])
    unfold = lambda t, n_: t.reshape(n_, t.shape[0] // n_, t.shape[1], t.shape[2], t.shape[3]).transpose(1, 2)
    # MaxPool2d(2) on the depth-folded tensor (unet2d5_dsbn.py:104, 110-127) and its backward with a skip gradient
    xr = x.clone().requires_grad_(True)
    yr = unfold(F.max_pool2d(fold(xr), 2, 2), n)
    dy = torch.from_numpy(detdata.normal("p122.dy", tuple(yr.shape)))
    dsk = torch.from_numpy(detdata.normal("p122.ds", (n, c, d, h, w)))
    yr.backward(dy)
    xg = cl(x).to(dtype).cuda()
    y = torch.empty((n * d * (h // 2) * (w // 2), c), dtype=dtype, device="cuda")
    ops.maxpool2_fwd(xg, y, (n, d, h, w), c, pd=1)
    assert torch.equal(uncl(y.float().cpu(), n, d, h // 2, w // 2), yr.detach())
    dx = torch.empty_like(xg)
    ops.maxpool2_bwd(xg, cl(dy).to(dtype).cuda(), cl(dsk).to(dtype).cuda(), dx, (n, d, h, w), c, pd=1)
    got = uncl(dx.float().cpu(), n, d, h, w)
    ref = xr.grad + dsk
    assert float((got - ref).abs().max()) < max(tol, 1e-6) * float(ref.abs().max()) * (1 if dtype == torch.float32 else 2)
    # ConvTranspose2d(k=2, s=2) on the depth-folded tensor (unet2d5_dsbn.py:151, 179): forward, dgrad, wgrad, bias grad
    wt = torch.from_numpy(detdata.normal("d122.w", (c, co, 2, 2), 0.3))
    b = torch.from_numpy(detdata.normal("d122.b", (co,)))
    if dtype == torch.bfloat16:
        wt = wt.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = unfold(F.conv_transpose2d(fold(xr), wr, br, stride=2), n)
    dy = torch.from_numpy(detdata.normal("d122.dy", tuple(yr.shape)))
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    yr.backward(dy)
    wf, wb = ops.pack_deconv_weight(wt.cuda(), dtype)
    assert wf.shape == (4, co, c) and wb.shape == (4, c, co)
    yg = torch.empty((n * d * 2 * h * 2 * w, co), dtype=dtype, device="cuda")
    ops.deconv2_fwd(xg, wf, b.cuda(), yg, (n, d, h, w), c, co, sd=1)
    assert float((uncl(yg.float().cpu(), n, d, 2 * h, 2 * w) - yr.detach()).abs().max()) < tol * float(yr.abs().max())
    dyg = cl(dy).to(dtype).cuda()
    dxg = torch.empty_like(xg)
    ops.deconv2_dgrad(dyg, wb, dxg, (n, d, h, w), c, co, sd=1)
    assert float((uncl(dxg.float().cpu(), n, d, h, w) - xr.grad).abs().max()) < tol * float(xr.grad.abs().max())
    ws = torch.empty(ops.deconv2_wgrad_ws_bytes((n, d, h, w), c, co, sd=1), dtype=torch.uint8, device="cuda")
    dw = torch.empty((c, co, 2, 2), dtype=torch.float32, device="cuda")
    db = torch.empty(co, dtype=torch.float32, device="cuda")
    ops.deconv2_wgrad(xg, dyg, dw, db, (n, d, h, w), c, co, ws, sd=1)
    assert float((dw.cpu() - wr.grad).abs().max()) < max(tol, 1e-4) * float(wr.grad.abs().max())
    assert float((db.cpu() - br.grad).abs().max()) < max(tol, 1e-4) * float(br.grad.abs().max())


@pytest.mark.parametrize("dtype,shape", [(torch.float32, (2, 5, 12, 3, 9, 10)), (torch.bfloat16, (1, 32, 64, 4, 16, 24)),
                                         (torch.bfloat16, (1, 64, 32, 6, 16, 80)), (torch.bfloat16, (2, 64, 64, 6, 16, 80)),
                                         (torch.bfloat16, (1, 128, 64, 3, 8, 9)), (torch.bfloat16, (2, 64, 128, 2, 5, 6)),
                                         # the march kernels' middle-plane mode: Cin 32 / 64, ragged tiles, depth segments
                                         (torch.bfloat16, (1, 32, 32, 5, 16, 64)), (torch.bfloat16, (2, 32, 64, 21, 20, 70)),
                                         (torch.bfloat16, (1, 64, 96, 9, 24, 64)), (torch.bfloat16, (1, 64, 32, 4, 9, 70)),
                                         # Cin = 128 march (streamed weights: only the nine live taps are fetched)
                                         (torch.bfloat16, (1, 128, 64, 5, 16, 64)), (torch.bfloat16, (2, 128, 32, 9, 20, 70))])
def test_conv2d_through_the_3d_kernels(dtype, shape):
    """Conv2d(3x3) per depth slice = the 3x3x3 kernels on weights packed into the middle depth plane: forward, data
    gradient, and the weight gradient as the middle plane of the 27-tap gradient."""
    from fplx import ops
    n, cin, cout, d, h, w = shape
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    q = (lambda t: t.bfloat16().float()) if dtype == torch.bfloat16 else (lambda t: t)
    x = q(torch.from_numpy(detdata.normal("c2d.x%s" % (shape,), (n, cin, d, h, w))))
    wt = q(torch.from_numpy(detdata.normal("c2d.w%s" % (shape,), (cout, cin, 3, 3), 0.2)))
    b = torch.from_numpy(detdata.normal("c2d.b%s" % (shape,), (cout,)))
    dy = q(torch.from_numpy(detdata.normal("c2d.dy%s" % (shape,), (n, cout, d, h, w))))
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr.unsqueeze(2), b, padding=(0, 1, 1))             # == Conv2d on every depth slice
    yr.backward(dy)
    dt, dims = ops._DT[dtype], (n, d, h, w)
    wf, wb = ops.pack_conv2d_weight(wt.cuda(), dtype)
    assert float(wf[:9].float().abs().max()) == 0 and float(wf[18:].float().abs().max()) == 0
    xg, dyg = cl(x).to(dtype).cuda(), cl(dy).to(dtype).cuda()
    y = torch.empty((xg.shape[0], cout), dtype=dtype, device="cuda")
    ops.conv3d_fwd(xg, ops.cl_strides(d, h, w, cin), dt, wf, b.cuda(), y, ops.cl_strides(d, h, w, cout), dt, dims, cin, cout,
                   (3, 3, 3), None)
    assert float((uncl(y.float().cpu(), n, d, h, w) - yr.detach()).abs().max()) < tol * float(yr.abs().max())
    # the fplx_conv2d_* form (hint: only the middle plane is live -> 9 taps where the tile kernel applies), + statistics
    rows = ops.conv3d_stats_rows(dims, cin, cout, (3, 3, 3), dt, dt, mid=True)
    stats = torch.zeros((rows, 2, cout), dtype=torch.float32, device="cuda")
    y2 = torch.empty_like(y)
    ops.conv3d_fwd(xg, ops.cl_strides(d, h, w, cin), dt, wf, b.cuda(), y2, ops.cl_strides(d, h, w, cout), dt, dims, cin,
                   cout, (3, 3, 3), stats, mid=True)
    assert float((uncl(y2.float().cpu(), n, d, h, w) - yr.detach()).abs().max()) < tol * float(yr.abs().max())
    yf = cl(yr.detach())
    np.testing.assert_allclose(stats.sum(0)[0].cpu().numpy(), yf.sum(0).numpy(),
                               atol=tol * float(yr.abs().max()) * yf.shape[0] ** 0.5 + 1e-3)
    np.testing.assert_allclose(stats.sum(0)[1].cpu().numpy(), (yf * yf).sum(0).numpy(), rtol=max(tol, 1e-4) * 4)
    dx = torch.empty_like(xg)
    ops.conv3d_fwd(dyg, ops.cl_strides(d, h, w, cout), dt, wb, None, dx, ops.cl_strides(d, h, w, cin), dt, dims, cout, cin,
                   (3, 3, 3), None)
    assert float((uncl(dx.float().cpu(), n, d, h, w) - xr.grad).abs().max()) < tol * float(xr.grad.abs().max())
    ws = torch.empty(ops.conv3d_wgrad_ws_bytes(dims, cin, cout, (3, 3, 3)), dtype=torch.uint8, device="cuda")
    dw27 = torch.empty((cout, cin, 27), dtype=torch.float32, device="cuda")
    dw9 = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device="cuda")
    ops.conv3d_wgrad(xg, ops.cl_strides(d, h, w, cin), dt, dyg, ops.cl_strides(d, h, w, cout), dt, dw27, None, dims, cin,
                     cout, (3, 3, 3), ws)
    dw9.copy_(dw27[:, :, 9:18].reshape(cout, cin, 3, 3))          # the middle depth plane of the 27-tap gradient
    assert float((dw9.cpu() - wr.grad).abs().max()) < max(tol, 1e-4) * float(wr.grad.abs().max())
    # the fplx_conv2d_wgrad form: 9-tap gradient directly (+ bias gradient); MFMA stream kernel in middle-plane mode
    ws2 = torch.empty(ops.conv2d_wgrad_ws_bytes(dims, cin, cout), dtype=torch.uint8, device="cuda")
    dw9b = torch.full((cout, cin, 3, 3), 7.0, dtype=torch.float32, device="cuda")
    db = torch.empty(cout, dtype=torch.float32, device="cuda")
    ops.conv2d_wgrad(xg, ops.cl_strides(d, h, w, cin), dt, dyg, ops.cl_strides(d, h, w, cout), dt, dw9b, db, dims, cin, cout,
                     ws2)
    assert float((dw9b.cpu() - wr.grad).abs().max()) < max(tol, 1e-4) * float(wr.grad.abs().max())
    assert float((db.cpu() - cl(dy).sum(0)).abs().max()) < max(tol, 1e-4) * float(cl(dy).sum(0).abs().max())
    if dtype == torch.bfloat16 and cin in (64, 128) and w >= 64:
        # the Cin = 64 / 128 depth march in Conv2d mode has two forms: conv_fwd_march64_lw (loader waves, the default) and
        # conv_fwd_march64 (knob march64_lw = 0) - same products in the same order: the same bits, statistics included
        from fplx import _lib
        assert _lib.get_tuning("march64_lw") == 1
        _lib.set_tuning("march64_lw", 0)
        try:
            y3, stats3 = torch.empty_like(y), torch.zeros_like(stats)
            ops.conv3d_fwd(xg, ops.cl_strides(d, h, w, cin), dt, wf, b.cuda(), y3, ops.cl_strides(d, h, w, cout), dt, dims, cin,
                           cout, (3, 3, 3), stats3, mid=True)
            dx3 = torch.empty_like(dx)
            ops.conv3d_fwd(dyg, ops.cl_strides(d, h, w, cout), dt, wb, None, dx3, ops.cl_strides(d, h, w, cin), dt, dims, cout, cin,
                           (3, 3, 3), None, mid=True)
        finally:
            _lib.set_tuning("march64_lw", 1)
        assert torch.equal(y3, y2) and torch.equal(stats3, stats)
        dx2 = torch.empty_like(dx)
        ops.conv3d_fwd(dyg, ops.cl_strides(d, h, w, cout), dt, wb, None, dx2, ops.cl_strides(d, h, w, cin), dt, dims, cout, cin,
                       (3, 3, 3), None, mid=True)
        assert torch.equal(dx3, dx2)
        assert float((uncl(dx2.float().cpu(), n, d, h, w) - xr.grad).abs().max()) < tol * float(xr.grad.abs().max())


# (fplx_conv3d_cat2_ok: 32 + 32 -> 32 channels, W >= 64, at least 32000 voxels per sample)
@pytest.mark.parametrize("shape", [(1, 32, 16, 32, 64), (2, 32, 12, 48, 96), (1, 32, 9, 56, 72), (3, 32, 5, 88, 80)])
def test_conv2d_wgrad_two_tensors_and_depth_segments(shape):
    """the 2.5D decoder's weight gradient on the two halves of a concatenation (x0 | x1, never materialised) and the 9-tap
    rolling-window kernel over several depth segments and ragged footprints - against autograd of the Conv2d
    applied per depth slice; the 9-tap result equals the middle plane of the 27-tap kernel's result on the same operands"""
    from fplx import ops
    n, cout, d, h, w = shape
    cin, tol = 64, 2e-2
    dims = (n, d, h, w)
    q = lambda t: t.bfloat16().float()
    x0 = q(torch.from_numpy(detdata.normal("c2d2.x0%s" % (shape,), (n, 32, d, h, w))))
    x1 = q(torch.from_numpy(detdata.normal("c2d2.x1%s" % (shape,), (n, 32, d, h, w))))
    dy = q(torch.from_numpy(detdata.normal("c2d2.dy%s" % (shape,), (n, cout, d, h, w))))
    wr = torch.zeros(cout, cin, 3, 3).requires_grad_(True)
    F.conv3d(torch.cat([x0, x1], 1), wr.unsqueeze(2), None, padding=(0, 1, 1)).backward(dy)
    bf = torch.bfloat16
    g0, g1, dyg = cl(x0).to(bf).cuda(), cl(x1).to(bf).cuda(), cl(dy).to(bf).cuda()
    ws = torch.empty(max(ops.conv3d_wgrad_ws_bytes(dims, cin, cout, (3, 3, 3)), 16), dtype=torch.uint8, device="cuda")
    dw9 = torch.full((cout, cin, 3, 3), 7.0, dtype=torch.float32, device="cuda")
    ops.conv3d_wgrad_cat2(g0, g1, dyg, dw9, dims, cin, cout, ws, mid=True)
    gmax = float(wr.grad.abs().max())
    assert float((dw9.cpu() - wr.grad).abs().max()) < tol * gmax
    # the same layer through the single-tensor entry point on the materialised concatenation: the same kernel, the same bits
    cat = torch.cat([g0, g1], 1).contiguous()
    dt = ops._DT[bf]
    dw9b = torch.empty_like(dw9)
    ws2 = torch.empty(max(ops.conv2d_wgrad_ws_bytes(dims, cin, cout), 16), dtype=torch.uint8, device="cuda")
    ops.conv2d_wgrad(cat, ops.cl_strides(d, h, w, cin), dt, dyg, ops.cl_strides(d, h, w, cout), dt, dw9b, None, dims, cin, cout,
                     ws2)
    assert torch.equal(dw9, dw9b)
    # and the 27-tap kernel's middle plane (another kernel, another order of additions)
    dw27 = torch.empty((cout, cin, 27), dtype=torch.float32, device="cuda")
    ops.conv3d_wgrad_cat2(g0, g1, dyg, dw27, dims, cin, cout, ws)
    mid = dw27.view(cout, cin, 3, 3, 3)[:, :, 1]
    assert float((mid - dw9).abs().max()) < 1e-3 * gmax


def test_25d_training_all_matches_reference(golden_dir):
    """five iterations of SegmentationAgent.training_all (pixel / image weights on domain 1) in the shipped
    dimensionality pattern: loss, class Dice, lr schedule, parameters after 1 and 3 Adam steps."""
    import fplx
    g = np.load(os.path.join(golden_dir, "train_step25.npz"))
    name = "tiny25"
    tcfg = {"dis": False, "train_fpl_uda": True, "loss_type": "DiceLoss", "optimizer": "Adam", "learning_rate": 1e-3,
            "momentum": 0.9, "weight_decay": 1e-5, "lr_scheduler": "MultiStepLR", "lr_gamma": 0.5, "lr_milestones": [2, 4],
            "iter_valid": 1, "gpus": [0]}
    cfg = {"dataset": {"tensor_type": "float"}, "network": dict(NETS[name]), "training": tcfg, "testing": {}}
    agent = fplx.SegmentationAgent(cfg, "train")
    agent.create_network()
    load_det_weights(agent.net, cfg["network"], "cuda")
    agent.create_optimizer()
    agent.create_loss_calculator()
    n, _, D, H, W = SHAPES[name]
    batches = []
    for dom in (0, 1):
        b = {"image": torch.from_numpy(detdata.normal("ts25.x.d%d" % dom, SHAPES[name])),
             "label_prob": torch.from_numpy(detdata.ball_label((D, H, W), 5.0, n=n, offsets=[(dom, 1, -2), (1, -3, 2 + dom)]))}
        if dom == 1:
            pw = (detdata.uniform("ts25.pw", (n, 1, D, H, W)) > 0.25).astype(np.float32)
            iw = np.array([0.8, 0.45], np.float32)
            b["pixel_weight"] = torch.from_numpy(pw * iw[:, None, None, None, None])
            b["image_weight"] = torch.from_numpy(iw)
        batches.append(b)
    agent.set_loaders([batches[0]], [batches[1]])
    lrs = []
    for step in range(1, 6):
        sc = agent.training_all()
        assert abs(sc["loss"] - float(g["step%d.loss" % step])) < 5e-5, (step, sc["loss"])
        np.testing.assert_allclose(sc["class_dice"], g["step%d.class_dice" % step], atol=3e-3)
        lrs.append(agent.optimizer.param_groups[0]["lr"])
        if step in (1, 3):
            sd = agent.net.state_dict()
            pre = "step%d." % step
            for k in g.files:
                if not k.startswith(pre) or k[len(pre):] not in sd:
                    continue
                kk = k[len(pre):]
                if kk.endswith("bias") and ("conv3d_" in kk or "conv2d_" in kk):
                    continue                  # conv bias under BN: see test_gpu_loss_filter_parity
                # Adam normalises every gradient element to +-lr: an element whose true gradient is ~0 (level 4 holds 32
                # voxels here, its BatchNorm gradients are fp32 noise in the reference too) may take the other sign,
                # i.e. differ by 2 lr per step - allowed for under 1 % of a tensor
                ref = g[k]
                diff = np.abs(sd[kk].cpu().numpy() - ref)
                assert diff.max() <= 2.1e-3 * step + 1e-6, (k, diff.max())
                # (the 72-element first-layer weight: its gradient moves by 0.2 % of its max in the reference itself when
                # only the summation order changes, see tests/test_oracle_golden.py GRAD_TOL)
                # 1e-4 per step: the single-step gradients agree to 1 % of their max (test_gpu_net_parity[tiny25]); from the
                # second step on Adam turns that into lr x (a few %) per element and step
                assert (diff <= 1e-4 * step + 1e-4 * np.abs(ref)).mean() >= (0.97 if ref.size >= 1000 else 0.95), k
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-12)


def test_25d_bf16_engine_step_and_checkpoint_keys(tmp_path):
    """bf16 engine mode on a VS-config-shaped crop (depth 28 is not a multiple of 16: only the two 3D levels halve it):
    steps run, the loss goes down, the run is reproducible, and the saved state carries the reference's 484 keys."""
    import fplx
    from fplx import checkpoint as C
    p = dict(in_chns=1, feature_chns=[16, 32, 64, 64, 64], dropout=[0.0, 0.0, 0.3, 0.4, 0.5], conv_dims=[2, 2, 3, 3, 3],
             class_num=2, bilinear=False, num_domains=2, precision="bf16")

    def run():
        torch.manual_seed(2)
        net = fplx.UNet2D5_dsbn(dict(p)).cuda()
        ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-2, weight_decay=1e-5)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(2, 1, 28, 64, 80, generator=g).cuda()
        lab = torch.zeros(2, 2, 28, 64, 80)
        lab[:, 0] = 1.0
        lab[:, 0, 8:20, 20:44, 30:60] = 0.0
        lab[:, 1, 8:20, 20:44, 30:60] = 1.0
        lab = lab.cuda()
        losses = [float(ts.step(x, lab, i % 2)[0].item()) for i in range(8)]
        return net, losses

    net, la = run()
    _, lb = run()
    assert la == lb and all(np.isfinite(la)) and min(la[4:]) < la[0]
    with pytest.raises(ValueError):
        net(torch.zeros(1, 1, 30, 64, 80).cuda(), domain_label=torch.zeros(1, dtype=torch.long))   # depth 30 % 4 != 0
    msd = C.reference_model_state_dict(net)
    assert list(msd.keys()) == C.reference_state_keys(2)
    assert tuple(msd["block0.conv.conv3d_1.weight"].shape) == (16, 1, 3, 3, 3)       # dead 3D twin of a 2D level
    assert tuple(msd["up4.trans3d.weight"].shape) == (32, 16, 2, 2, 2) and tuple(msd["up4.trans2d.weight"].shape) == (32, 16, 2, 2)


SHIPPED_STYLE_CFG = """
[dataset]
tensor_type = float
task_type = seg
root_dir  = {root}
1_train_csv = {root}/train_1.csv
1_valid_csv = {root}/valid_1.csv
2_train_csv = {root}/train_2.csv
2_valid_csv = {root}/valid_2.csv
test_csv  = {root}/test.csv
train_batch_size = 2
modal_num = 1

# data transforms (the chain of config_dual/data_vs/vs_t1s_g.cfg)
train_transform = [NormalizeWithMeanStd, Pad, RandomCrop, RandomFlip, LabelToProbability]
valid_transform = [NormalizeWithMeanStd, Pad, LabelToProbability]
test_transform  = [NormalizeWithMeanStd, Pad]
NormalizeWithMeanStd_channels = [0]
NormalizeWithMeanStd_mean = None
NormalizeWithMeanStd_std  = None
NormalizeWithMeanStd_inverse = False
Pad_output_size = [12, 32, 48]
Pad_ceil_mode   = False
Pad_inverse     = True
RandomCrop_output_size = [12, 32, 32]
RandomCrop_foreground_focus = True
RandomCrop_foreground_ratio = 0.5
Randomcrop_mask_label       = [1, 2]
RandomCrop_inverse     = False
RandomFlip_flip_depth  = False
RandomFlip_flip_height = True
RandomFlip_flip_width  = True
RandomFlip_inverse     = False
LabelToProbability_class_num = 2
LabelToProbability_inverse   = False

[network]
net_type = UNet2D5_dsbn
num_domains = 2
class_num     = 2
in_chns       = 1
feature_chns  = [8, 16, 32, 32, 32]
conv_dims     = [2, 2, 3, 3, 3]
dropout       = [0.0, 0.0, 0.3, 0.4, 0.5]
bilinear      = False
deep_supervise = False
aes = False

[training]
aes = False
train_fpl_uda  = True
dis = False
val_t1 = False
val_t2 = True
dual = False
gpus       = [0]
loss_type     = DiceLoss
optimizer     = Adam
learning_rate = 1e-2
momentum      = 0.9
weight_decay  = 1e-5
lr_scheduler  = MultiStepLR
lr_gamma      = 0.5
lr_milestones = [10000, 20000]
ckpt_save_dir    = {root}/model_dual/vs_t1s_g
ckpt_save_prefix = ignored_like_in_the_reference
iter_start = 0
iter_max   = 4
iter_valid = 2
iter_save  = 2
early_stop_patience = 100

[testing]
fpl = False
gpus       = [0]
domian_label = 1
ae = None
ckpt_mode         = 1
output_dir        = {root}/results_dual/
evaluation_mode   = True
test_time_dropout = False
tta_mode = 1
sliding_window_enable = True
sliding_window_size   = [12, 32, 32]
sliding_window_stride = [12, 32, 32]

[evaluation]
metric_1 = dice
metric_2 = assd
label_list = [1]
organ_name = tumor
ground_truth_folder_root = {root}/lab
test_evaluation_image_pair  = {root}/pair.csv
valid_evaluation_image_pair = {root}/pair.csv
"""


def test_shipped_style_cfg_runs_train_test_evaluate(tmp_path):
    """`python -m fplx.net_run train config.cfg` on a .cfg with the key surface of config_dual/data_vs/vs_t1s_g.cfg
    (2.5D network, dual = False, val_t2, TTA + sliding window, ckpt_mode 1, Dice + ASSD evaluation) over synthetic
    .nii.gz cases: training with validation and checkpoints, then inference with the best checkpoint, uint8 masks with
    the inputs' geometry, and the Dice report."""
    from fplx import net_run, nifti
    root = tmp_path
    rs = np.random.RandomState(9)
    (root / "img").mkdir()
    (root / "lab").mkdir()
    names = []
    for i in range(5):
        shp = (12, 30 + 2 * (i % 2), 44)
        lab = np.zeros(shp, np.uint8)
        lab[3:9, 8 + i:20 + i, 10:30] = 1
        img = rs.randn(*shp) * 15 + 90 + 70.0 * lab
        nifti.write_nifti(str(root / "img" / ("c%d.nii.gz" % i)), img.astype(np.float32), (0.5, 0.5, 1.5))
        nifti.write_nifti(str(root / "lab" / ("c%d.nii.gz" % i)), lab, (0.5, 0.5, 1.5))
        names.append("c%d.nii.gz" % i)
    rows = ["img/%s,lab/%s" % (n, n) for n in names]
    (root / "train_1.csv").write_text("image,label\n" + "\n".join(rows[:3]) + "\n")
    (root / "train_2.csv").write_text("image,label,pixel_weight,image_weight\n" + "\n".join(
        "%s,lab/%s,0.9" % (r, n) for r, n in zip(rows[2:], names[2:])) + "\n")        # weights: the label volume itself
    (root / "valid_1.csv").write_text("image,label\n" + rows[0] + "\n")
    (root / "valid_2.csv").write_text("image,label\n" + rows[4] + "\n")
    (root / "test.csv").write_text("image\n" + "\n".join("img/" + n for n in names[3:]) + "\n")
    (root / "pair.csv").write_text("ground_truth,segmentation\n" + "\n".join("%s,%s" % (n, n) for n in names[3:]) + "\n")
    cfg = root / "vs_like.cfg"
    cfg.write_text(SHIPPED_STYLE_CFG.format(root=str(root)))
    import random
    random.seed(3)
    res = net_run.main(["fplx.net_run", "train", str(cfg)])
    ck = root / "model_dual" / "vs_t1s_g"
    files = sorted(os.listdir(str(ck)))
    assert "vs_t1s_g_latest.txt" in files and "vs_t1s_g_best.txt" in files and "log_train.txt" in files
    assert any(f.endswith(".pt") for f in files)
    out = root / "results_dual" / "vs_t1s_g_test"
    for n in names[3:]:
        m = nifti.load_nifty_volume_as_4d_array(str(out / n))
        ref = nifti.load_nifty_volume_as_4d_array(str(root / "img" / n))
        assert m["data_array"].dtype == np.uint8 and m["data_array"].shape == ref["data_array"].shape
        assert m["spacing"] == ref["spacing"]
    rep = (out / "test_tumor_dice_all.csv").read_text().strip().splitlines()
    assert rep[0] == "image,class_1" and rep[-2].startswith("mean,") and rep[-1].startswith("std,")
    assert res["test"][0].shape == (1,) and 0.0 <= float(res["test"][0][0]) <= 1.0
    rep2 = (out / "test_tumor_assd_all.csv").read_text().strip().splitlines()    # evaluation_2: metric_2 = assd
    assert rep2[0] == "image,class_1" and len(rep2) == len(rep) and rep2[-2].startswith("mean,")
    for line in rep2[1:-2]:
        v = float(line.split(",")[1])
        assert np.isnan(v) or 0.0 <= v <= 50.0               # nan: neither volume has foreground (the reference's 0 / 0)
    assert (out / "valid_tumor_assd_all.csv").exists()


def test_25d_split_concat_path_agrees_with_the_single_buffer_path():
    """shipped channel widths (32 at level 0): the level-0 decoder convolution of a dim-2 level takes skip and up as two
    tensors (fplx_conv3d_*_cat2) and its 27-tap weight gradient is reduced to the middle plane - same gradients as the
    path that materialises the concatenation (FPLX_SPLIT_CAT=0), and the bf16-emulating oracle agrees."""
    import fplx
    from fplx import ops
    from oracle import torch_ref as R
    p = dict(in_chns=1, feature_chns=[32, 64, 64, 64, 64], dropout=[0, 0, 0, 0, 0], conv_dims=[2, 2, 3, 3, 3], class_num=2,
             bilinear=False, num_domains=2, precision="bf16")
    shape = (1, 1, 8, 64, 64)
    assert ops.conv3d_cat2_ok((1, 8, 64, 64), 64, 32)
    x = torch.from_numpy(detdata.normal("sc25.x", shape))
    y = torch.zeros(1, 2, 8, 64, 64)
    y[:, 0] = 1.0
    y[:, 0, 2:6, 20:44, 16:50] = 0.0
    y[:, 1, 2:6, 20:44, 16:50] = 1.0
    grads = []
    for split in (True, False):
        net = fplx.UNet2D5_dsbn(dict(p))
        load_det_weights(net, p, "cuda")
        net.engine.use_split_cat = split
        net.train()
        lt = net(x.cuda(), domain_label=torch.ones(1, dtype=torch.long))
        loss = fplx.DiceLoss()({"prediction": lt, "ground_truth": y.cuda()})
        loss.backward()
        grads.append(({k: t.grad.float().cpu().numpy().astype(np.float64) for k, t in net.named_parameters()
                       if t.grad is not None}, float(loss.item()), lt.detach().float().cpu().numpy()))
    (ga, la, oa), (gb, lb, ob) = grads
    assert abs(la - lb) < 2e-3 and np.abs(oa - ob).max() < 5e-2 * np.abs(ob).max()
    for k in ("up4.conv.conv2d_1.weight", "up4.trans2d.weight", "block0.conv.conv2d_2.weight", "up4.conv.conv2d_2.weight",
              "block1.conv.conv2d_1.weight", "out_conv.weight"):
        a, b = ga[k].reshape(-1), gb[k].reshape(-1)
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
        assert cos > 0.98 and abs(np.linalg.norm(a) / np.linalg.norm(b) - 1) < 0.1, (k, cos)
    sd, prm = R.split_state(detdata.state_dict_3d(p))
    ref = R.unet_forward(sd, p, x, 1, True, act_dtype=torch.bfloat16)
    rl = R.dice_loss(ref, y)
    rl.backward()
    assert abs(la - rl.item()) < 2e-3
    for k in ("up4.conv.conv2d_1.weight", "up4.trans2d.weight", "block0.conv.conv2d_2.weight"):
        r, g = prm[k].grad.numpy().reshape(-1).astype(np.float64), ga[k].reshape(-1)
        cos = float(g @ r / (np.linalg.norm(g) * np.linalg.norm(r)))
        assert cos > 0.9, (k, cos)


@pytest.mark.parametrize("path", ["step_all", "autograd"])
def test_25d_32base_backward_is_bitwise_reproducible_with_the_side_stream(path):
    """The shipped-style network (32-base, conv_dims = [2,2,3,3,3], bf16, 4 x 1 x 28 x 128 x 128) with the weight-gradient
    stream ON: repeated runs of the same iteration give the same bits, through TrainStep.step_all (no reducer hook) and
    through net(x) / loss.backward() (what SegmentationAgent runs)."""
    import fplx
    p = dict(in_chns=1, feature_chns=[32, 64, 128, 256, 512], dropout=[0.0, 0.0, 0.3, 0.4, 0.5], conv_dims=[2, 2, 3, 3, 3],
             class_num=2, bilinear=False, num_domains=2, precision="bf16")
    shape = (4, 1, 28, 128, 128)
    g = torch.Generator().manual_seed(0)
    bs = []
    for dmn in range(2):
        x = torch.randn(*shape, generator=g)
        lab = torch.zeros(4, 2, *shape[2:])
        lab[:, 0] = 1.0
        lab[:, 0, 7:14, 32:64, 42:84] = 0.0
        lab[:, 1, 7:14, 32:64, 42:84] = 1.0
        x[:, 0] += 2.0 * lab[:, 1]
        bs.append({"image": x.cuda(), "label_prob": lab.cuda()})

    def run():
        torch.manual_seed(1)
        net = fplx.UNet2D5_dsbn(dict(p)).cuda()
        assert net.engine.use_side_stream
        if path == "step_all":
            ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-3, weight_decay=1e-5)
            for _ in range(4):
                ts.step_all(bs)
            torch.cuda.synchronize()
            return net.flat_params.clone()
        net.train()
        net._ensure_flat()
        grads = []
        for it in range(4):
            b = bs[it % 2]
            for q in net.parameters():
                q.grad = None
            out = net(b["image"], domain_label=(it % 2) * torch.ones(4, dtype=torch.long))
            fplx.DiceLoss()({"prediction": out, "ground_truth": b["label_prob"]}).backward()
            grads.append(torch.cat([q.grad.reshape(-1) for q in net.parameters() if q.grad is not None]))
        torch.cuda.synchronize()
        return torch.cat(grads)

    a = run()
    for _ in range(3):
        assert torch.equal(run(), a)

"""Properties at BASELINE.json's full sizes (2 x 1 x 80 x 160 x 160 bf16 train step, 48 x 160 x 272 filter volume)
that need no oracle run of that size: bitwise run-to-run reproducibility (fixed-order reductions), training makes
progress, dropout stream determinism, and the filter against the numpy oracle (seconds at this size)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NET = dict(in_chns=1, feature_chns=[32, 64, 128, 256, 512], dropout=[0.0, 0.0, 0.3, 0.4, 0.5],
           conv_dims=[3, 3, 3, 3, 3], class_num=2, bilinear=False, num_domains=2, precision="bf16")
SHAPE = (2, 1, 80, 160, 160)


def _batch(seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(SHAPE, generator=g)
    lab = torch.zeros((2, 2) + SHAPE[2:])
    lab[:, 0] = 1.0
    lab[:, 0, 30:50, 60:100, 60:100] = 0.0
    lab[:, 1, 30:50, 60:100, 60:100] = 1.0
    return x.cuda(), lab.cuda()


def _run(steps, seed=1):
    import fplx
    torch.manual_seed(seed)
    net = fplx.UNet2D5_dsbn(dict(NET)).cuda()
    ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-3, weight_decay=1e-5)
    b = [_batch(10), _batch(11)]
    losses = []
    for i in range(steps):
        out = ts.step(b[i % 2][0], b[i % 2][1], i % 2)
        losses.append(out[0])
    torch.cuda.synchronize()
    return net, [float(l.item()) for l in losses], ts


def test_full_size_train_steps_are_bitwise_reproducible_and_learn():
    net_a, loss_a, _ = _run(6)
    net_b, loss_b, _ = _run(6)
    assert loss_a == loss_b                                  # fixed-order reductions: bit-identical runs
    assert torch.equal(net_a.flat_params, net_b.flat_params)
    assert all(np.isfinite(loss_a))
    assert min(loss_a[4:]) < max(loss_a[:2])                 # Dice loss goes down on both domains
    sd = net_a.state_dict()
    # only the BN sets of the domain used in a step were touched: 3 steps each
    assert int(sd["block0.conv.bn3d1.bns.0.num_batches_tracked"]) == 3
    assert int(sd["block0.conv.bn3d1.bns.1.num_batches_tracked"]) == 3


def test_full_size_out_conv_gradients_without_the_stored_activation():
    """the out_conv weight / bias gradient from the pre-BatchNorm tensor (fplx_outconv_wgrad_bn, reference unet2d5_dsbn.py:
    79-81 + 293-294, 307) at the benchmark's level-0 size 2 x 80 x 160 x 160, through size-independent properties: (a) it equals
    fplx_conv3d_wgrad on the activation fplx_outconv_fwd_bn would have stored (same bf16 operands: 1e-4 of the largest entry);
    (b) linearity in the dlogits: dw(g1 + g2) = dw(g1) + dw(g2) for dlogits that are exact in bf16 (the sums of bf16-exact
    integers stay exact in the operand rounding); (c) bitwise run-to-run reproducibility; (d) the logits of the forward that
    stores no activation are the bits of the one that does."""
    from fplx import ops
    n, d, h, w, ncls, c0 = 2, 80, 160, 160, 2, 32
    dims, v, bf = (n, d, h, w), n * d * h * w, torch.bfloat16
    assert ops.outconv_wgrad_bn_ws_bytes(dims, c0, ncls) > 0
    g = torch.Generator(device="cuda").manual_seed(3)
    y = (torch.randn(v, c0, generator=g, device="cuda") * 0.8 + 0.1).to(bf)
    gamma, beta = torch.rand(c0, generator=g, device="cuda") + 0.5, torch.randn(c0, generator=g, device="cuda") * 0.2
    mean, rstd = torch.randn(c0, generator=g, device="cuda") * 0.1, torch.rand(c0, generator=g, device="cuda") + 0.7
    bnbuf = torch.stack([mean, rstd, gamma * rstd, beta - mean * gamma * rstd])
    slope = torch.tensor([0.25], device="cuda")
    wo = torch.randn(ncls, c0, 1, 3, 3, generator=g, device="cuda") * 0.1
    wof, _ = ops.pack_conv_weight(wo, torch.float32, False)
    bias = torch.randn(ncls, generator=g, device="cuda")
    a = torch.empty(v, c0, dtype=bf, device="cuda")
    lg_a = torch.empty(n, ncls, d, h, w, device="cuda")
    lg = torch.empty(n, ncls, d, h, w, device="cuda")
    ops.outconv_fwd_bn(y, bnbuf, slope, a, wof, bias, lg_a, dims, c0, ncls)
    ops.outconv_fwd_bn(y, bnbuf, slope, None, wof, bias, lg, dims, c0, ncls)
    assert torch.equal(lg, lg_a)                                                       # (d)
    g1 = torch.randint(-8, 9, (n, ncls, d, h, w), generator=g, device="cuda").float() / 64.0
    g2 = torch.randint(-8, 9, (n, ncls, d, h, w), generator=g, device="cuda").float() / 64.0
    ws = torch.empty(max(ops.conv3d_wgrad_ws_bytes(dims, c0, ncls, (1, 3, 3)), ops.outconv_wgrad_bn_ws_bytes(dims, c0, ncls)),
                     dtype=torch.uint8, device="cuda")

    def wg(dl):
        dw, db = torch.empty(ncls, c0, 1, 3, 3, device="cuda"), torch.empty(ncls, device="cuda")
        ops.outconv_wgrad_bn(y, bnbuf, slope, dl, dw, db, dims, c0, ncls, ws)
        return dw, db

    dw1, db1 = wg(g1)
    dw2, db2 = wg(g2)
    dw12, db12 = wg(g1 + g2)
    dw1b, db1b = wg(g1)
    assert torch.equal(dw1, dw1b) and torch.equal(db1, db1b)                           # (c)
    scale = float(dw12.abs().max()) + float(dw1.abs().max())
    assert float((dw12 - (dw1 + dw2)).abs().max()) <= 2e-5 * scale                     # (b): fp32 summation order only
    assert float((db12 - (db1 + db2)).abs().max()) <= 2e-5 * (float(db1.abs().max()) + float(db2.abs().max()) + 1.0)
    dw_ref, db_ref = torch.empty(ncls, c0, 1, 3, 3, device="cuda"), torch.empty(ncls, device="cuda")
    ops.conv3d_wgrad(a, ops.cl_strides(d, h, w, c0), ops.BF16, g1, ops.planar_strides(ncls, d, h, w), ops.F32, dw_ref, db_ref, dims, c0,
                     ncls, (1, 3, 3), ws)
    assert float((dw1 - dw_ref).abs().max()) <= 1e-4 * float(dw_ref.abs().max())      # (a)
    assert float((db1 - db_ref).abs().max()) <= 1e-4 * float(db_ref.abs().max()) + 1e-3


def test_full_size_forward_is_deterministic_under_dropout_seed():
    import fplx
    torch.manual_seed(3)
    net = fplx.UNet2D5_dsbn(dict(NET)).cuda()
    net.eval()
    for m in net.modules():
        if type(m) == torch.nn.Dropout:
            m.train()
    x, _ = _batch(5)
    outs = []
    for rep in range(2):
        net._fwd_counter = 7
        with torch.no_grad():
            outs.append(net(x, domain_label=torch.ones(2, dtype=torch.long)))
    assert torch.equal(outs[0], outs[1])
    with torch.no_grad():
        other = net(x, domain_label=torch.ones(2, dtype=torch.long))       # next forward counter -> new masks
    assert not torch.equal(outs[0], other)


def test_full_volume_filter_matches_numpy_oracle():
    import fplx
    from oracle import np_ref as N
    rng = np.random.default_rng(5)
    stack = (rng.standard_normal((6, 2, 48, 160, 272)) * 1.5).astype(np.float32)
    stack[:, 1, :, :80] -= 6.0                                # confident background half
    ref = N.fpl_filter(stack)
    r = fplx.ops.mc_filter(torch.from_numpy(stack).cuda(), want_maps=True)
    assert np.array_equal(r["hards"].cpu().numpy(), ref["hards"])            # masks: bit exact
    st = r["stats"].cpu().numpy()
    # boundary = #(u > 0.01): decided on the float32 mean m itself (two cut points of numpy's float32 u(m), found by
    # tools/filter_cutpoints.py), so given m the count is numpy's exactly ...
    m_gpu = r["means"].cpu().numpy()
    with np.errstate(divide="ignore"):
        pred_gpu = (-1.0 * (m_gpu * np.log(m_gpu + 1e-6))) > 0.01
    assert int(st[1]) == int(pred_gpu.sum())
    # ... and m is numpy's mean up to the last bits of exp (numpy's float32 exp is not correctly rounded either): a voxel
    # may only be counted differently if its two means straddle a cut point
    flipped = pred_gpu != (ref["uncertainty"] > 0.01)
    assert int(st[1]) - ref["boundary"] == int(pred_gpu.sum()) - int((ref["uncertainty"] > 0.01).sum())
    assert flipped.sum() <= 2
    if flipped.any():
        assert np.abs(m_gpu[flipped] - ref["means"][flipped]).max() <= 4 * np.spacing(ref["means"][flipped]).max()
    assert np.abs(m_gpu - ref["means"]).max() <= 2.5e-7
    assert abs(st[0] - float(ref["vars"])) <= 2e-5 * float(ref["vars"])
    assert abs(st[2] - float(ref["uncer_one"])) <= 2e-5 * float(ref["uncer_one"])
    a = (rng.random((48, 160, 272)) > 0.5).astype(np.uint8)
    b = (rng.random((48, 160, 272)) > 0.5).astype(np.uint8)
    w = fplx.filter.pixel_weight_from_masks(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), 0.37).cpu().numpy()
    assert np.array_equal(w, N.set_weight(np.float32(0.37), N.pixel_weight_from_masks(a.copy(), b.copy()).astype(np.float32)))


def test_agent_fpl_inference_pipeline():
    """SegmentationAgent.infer, FPL branch (reference agent_seg.py:897-961): MC passes with test-time dropout ->
    filter -> ascending (uncertainty, name) list; re-derived from the recorded logits with the numpy oracle."""
    import fplx
    from oracle import np_ref as N
    net_cfg = dict(in_chns=1, feature_chns=[8, 16, 32, 64, 128], dropout=[0, 0, 0.3, 0.4, 0.5], conv_dims=[3] * 5,
                   class_num=2, bilinear=False, num_domains=2, net_type="UNet2D5_dsbn")
    cfg = {"dataset": {"tensor_type": "float"}, "network": net_cfg, "training": {},
           "testing": {"domian_label": 1, "gpus": [0], "fpl": True, "evaluation_mode": True, "tta_mode": 1,
                       "sliding_window_enable": True, "sliding_window_size": [16, 32, 32],
                       "sliding_window_stride": [16, 32, 32]}}
    agent = fplx.SegmentationAgent(cfg, "test")
    torch.manual_seed(0)
    agent.create_network()
    vols = [{"image": torch.randn(1, 1, 16, 64, 32) * (1 + i), "names": ["v%d.nii.gz" % i]} for i in range(3)]
    agent.set_loaders(test_loader=vols)
    recorded = []

    class Rec(fplx.Inferer):
        def run(self, model, image, domain_label):
            r = fplx.Inferer.run(self, model, image, domain_label)
            recorded.append(r.cpu().numpy().copy())
            return r

    icfg = dict(cfg["testing"])
    icfg["class_num"] = 2
    agent.set_inferer(Rec(icfg))
    srt = agent.infer()
    assert len(recorded) == 18 and len(srt) == 3
    assert [s[0][0] for s in srt] == sorted(s[0][0] for s in srt)
    exp = {}
    for i in range(3):
        exp["v%d.nii.gz" % i] = float(N.fpl_filter(np.concatenate(recorded[6 * i:6 * i + 6], 0))["uncer_one"])
    for (u, name) in srt:
        assert abs(u[0] - exp[name]) <= 1e-5 * exp[name] + 1e-14      # (saturated softmax: variances ~1e-13)
    # MC dropout really is active: the six passes of a volume differ
    assert not np.array_equal(recorded[0], recorded[1])


def test_config4_mc_inference_and_filter_at_full_size():
    """BASELINE config 4 at its size (reference agent_seg.py:897-931, infer_func.py:188-222): the 32-base bf16 network in
    the shipped 2.5D pattern (conv_dims [2, 2, 3, 3, 3], config_dual/data_vs/vs_t1s_g.cfg:58 - the all-3D network cannot take
    the shipped 28-slice window: four depth poolings), eval-mode BatchNorm + test-time dropout, one hrT2-sized volume
    1 x 1 x 48 x 160 x 272 through the shipped sliding window (28 x 128 x 128, stride = window: 12 tiles), (a) T = 4 Monte-Carlo
    passes and (b) the reference-literal 6 passes x 4-flip TTA = 24 forwards per tile.  Two runs are bit-identical; the batched plan equals the tile-by-tile loop of forwards (dropout
    off: the masks are keyed by the element index inside a forward batch); hard pseudo-labels and filter scalars are those of
    the numpy oracle on the recorded logits."""
    import fplx
    from oracle import np_ref as N
    torch.manual_seed(4)
    net = fplx.UNet2D5_dsbn(dict(NET, conv_dims=[2, 2, 3, 3, 3])).cuda()
    # a network that has seen data: a few train steps on shipped-size crops (4 x 1 x 28 x 128 x 128) give BatchNorm running
    # statistics and non-trivial logits
    ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-3, weight_decay=1e-5)
    gen = torch.Generator().manual_seed(12)
    xb = torch.randn((4, 1, 28, 128, 128), generator=gen).cuda()
    lb = torch.zeros((4, 2, 28, 128, 128))
    lb[:, 0] = 1.0
    lb[:, 0, 8:20, 40:90, 40:90] = 0.0
    lb[:, 1, 8:20, 40:90, 40:90] = 1.0
    lb = lb.cuda()
    for i in range(2):
        ts.step(xb, lb, 1)
    g = torch.Generator().manual_seed(8)
    x = torch.randn((1, 1, 48, 160, 272), generator=g).cuda()
    dl = torch.ones(1, dtype=torch.long)
    cfg = dict(sliding_window_enable=True, sliding_window_size=[28, 128, 128], sliding_window_stride=[28, 128, 128],
               class_num=2)

    def dropout(on):
        net.eval()
        for m in net.modules():
            if type(m) == torch.nn.Dropout:
                m.train(on)

    with torch.no_grad():
        # ---- batched plan == the literal loop of forwards (deterministic network)
        dropout(False)
        for tta in (0, 1):
            inf = fplx.Inferer(dict(cfg, tta_mode=tta))
            a = inf.run(net, x, dl)
            inf.model = net
            lit = inf._run_generic(x, dl)
            rng = float(lit.abs().max())
            # (bf16 activations: a forward of 12 tiles and a forward of one tile pick different split-K plans at the deep
            # levels, so single bf16 roundings may differ - not the arithmetic)
            assert float((a - lit).abs().max()) <= 3e-2 * rng, tta
            agree = float((fplx.filter.hard_label(a) == fplx.filter.hard_label(lit)).float().mean())
            assert agree >= 0.999, (tta, agree)
        # ---- Monte-Carlo passes at size
        dropout(True)
        net.dropout_seed = 5
        for passes, tta in ((4, 0), (6, 1)):
            inf = fplx.Inferer(dict(cfg, tta_mode=tta))
            runs = []
            for rep in range(2):
                net._fwd_counter = 3
                runs.append(inf.run_mc(net, x, dl, passes)[:, 0].clone())
            assert tuple(runs[0].shape) == (passes, 2, 48, 160, 272)
            assert torch.equal(runs[0], runs[1])                                # bitwise reproducible
            assert not torch.equal(runs[0][0], runs[0][1])                       # the passes differ: dropout is active
            r = fplx.ops.mc_filter(runs[0], want_maps=True)
            ref = N.fpl_filter(runs[0].cpu().numpy())
            assert np.array_equal(r["hards"].cpu().numpy(), ref["hards"])        # pseudo-label masks: bit exact
            st = r["stats"].cpu().numpy()
            m_gpu = r["means"].cpu().numpy()
            with np.errstate(divide="ignore", invalid="ignore"):
                pred_gpu = (-1.0 * (m_gpu * np.log(m_gpu + 1e-6))) > 0.01
            assert int(st[1]) == int(pred_gpu.sum())
            assert int((pred_gpu != (ref["uncertainty"] > 0.01)).sum()) <= 2
            assert abs(st[0] - float(ref["vars"])) <= 2e-5 * float(ref["vars"]) + 1e-12
            if ref["boundary"] >= 50:
                assert abs(st[2] - float(ref["uncer_one"])) <= 2e-5 * float(ref["uncer_one"]) + 1e-12


def test_bench_under_torchrun_exercises_the_rccl_path():
    """bench.py launched the way the driver launches it (torch.distributed.run, one rank per GPU) with
    FPLX_DDP_FORCE=1, so that on this 1-GPU box the RCCL process group, the bucketed asynchronous all-reduces of the
    flat gradient, the barriers and the max-over-ranks timing all really run (world size 1)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FPLX_DDP_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3",
           "--warmup", "1", "--no-kernel-timing", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 10 and np.isfinite(d["final_loss"])
    assert d["repeats"] == 5 and d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    # the record describes what RCCL saw and what the exchange costs: world, version, one device entry per
    # rank gathered to rank 0, the all-reduce of the flat gradient timed with the step's own buckets, the exposed share
    rc = d["rccl"]
    assert rc["world"] == 1 and rc["backend"] == "nccl" and rc["version"] and len(rc["devices"]) == 1
    assert rc["devices"][0]["rank"] == 0 and rc["devices"][0]["device"] and rc["distinct_devices"] == 1
    assert 80 < rc["flat_gradient_MB"] < 100 and rc["buckets"] == len(rc["bucket_MB"]) + 1 >= 3
    assert rc["allreduce_flat_ms"] > 0 and rc["allreduce_90MB_ms"] > 0 and rc["allreduce_buckets_ms"] > 0
    assert rc["busbw_GBps"] == 0.0 and rc["algbw_GBps"] > 10          # one rank: nothing crosses a link
    assert abs(d["exposed_comm_ms"]) < 0.25 * d["ms_per_step"] and d["comm"]["ms_per_step_without_collectives"] > 0
    # the other way: no launcher - `bench.py --gpus 1` runs in-process, and with --launch it starts the rank itself as a child
    # torch.distributed.run (what `bench.py --gpus N`, N > 1, does when RANK is not set) and relays rank 0's line
    env_nolaunch = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    tail = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--repeats", "2", "--no-kernel-timing", "--no-cpu-baseline"]
    for extra in ([], ["--launch"]):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + tail + extra, cwd=root, env=env_nolaunch,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        d = json.loads(lines[0])
        assert d["n_gpus"] == 1 and d["steps"] == 2 and d["repeats"] == 2 and d["value"] > 10
        assert ("rccl" in d) == bool(extra)               # in-process single rank: no process group, no rccl object


def test_config5_full_size_weighted_dual_domain_step():
    """BASELINE config 5 at its full size on one GPU: dual-domain UNet3D-DSBN, in_chns 4, 32-base bf16, 2 x 4 x 128^3 crops
    per domain, weighted Dice + CE with pixel_weight in {0, w_img} (= NiftyDataset.set_weight_ of a {1, 0.5} map,
    nifty_dataset.py:165-168) - `training_all` iterations (both domains, one Adam step each): two runs bit-identical,
    finite, the loss falls.  Timing of the same step: tools/cfg5_bench.py."""
    import fplx
    net_cfg = dict(NET, in_chns=4)
    shape = (2, 4, 128, 128, 128)

    def batches():
        out = []
        for dmn in range(2):
            g = torch.Generator().manual_seed(20 + dmn)
            x = torch.randn(shape, generator=g)
            lab = torch.zeros((2, 2) + shape[2:])
            lab[:, 0] = 1.0
            lab[:, 0, 40:90, 30:100, 50:110] = 0.0
            lab[:, 1, 40:90, 30:100, 50:110] = 1.0
            x[:, :2] += 1.5 * lab[:, 1:2]
            w_img = torch.rand(2, generator=g) + 0.01                                   # U(0.01, 1.01)
            agree = (torch.rand((2, 1) + shape[2:], generator=g) > 0.1).float()          # 10 % of the voxels filtered out
            pw = agree * w_img.view(2, 1, 1, 1, 1)
            out.append({"image": x.cuda(), "label_prob": lab.cuda(), "pixel_weight": pw.cuda(), "image_weight": w_img.cuda()})
        return out

    def run(steps):
        torch.manual_seed(3)
        net = fplx.UNet2D5_dsbn(dict(net_cfg)).cuda()
        loss = fplx.make_loss({"loss_type": ["DiceLoss", "CrossEntropyLoss"], "loss_weight": [0.5, 0.5]})
        ts = fplx.TrainStep(net, loss.terms, True, lr=1e-3, weight_decay=1e-5)
        bs = batches()
        ls = []
        for _ in range(steps):
            outs = ts.step_all(bs)
            ls.append(0.5 * (outs[0][0] + outs[1][0]))
        torch.cuda.synchronize()
        return net, [float(v.item()) for v in ls]

    net_a, la = run(5)
    net_b, lb = run(5)
    assert la == lb and torch.equal(net_a.flat_params, net_b.flat_params)
    assert all(np.isfinite(la)) and bool(torch.isfinite(net_a.flat_params).all())
    assert la[-1] < la[0]

"""GPU parity of the fused loss kernels, the training step, the inferer and the pseudo-label
filter against reference-generated fixtures and the CPU oracle."""
import json
import os
import numpy as np
import pytest
import torch

import detdata
from make_golden_cfg import NETS, SHAPES
from util import load_det_weights

pytestmark = pytest.mark.gpu


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_losses_match_reference(golden_dir):
    import fplx
    g = np.load(os.path.join(golden_dir, "losses.npz"))
    lab, pw, iw = _cuda(g["label"]), _cuda(g["pixel_weight"]), _cuda(g["image_weight"])
    pw_bin = (pw > 0).float()

    def check(tag, mod, extra, logits_key="logits", label=lab):
        lg = _cuda(g[logits_key]).requires_grad_(True)
        d = {"prediction": lg, "ground_truth": label}
        d.update(extra)
        v = mod(d)
        v.backward()
        assert abs(v.item() - float(g[tag + ".loss"])) < 5e-6, (tag, v.item())
        ref = g[tag + ".dlogits"]
        np.testing.assert_allclose(lg.grad.cpu().numpy(), ref, atol=2e-4 * np.abs(ref).max(), rtol=1e-3, err_msg=tag)

    check("dice", fplx.DiceLoss(), {})
    check("dice_pw", fplx.DiceLoss(), {"pixel_weight": pw})
    check("ce", fplx.CrossEntropyLoss(), {})
    check("ce_pw", fplx.CrossEntropyLoss(), {"pixel_weight": pw})
    check("dice_weight", fplx.DiceLoss_weight(), {"pixel_weight": pw_bin, "image_weight": iw})
    comb = fplx.CombinedLoss({"loss_type": ["DiceLoss", "CrossEntropyLoss"], "loss_weight": [0.6, 0.4]},
                             fplx.SegLossDict)
    check("combined", comb, {})
    check("combined_pw", comb, {"pixel_weight": pw})
    check("entropy", fplx.EntropyTerm(), {})
    lab3 = _cuda(g["label3"])
    check("dice3", fplx.DiceLoss(), {}, "logits3", lab3)
    check("ce3", fplx.CrossEntropyLoss(), {}, "logits3", lab3)
    with pytest.raises(ValueError):
        fplx.make_loss({"loss_type": "NoSuchLoss"})
    with pytest.raises(ValueError):
        fplx.CombinedLoss({"loss_type": ["DiceLoss", "Nope"], "loss_weight": [1, 1]}, fplx.SegLossDict)
    with pytest.raises(KeyError):
        fplx.DiceLoss_weight()({"prediction": _cuda(g["logits"]), "ground_truth": lab})


def _batches(with_w):
    name = "tiny"
    n, _, D, H, W = SHAPES[name]
    out = []
    for dom in (0, 1):
        b = {"image": torch.from_numpy(detdata.normal("ts.x.d%d" % dom, SHAPES[name])),
             "label_prob": torch.from_numpy(detdata.ball_label((D, H, W), 5.0, n=n,
                                                               offsets=[(dom, 1, -2), (1, -3, 2 + dom)]))}
        if with_w and dom == 1:
            pw = (detdata.uniform("ts.pw", (n, 1, D, H, W)) > 0.25).astype(np.float32)
            iw = np.array([0.8, 0.45], np.float32)
            b["pixel_weight"] = torch.from_numpy(pw * iw[:, None, None, None, None])
            b["image_weight"] = torch.from_numpy(iw)
        out.append(b)
    return out


def _check_params(sd, g, variant, step):
    for k in g.files:
        pre = "%s.step%d." % (variant, step)
        if k.startswith(pre) and k[len(pre):] in sd:
            kk = k[len(pre):]
            if kk.endswith("bias") and "conv3d" in kk:
                continue      # conv bias under BN: reference random-walks on fp noise (Adam), fplx decays it
            ref = g[k]
            diff = np.abs(sd[kk].cpu().numpy() - ref)
            assert diff.max() <= 1e-3 * step + 1e-6, (k, diff.max())
            assert (diff <= 5e-5 * step + 1e-4 * np.abs(ref)).mean() >= 0.99, k


def test_seg_loss_fwd_stays_inside_the_documented_part_buffer():
    """with N = 1 and an odd rows x K the 8-byte aligned (N + 1) x K doubles behind the partial rows ended one
    float past N x fplx_loss_rows x K.  Sub-allocate exactly what include/fplx.h documents, poison what follows."""
    from fplx import ops
    from oracle import torch_ref as R
    for (n, c, vol) in [(1, 2, (4, 16, 16)), (1, 2, (9, 32, 48)), (1, 3, (5, 16, 16)), (2, 2, (3, 16, 16))]:
        v = vol[0] * vol[1] * vol[2]
        k = ops.loss_k(c)
        need = n * ops.loss_rows(v) * k
        arena = torch.full((need + 64,), -777.0, dtype=torch.float32, device="cuda")
        part = arena[:need]
        g = torch.Generator().manual_seed(3)
        logits = torch.randn((n, c) + vol, generator=g)
        lab = torch.nn.functional.one_hot(torch.randint(0, c, (n,) + vol, generator=g), c).permute(0, 4, 1, 2, 3).float().contiguous()
        out = torch.empty(4 + c, dtype=torch.float32, device="cuda")
        coef = torch.empty(n * c * 2 + 2, dtype=torch.float32, device="cuda")
        ops.seg_loss_fwd(logits.cuda(), lab.cuda(), None, None, (1.0, 0.0, 0.0, 0.0), True, part, out, coef)
        torch.cuda.synchronize()
        assert bool((arena[need:] == -777.0).all()), (n, c, vol)
        ref = float(R.dice_loss(logits, lab))
        assert abs(float(out[0]) - ref) < 1e-5


@pytest.mark.parametrize("route", ["engine", "autograd"])
@pytest.mark.parametrize("variant", ["dice", "dice_pw", "combined"])
def test_training_all_agent_matches_reference(golden_dir, variant, route):
    """SegmentationAgent.training_all through the drop-in registry/agent surface: routed through fplx.TrainStep (the default
    when network, loss and optimiser are fplx's own) and in autograd mode (net(x), loss module, loss.backward(),
    optimizer.step() - what a foreign loss or optimiser gets)."""
    import fplx
    g = np.load(os.path.join(golden_dir, "train_step.npz"))
    tcfg = {"dis": False, "train_fpl_uda": True, "loss_type": "DiceLoss", "optimizer": "Adam",
            "learning_rate": 1e-3, "momentum": 0.9, "weight_decay": 1e-5, "lr_scheduler": "MultiStepLR",
            "lr_gamma": 0.5, "lr_milestones": [2, 4], "iter_valid": 1, "gpus": [0]}
    if variant == "combined":
        tcfg.update({"loss_type": ["DiceLoss", "CrossEntropyLoss"], "loss_weight": [0.5, 0.5]})
    cfg = {"dataset": {"tensor_type": "float"}, "network": dict(NETS["tiny"]), "training": tcfg, "testing": {}}
    agent = fplx.SegmentationAgent(cfg, "train")
    agent.create_network()
    load_det_weights(agent.net, cfg["network"], "cuda")
    agent.create_optimizer()
    agent.create_loss_calculator()
    agent.engine_mode = route == "engine"
    assert (agent._engine_step() is not None) == (route == "engine")
    b = _batches(variant != "dice")
    agent.set_loaders([b[0]], [b[1]])
    lrs = []
    for step in range(1, 6):
        sc = agent.training_all()
        assert abs(sc["loss"] - float(g["%s.step%d.loss" % (variant, step)])) < 5e-5, (step, sc)
        np.testing.assert_allclose(sc["class_dice"], g["%s.step%d.class_dice" % (variant, step)], atol=3e-3)
        lrs.append(agent.optimizer.param_groups[0]["lr"])
        if step in (1, 3):
            _check_params(agent.net.state_dict(), g, variant, step)
    np.testing.assert_allclose(lrs, g["%s.lrs" % variant], rtol=1e-12)


@pytest.mark.parametrize("dual", [True, False])
def test_agent_engine_route_equals_autograd_route(dual):
    """SegmentationAgent.training_all / training (dual = False: one backward + optimiser + scheduler step per domain,
    agent_seg.py:336-357, with the entropy regulariser of 352-354) routed through fplx.TrainStep give the parameters, losses and
    class Dice of the autograd route - same kernels on the same buffers."""
    import fplx
    res = []
    for route in (True, False):
        tcfg = {"dis": False, "train_fpl_uda": True, "loss_type": ["DiceLoss", "CrossEntropyLoss"], "loss_weight": [0.5, 0.5],
                "optimizer": "Adam", "learning_rate": 1e-3, "momentum": 0.9, "weight_decay": 1e-5, "lr_scheduler": "MultiStepLR",
                "lr_gamma": 0.5, "lr_milestones": [2, 4], "iter_valid": 3, "gpus": [0]}
        cfg = {"dataset": {"tensor_type": "float"}, "network": dict(NETS["tiny"]), "training": tcfg, "testing": {}}
        agent = fplx.SegmentationAgent(cfg, "train")
        agent.create_network()
        load_det_weights(agent.net, cfg["network"], "cuda")
        agent.create_optimizer()
        agent.create_loss_calculator(0.0 if dual else 1.0)
        agent.engine_mode = route
        b = _batches(True)
        agent.set_loaders([b[0]], [b[1]])
        scs = [agent.training_all() if dual else agent.training() for _ in range(2)]
        res.append((agent.net.flat_params.detach().clone(), scs, agent.optimizer.param_groups[0]["lr"]))
    (pa, sa, la), (pb, sb, lb) = res
    assert la == lb
    for a, b_ in zip(sa, sb):
        assert abs(a["loss"] - b_["loss"]) < 1e-6 and np.abs(a["class_dice"] - b_["class_dice"]).max() < 1e-6
    assert float((pa - pb).abs().max()) <= 1e-6 * float(pb.abs().max())


def test_training_all_engine_mode_matches_reference(golden_dir):
    """fplx.TrainStep.step_all (flat buffers, no autograd) == the same five iterations."""
    import fplx
    g = np.load(os.path.join(golden_dir, "train_step.npz"))
    p = dict(NETS["tiny"])
    net = fplx.UNet2D5_dsbn(p)
    load_det_weights(net, p, "cuda")
    ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-3, weight_decay=1e-5, milestones=[2, 4], gamma=0.5)
    b = [{k: v.cuda() for k, v in d.items()} for d in _batches(True)]
    for step in range(1, 6):
        outs = ts.step_all(b)
        loss = 0.5 * (outs[0][0].item() + outs[1][0].item())
        assert abs(loss / 2 - float(g["dice_pw.step%d.loss" % step])) < 5e-5
        if step in (1, 3):
            _check_params(net.state_dict(), g, "dice_pw", step)


def test_second_domain_reuses_the_weight_packs_bit_identically():
    """step_all: the second domain's forward of an iteration uses the first one's weight packs (no parameter changes in between,
    agent_seg.py:462-486) - the parameters after three iterations are the bits of the run that repacks every forward, and the
    next iteration does repack (a stale pack would show in the second and third iteration)."""
    import fplx
    res = []
    for reuse in (True, False):
        p = dict(NETS["tiny"])
        net = fplx.UNet2D5_dsbn(p)
        load_det_weights(net, p, "cuda")
        net.engine.allow_pack_reuse = reuse
        ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-3, weight_decay=1e-5, milestones=[2, 4], gamma=0.5)
        b = [{k: v.cuda() for k, v in d.items()} for d in _batches(True)]
        calls = [0]
        inner = net.engine._pack

        def counting_pack(adt, _inner=inner, _calls=calls):
            _calls[0] += 1
            return _inner(adt)
        net.engine._pack = counting_pack
        for _ in range(3):
            calls[0] = 0
            ts.step_all(b)
            assert calls[0] == (1 if reuse else 2), (reuse, calls[0])     # ONE set of weight packs per iteration when reused
            assert net.engine._train_packs is None        # the optimiser step drops every pack (FusedAdam.step_flat)
        res.append(net.flat_params.clone())
    assert torch.equal(res[0], res[1])


def test_inferer_matches_reference(golden_dir):
    import fplx
    g = np.load(os.path.join(golden_dir, "inferer.npz"))
    p = dict(NETS["tiny"])
    net = fplx.UNet2D5_dsbn(p)
    load_det_weights(net, p, "cuda")
    net.eval()
    x = torch.from_numpy(detdata.normal("inf.x", (1, 1, 40, 72, 72))).cuda()
    dl = torch.ones(1, dtype=torch.long)
    cfgs = {"sw_nooverlap_tta0": dict(sliding_window_enable=True, sliding_window_size=[16, 32, 32],
                                     sliding_window_stride=[16, 32, 32], tta_mode=0, class_num=2),
            "sw_overlap_tta1": dict(sliding_window_enable=True, sliding_window_size=[16, 32, 32],
                                    sliding_window_stride=[8, 24, 16], tta_mode=1, class_num=2),
            "full_tta1": dict(sliding_window_enable=False, tta_mode=1, class_num=2)}
    with torch.no_grad():
        for tag, c in cfgs.items():
            xx = x if tag != "full_tta1" else x[:, :, :32, :64, :64]
            out = fplx.Inferer(c).run(net, xx, dl)
            assert np.abs(out.cpu().numpy() - g[tag]).max() < 1e-3, tag
    with pytest.raises(ValueError):
        fplx.Inferer({"tta_mode": 3}).run(net, x, dl)


def test_fpl_filter_matches_reference_run_and_oracle(golden_dir):
    import fplx
    from oracle import np_ref as N
    g = np.load(os.path.join(golden_dir, "fpl_filter.npz"))
    unc = {}
    for i in range(3):
        stack = g["vol%d.logits" % i]
        ref = N.fpl_filter(stack)
        r = fplx.filter.fpl_filter(_cuda(stack), want_maps=True)
        st = r["stats"].cpu().numpy()
        assert np.array_equal(r["hards"].cpu().numpy(), ref["hards"])              # masks: bit exact
        assert int(st[1]) == ref["boundary"]
        assert abs(st[0] - float(ref["vars"])) <= 1e-5 * float(ref["vars"])
        assert abs(st[2] - float(ref["uncer_one"])) <= 1e-5 * float(ref["uncer_one"])
        np.testing.assert_allclose(r["means"].cpu().numpy(), ref["means"], atol=2e-7)
        np.testing.assert_allclose(r["uncertainty"].cpu().numpy(), ref["uncertainty"], atol=2e-7)
        unc["./dataset/hrT2_train/img/vol%d.nii.gz" % i] = fplx.filter.fpl_uncertainty(_cuda(stack))["uncer_one"]
    srt = fplx.filter.sort_uncertainty(unc)
    assert [s[1] for s in srt] == [str(s) for s in g["sorted_names"]]
    np.testing.assert_allclose([s[0][0] for s in srt], g["sorted_uncertainty"], rtol=1e-5)
    # confident everywhere -> boundary < 50 -> 1 (agent_seg.py:926-927)
    conf = np.zeros((6, 2, 4, 8, 8), np.float32)
    conf[:, 0] = 30.0
    assert fplx.filter.fpl_uncertainty(_cuda(conf))["uncer_one"] == 1
    # ties and near-ties: exact tie -> class 0 (first maximum of the probabilities)
    t = np.zeros((4, 2, 1, 1, 64), np.float32)
    t[:, 1, 0, 0, 1::2] = np.float32(1e-3)
    t[:, 0, 0, 0, 2::4] = np.float32(3.0)
    ref = N.fpl_filter(t)
    assert np.array_equal(fplx.filter.fpl_filter(_cuda(t))["hards"].cpu().numpy(), ref["hards"])
    lg = detdata.normal("hard.lg", (2, 3, 4, 8, 8), 2.0)
    assert np.array_equal(fplx.filter.hard_label(_cuda(lg)).cpu().numpy(), N.hard_label(lg))


@pytest.mark.parametrize("classes,T", [(2, 6), (3, 4), (4, 5)])
def test_mc_filter_vector_form_equals_the_scalar_form(classes, T):
    """mc_filter's four-voxels-per-thread kernel (16-byte loads, packed label stores; taken when V % 4 == 0 and the pointers are
    16-byte aligned) against the one-voxel kernel (forced by an odd volume / a misaligned view of the same numbers): hard labels,
    means and uncertainties bit for bit, the boundary count exactly, the variance sum to double rounding; and both against numpy."""
    from fplx import ops
    from oracle import np_ref as N
    vol = (6, 10, 12)
    v = vol[0] * vol[1] * vol[2]
    lg = detdata.normal("mcf4.%d.%d" % (classes, T), (T, classes) + vol, 2.5).astype(np.float32)
    lg[:, 1, 0, 0, ::3] = lg[:, 0, 0, 0, ::3]                         # ties: first maximum
    a = ops.mc_filter(torch.from_numpy(lg).cuda(), 0.01, True, True)                     # aligned, V % 4 == 0: vector kernel
    # the same numbers at a pointer that is 4 bytes off a 16-byte boundary: scalar kernel
    flat = torch.zeros(lg.size + 1, dtype=torch.float32, device="cuda")
    flat[1:] = torch.from_numpy(lg).cuda().reshape(-1)
    b = ops.mc_filter(flat[1:].view((T, classes) + vol), 0.01, True, True)
    assert flat[1:].data_ptr() % 16 == 4
    for k in ("hards", "means", "uncertainty"):
        assert torch.equal(a[k], b[k]), k
    sa, sb = a["stats"].cpu().numpy(), b["stats"].cpu().numpy()
    assert sa[1] == sb[1] and abs(sa[0] - sb[0]) <= 1e-12 * abs(sb[0])
    if classes == 2:
        ref = N.fpl_filter(lg)
        assert np.array_equal(a["hards"].cpu().numpy(), ref["hards"])
    # an odd volume goes through the scalar kernel as well and agrees voxel by voxel on the common part
    odd = ops.mc_filter(torch.from_numpy(np.ascontiguousarray(lg.reshape(T, classes, v)[:, :, :v - 1])).cuda(), 0.01, True, True)
    assert torch.equal(odd["hards"].reshape(T, -1), a["hards"].reshape(T, -1)[:, :v - 1])
    assert torch.equal(odd["means"].reshape(-1), a["means"].reshape(-1)[:v - 1])


def test_pixel_weight_bit_exact(golden_dir):
    import fplx
    from oracle import np_ref as N
    g = np.load(os.path.join(golden_dir, "pixel_weight.npz"))
    a, b = _cuda(g["mask_a"]), _cuda(g["mask_b"])
    w = fplx.filter.pixel_weight_from_masks(a, b).cpu().numpy()
    assert np.array_equal(w.astype(np.float64), g["weight"])                          # {1.0, 0.5} exact
    for iw in ("0.37", "1.0"):
        got = fplx.filter.pixel_weight_from_masks(a, b, image_weight=float(iw)).cpu().numpy()
        assert np.array_equal(got, g["set_weight_" + iw]), iw
    # ragged / odd sizes and all-equal / all-different masks
    for shape in [(1,), (3, 5, 7), (1, 1, 257)]:
        m = (detdata.uniform("pwm%s" % (shape,), shape) > 0.5).astype(np.uint8)
        for other in (m, 1 - m):
            got = fplx.filter.pixel_weight_from_masks(_cuda(m), _cuda(other)).cpu().numpy()
            assert np.array_equal(got.astype(np.float64), N.pixel_weight_from_masks(m.copy(), other.copy()))
    with pytest.raises(ValueError):
        fplx.filter.pixel_weight_from_masks(a, b[:1])


def test_image_weight_known_answer(golden_dir):
    import fplx
    d = json.load(open(os.path.join(golden_dir, "image_weight_kat.json")))
    got = fplx.filter.image_weights([(u, p) for u, p in d["input"]])
    assert [str(w) for w in got] == [r[3] for r in d["csv_rows"]]


def test_batched_inferer_equals_the_tile_by_tile_loop():
    """fplx.Inferer gathers all tiles x flips (x MC passes) into one batch and merges them with the reference's additions
    in the reference's order; the literal loop of forwards (`_run_generic`, infer_func.py:96-112, 199-219) must agree."""
    import fplx
    p = dict(NETS["tiny"], dropout=[0, 0, 0.3, 0.4, 0.5])
    net = fplx.UNet2D5_dsbn(p)
    load_det_weights(net, p, "cuda")
    x = torch.from_numpy(detdata.normal("inf.x", (1, 1, 40, 72, 72))).cuda()
    dl = torch.ones(1, dtype=torch.long)
    c = dict(sliding_window_enable=True, sliding_window_size=[16, 32, 32], sliding_window_stride=[8, 24, 16], tta_mode=1, class_num=2)
    with torch.no_grad():
        for train_bn in (False, True):                     # eval: tiles batched; train-mode BatchNorm: one tile per forward
            net.train(train_bn)
            for m in net.modules():
                if type(m) == torch.nn.Dropout:
                    m.eval()
            inf = fplx.Inferer(c)
            a = inf.run(net, x, dl)
            inf.model = net
            b = inf._run_generic(x, dl)
            assert float((a - b).abs().max()) <= 1e-5, train_bn
        # Monte-Carlo passes (test-time dropout) in one batch: right shape, passes differ, every pass a valid prediction
        net.eval()
        for m in net.modules():
            if type(m) == torch.nn.Dropout:
                m.train()
        mc = fplx.Inferer(c).run_mc(net, x, dl, 3)
        assert tuple(mc.shape) == (3, 1, 2, 40, 72, 72) and bool(torch.isfinite(mc).all())
        assert not torch.equal(mc[0], mc[1]) and not torch.equal(mc[1], mc[2])
        # a small forward budget cuts the batch differently; the merge does not care
        small = fplx.Inferer(dict(c, infer_batch_voxels=16 * 32 * 32 * 3))
        for m in net.modules():
            if type(m) == torch.nn.Dropout:
                m.eval()
        assert torch.equal(small.run(net, x, dl), fplx.Inferer(c).run(net, x, dl))


@pytest.mark.parametrize("case", ["tiles", "whole", "whole_odd_w"])
def test_sw_merge_mc_reads_the_chunked_pass_major_buffer(case):
    """fplx_sw_merge_mc (all Monte-Carlo passes in one launch, predictions left where the network's chunked calls wrote them:
    [chunks][passes][patches of the chunk]) == fplx_sw_merge per pass on that pass's patches gathered into one contiguous batch;
    and for the whole-volume case (no window: the 16-byte fast path where W % 4 == 0) == the reference's expression
    ((o1 + flip_H(o2)) + flip_W(o3) + flip_HW(o4)) / 4 of infer_func.py:199-219, bit for bit."""
    from fplx import ops
    from fplx._lib import call
    from fplx.infer import Inferer, _FLIPS_TTA
    n, classes, passes, chunk = 2, 3, 3, 5                       # 4 flips x tiles x 2 samples: the last chunk is ragged
    shape = {"tiles": (10, 24, 20), "whole": (6, 10, 24), "whole_odd_w": (6, 10, 22)}[case]
    cfg = dict(tta_mode=1, class_num=classes)
    if case == "tiles":
        cfg.update(sliding_window_enable=True, sliding_window_size=[8, 16, 16], sliding_window_stride=[4, 8, 12])
    inf = Inferer(cfg)
    img = torch.zeros((n, 1) + shape, device="cuda")
    window, starts, flips = inf._plan(img)
    tiles = len(starts[0]) * len(starts[1]) * len(starts[2])
    nb = len(flips) * tiles * n
    assert nb % chunk != 0
    pred = torch.from_numpy(detdata.normal("swmc.%s" % case, (passes, nb, classes) + tuple(window))).cuda()
    buf = torch.cat([pred[:, b0:b0 + chunk].reshape((-1, classes) + tuple(window)) for b0 in range(0, nb, chunk)])
    margs = Inferer._c_args((n, classes) + shape, window, starts, flips)
    got = torch.full((passes, n, classes) + shape, float("nan"), device="cuda")
    call("fplx_sw_merge_mc", ops.ptr(buf), passes, chunk, *margs, ops.ptr(got), ops.stream())
    for q in range(passes):
        one = torch.empty((n, classes) + shape, device="cuda")
        call("fplx_sw_merge", ops.ptr(pred[q].contiguous()), *margs, ops.ptr(one), ops.stream())
        assert torch.equal(got[q], one), q
    if case != "tiles":
        o = pred.view((passes, 4, n, classes) + shape)
        assert _FLIPS_TTA == (0, 2, 1, 3)
        ref = ((o[:, 0] + torch.flip(o[:, 1], [-2])) + torch.flip(o[:, 2], [-1]) + torch.flip(o[:, 3], [-2, -1])) / 4
        assert torch.equal(got, ref)


def test_ckpt_mode_3_ensemble_is_the_mean_of_the_checkpoints_logits(tmp_path):
    """agent_seg.py:966-1019: np.mean over the checkpoints' predictions, then the usual hard labels."""
    import fplx
    p = dict(NETS["tiny"])
    names = []
    x = torch.from_numpy(detdata.normal("ens.x", (1, 1, 16, 32, 32)))
    preds = []
    for k in range(3):
        net = fplx.UNet2D5_dsbn(dict(p))
        sd = {kk: torch.from_numpy(v) for kk, v in detdata.state_dict_3d(p, prefix="ens%d" % k).items()}
        net.load_state_dict(sd)
        net.cuda().eval()
        with torch.no_grad():
            preds.append(net(x.cuda(), domain_label=torch.ones(1, dtype=torch.long)))
        f = str(tmp_path / ("m_%d.pt" % k))
        torch.save({"model_state_dict": net.state_dict()}, f)
        names.append(f)
    cfg = {"dataset": {"tensor_type": "float"}, "network": dict(p), "training": {},
           "testing": {"domian_label": 1, "gpus": [0], "fpl": False, "evaluation_mode": True, "tta_mode": 0, "ckpt_mode": 3,
                       "ckpt_name": names, "sliding_window_enable": False}}
    agent = fplx.SegmentationAgent(cfg, "test")
    agent.create_network()
    agent.set_loaders(test_loader=[{"image": x, "names": ["case0.nii.gz"]}])
    out = agent.infer()
    mean = ((preds[0] + preds[1]) + preds[2]) / 3.0
    assert torch.equal(out["case0.nii.gz"], fplx.filter.hard_label(mean)[0])
    cfg["testing"]["ckpt_mode"] = 2
    agent2 = fplx.SegmentationAgent(cfg, "test")
    agent2.create_network()
    agent2.set_loaders(test_loader=[{"image": x, "names": ["case0.nii.gz"]}])
    with pytest.raises(ValueError):
        agent2.infer()

"""Pin the CPU oracle (oracle/torch_ref.py, oracle/np_ref.py) against fixtures produced by
RUNNING THE REFERENCE (tests/golden/make_golden.py) and against the reference's own
known-answer data pair.  CPU only."""
import json
import os
import numpy as np
import pytest
import torch

import detdata
from make_golden_cfg import NETS, SHAPES, label_for
from oracle import torch_ref as R
from oracle import np_ref as N

torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))


# max-normalised gradient tolerance: the fp32 reference itself moves by this much when only the
# CPU thread count (= summation order) changes (BN over few voxels at the deepest level)
GRAD_TOL = {"tiny": 5e-4, "tiny25": 5e-3, "c4": 5e-4, "cfg1": 5e-3, "tinybl": 5e-4, "tinybl25": 2e-2}   # tinybl25: BatchNorm over 16 voxels at level 4 amplifies fp32 round-off


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


# tiny25: conv_dims = [2, 2, 3, 3, 3]; tinybl / tinybl25: bilinear = True (1x1 convolution + (tri / bi)linear upsampling)
@pytest.mark.parametrize("name", ["tiny", "tiny25", "c4", "cfg1", "tinybl", "tinybl25"])
def test_net_forward_backward_matches_reference(golden_dir, name):
    g = _load(golden_dir, "net_%s.npz" % name)
    params = NETS[name]
    x = torch.from_numpy(detdata.normal("x." + name, SHAPES[name]))
    y = torch.from_numpy(label_for(name))
    for d in (0, 1):
        sd, _ = R.split_state(detdata.state_dict_3d(params), requires_grad=False)
        with torch.no_grad():
            le = R.unet_forward(sd, params, x, d, train=False)
        np.testing.assert_allclose(le.numpy(), g["logits_eval_d%d" % d], atol=2e-4, rtol=1e-5)
        sd, p = R.split_state(detdata.state_dict_3d(params))
        lt = R.unet_forward(sd, params, x, d, train=True)
        np.testing.assert_allclose(lt.detach().numpy(), g["logits_train_d%d" % d], atol=2e-4, rtol=1e-5)
        loss = R.dice_loss(lt, y)
        assert abs(loss.item() - float(g["loss_dice_d%d" % d])) < 1e-6
        loss.backward()
        for k in g.files:
            pre = "d%d." % d
            if k.startswith(pre) and ".bns." in k:
                np.testing.assert_allclose(sd[k[len(pre):]].numpy(), g[k], atol=1e-6, rtol=1e-5, err_msg=k)
            if k.startswith("grad_d%d." % d):
                kk = k[len("grad_d%d." % d):]
                ref = g[k]
                tol = max(GRAD_TOL[name] * np.abs(ref).max(), 2e-7)  # conv bias before BN: true grad is 0, fp noise
                np.testing.assert_allclose(p[kk].grad.numpy(), ref, atol=tol, rtol=0, err_msg=k)
            if k.startswith("gradsub") and ("_d%d." % d) in k:
                head, kk = k.split("_d%d." % d)
                stride = int(head[len("gradsub"):])
                ref = g[k]
                tol = max(GRAD_TOL[name] * np.abs(ref).max(), 2e-7)  # conv bias before BN: true grad is 0, fp noise
                np.testing.assert_allclose(p[kk].grad.numpy().reshape(-1)[::stride], ref, atol=tol, rtol=1e-3,
                                           err_msg=k)
        keys = [str(s) for s in g["gradnorm_keys_d%d" % d]]
        vals = g["gradnorm_vals_d%d" % d]
        # exactly the parameters the reference gives a gradient to (3D branch, active domain's BN only)
        ours = sorted(k for k, t in p.items() if t.grad is not None)
        assert ours == keys
        for k, v in zip(keys, vals):
            assert abs(float(p[k].grad.norm()) - v) <= max(GRAD_TOL[name] * v, 1e-6), k


def test_losses_match_reference(golden_dir):
    g = _load(golden_dir, "losses.npz")
    lab = torch.from_numpy(g["label"])
    pw = torch.from_numpy(g["pixel_weight"])
    iw = torch.from_numpy(g["image_weight"])
    pw_bin = (pw > 0).float()

    def check(tag, fn):
        lg = torch.from_numpy(g["logits"]).clone().requires_grad_(True)
        v = fn(lg)
        v.backward()
        assert abs(v.item() - float(g[tag + ".loss"])) < 2e-6, tag
        np.testing.assert_allclose(lg.grad.numpy(), g[tag + ".dlogits"], atol=1e-9, rtol=2e-4, err_msg=tag)

    check("dice", lambda l: R.dice_loss(l, lab))
    check("dice_pw", lambda l: R.dice_loss(l, lab, pw))
    check("ce", lambda l: R.ce_loss(l, lab))
    check("ce_pw", lambda l: R.ce_loss(l, lab, pw))
    check("dice_weight", lambda l: R.dice_loss_image_weighted(l, lab, pw_bin, iw))
    comb = R.loss_from_config({"loss_type": ["DiceLoss", "CrossEntropyLoss"], "loss_weight": [0.6, 0.4]})
    check("combined", lambda l: comb({"prediction": l, "ground_truth": lab}))
    check("combined_pw", lambda l: comb({"prediction": l, "ground_truth": lab, "pixel_weight": pw}))
    check("entropy", R.entropy_term)
    lab3 = torch.from_numpy(g["label3"])
    for tag, fn in (("dice3", R.dice_loss), ("ce3", R.ce_loss)):
        lg = torch.from_numpy(g["logits3"]).clone().requires_grad_(True)
        v = fn(lg, lab3)
        v.backward()
        assert abs(v.item() - float(g[tag + ".loss"])) < 2e-6
        np.testing.assert_allclose(lg.grad.numpy(), g[tag + ".dlogits"], atol=1e-9, rtol=2e-4)
    with pytest.raises(ValueError):
        R.loss_from_config({"loss_type": "NoSuchLoss"})


@pytest.mark.parametrize("variant", ["dice", "dice_pw", "combined"])
def test_training_all_matches_reference(golden_dir, variant):
    g = _load(golden_dir, "train_step.npz")
    name = "tiny"
    params = NETS[name]
    tcfg = {"loss_type": "DiceLoss"}
    if variant == "combined":
        tcfg = {"loss_type": ["DiceLoss", "CrossEntropyLoss"], "loss_weight": [0.5, 0.5]}
    with_w = variant != "dice"
    n, _, D, H, W = SHAPES[name]
    batches = []
    for dom in (0, 1):
        b = {"image": torch.from_numpy(detdata.normal("ts.x.d%d" % dom, SHAPES[name])),
             "label_prob": torch.from_numpy(detdata.ball_label((D, H, W), 5.0, n=n,
                                                               offsets=[(dom, 1, -2), (1, -3, 2 + dom)]))}
        if with_w and dom == 1:
            pw = (detdata.uniform("ts.pw", (n, 1, D, H, W)) > 0.25).astype(np.float32)
            iw = np.array([0.8, 0.45], np.float32)
            b["pixel_weight"] = torch.from_numpy(pw * iw[:, None, None, None, None])
            b["image_weight"] = torch.from_numpy(iw)
        batches.append(b)
    sd, p = R.split_state(detdata.state_dict_3d(params))
    opt = R.AdamRef(p, lr=1e-3, weight_decay=1e-5, milestones=[2, 4], gamma=0.5)
    loss_fn = R.loss_from_config(tcfg)
    lrs = []
    for step in range(1, 6):
        loss, dices = R.training_all_step(sd, p, opt, params, batches, loss_fn)
        # training_all returns train_loss / iter_valid / num_domains (agent_seg.py:496)
        assert abs(loss / 2 - float(g["%s.step%d.loss" % (variant, step)])) < 2e-5, (step, loss)
        cd = (dices[0] + dices[1]) / 2
        np.testing.assert_allclose(cd, g["%s.step%d.class_dice" % (variant, step)], atol=2e-3)
        lrs.append(opt.lr)
        if step in (1, 3):
            for k in g.files:
                pre = "%s.step%d." % (variant, step)
                if k.startswith(pre) and k[len(pre):] in sd:
                    ref = g[k]
                    # Adam turns a near-zero gradient into a +-lr step whose sign is fp noise: bound the
                    # bulk tightly and every element by lr * steps
                    diff = np.abs(sd[k[len(pre):]].detach().numpy() - ref)
                    assert diff.max() <= 1e-3 * step + 1e-6, k
                    assert (diff <= 3e-5 * step + 1e-4 * np.abs(ref)).mean() >= 0.995, k
    np.testing.assert_allclose(lrs, g["%s.lrs" % variant], rtol=1e-12)


def test_inferer_matches_reference(golden_dir):
    g = _load(golden_dir, "inferer.npz")
    params = NETS["tiny"]
    sd, _ = R.split_state(detdata.state_dict_3d(params), requires_grad=False)
    x = torch.from_numpy(detdata.normal("inf.x", (1, 1, 40, 72, 72)))

    def model(img):
        with torch.no_grad():
            return R.unet_forward(sd, params, img, 1, train=False)

    cfgs = {"sw_nooverlap_tta0": dict(sliding_window_enable=True, sliding_window_size=[16, 32, 32],
                                     sliding_window_stride=[16, 32, 32], tta_mode=0, class_num=2),
            "sw_overlap_tta1": dict(sliding_window_enable=True, sliding_window_size=[16, 32, 32],
                                    sliding_window_stride=[8, 24, 16], tta_mode=1, class_num=2),
            "full_tta1": dict(sliding_window_enable=False, tta_mode=1, class_num=2)}
    for tag, c in cfgs.items():
        xx = x if tag != "full_tta1" else x[:, :, :32, :64, :64]
        out = R.inferer_run(model, xx, c)
        np.testing.assert_allclose(out.numpy(), g[tag], atol=3e-5, rtol=1e-5, err_msg=tag)
    with pytest.raises(ValueError):
        R.inferer_run(model, x, {"tta_mode": 7})


def test_fpl_filter_matches_reference_run(golden_dir):
    g = _load(golden_dir, "fpl_filter.npz")
    unc = {}
    for i in range(3):
        r = N.fpl_filter(g["vol%d.logits" % i])
        unc["./dataset/hrT2_train/img/vol%d.nii.gz" % i] = r["uncer_one"]
        assert r["hards"].dtype == np.uint8 and r["maps"].dtype == np.float32
        assert r["boundary"] >= 50
    srt = N.sort_uncertainty(unc)
    assert [s[1] for s in srt] == [str(s) for s in g["sorted_names"]]
    # bit-exact: same float32 numpy arithmetic as the reference run
    assert [float(s[0][0]) for s in srt] == [float(v) for v in g["sorted_uncertainty"]]
    # confident-everywhere stack -> boundary < 50 -> uncer_one == 1 (agent_seg.py:926-927)
    conf = np.zeros((6, 2, 4, 8, 8), np.float32)
    conf[:, 0] = 30.0
    assert N.fpl_filter(conf)["uncer_one"] == 1


def test_pixel_weight_and_set_weight_match_reference(golden_dir):
    g = _load(golden_dir, "pixel_weight.npz")
    w = N.pixel_weight_from_masks(g["mask_a"].copy(), g["mask_b"].copy())
    assert w.dtype == g["weight"].dtype
    assert np.array_equal(w, g["weight"])
    for iw in ("0.37", "1.0"):
        got = N.set_weight(np.float32(float(iw)), np.asarray(g["weight"], np.float32).copy())
        assert np.array_equal(got, g["set_weight_" + iw]) and got.dtype == g["set_weight_" + iw].dtype
        got = N.set_weight(float(iw), np.asarray(g["weight"], np.float32).copy())
        assert np.array_equal(got, g["set_weight_py_" + iw])


def test_image_weight_known_answer(golden_dir):
    """The reference's own data pair: dataset/weight/cyc121_vst1s-gan.npy ->
    config_dual/data_vs/train_vs_t1s_wi+wp.csv (image_weight column)."""
    d = json.load(open(os.path.join(golden_dir, "image_weight_kat.json")))
    rows = [(u, p) for u, p in d["input"]]
    got = N.image_weights(rows)
    assert d["csv_header"] == ["image", "label", "pixel_weight", "image_weight"]
    assert len(got) == len(d["csv_rows"]) == 100
    for (u, path), w, row in zip(rows, got, d["csv_rows"]):
        assert row[0] == path
        assert str(w) == row[3]          # csv.writer writes repr(float): bit-exact float64


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    r = N.philox4x32_10([0], [0], [0], [0], 0, 0)
    assert [int(v[0]) for v in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = N.philox4x32_10([0xffffffff], [0xffffffff], [0xffffffff], [0xffffffff], 0xffffffff, 0xffffffff)
    assert [int(v[0]) for v in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = N.philox4x32_10([0x243f6a88], [0x85a308d3], [0x13198a2e], [0x03707344], 0xa4093822, 0x299f31d0)
    assert [int(v[0]) for v in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    keep = N.philox_keep_mask(1234, 5, 100000, 0.3)
    assert abs(keep.mean() - 0.7) < 0.01


def test_transform_chain_against_reference_fixture(golden_dir):
    """oracle/np_ref.py transforms vs the outputs of the reference's NormalizeWithMeanStd -> Pad -> RandomCrop ->
    RandomFlip -> LabelToProbability chain (tests/golden/make_golden_transforms.py): same `random` seeds, so the crop
    boxes and flip axes must be reproduced too; data movement is bit-exact."""
    import json
    import random
    from oracle import np_ref as R
    g = _load(golden_dir, "transforms.npz")
    p = json.loads(str(g["params_json"]))
    s0 = {"image": g["image"].copy()}
    np.testing.assert_array_equal(R.tf_normalize(s0, [0])["image"], g["after_NormalizeWithMeanStd_image"])
    s1 = R.tf_pad({"image": g["after_NormalizeWithMeanStd_image"].copy(), "label": g["label"].copy(),
                   "pixel_weight": g["pixel_weight"].copy()}, p["pad_output_size"])
    np.testing.assert_array_equal(s1["image"], g["after_Pad_image"])
    np.testing.assert_array_equal(s1["label"], g["after_Pad_label"])
    np.testing.assert_array_equal(s1["pixel_weight"], g["after_Pad_pixel_weight"])
    seeds = sorted(int(k[4:-6]) for k in g.files if k.endswith("_image") and k.startswith("seed"))
    assert len(seeds) >= 5
    for seed in seeds:
        random.seed(seed)
        s = R.tf_train_chain({"image": g["image"].copy(), "label": g["label"].copy(),
                              "pixel_weight": g["pixel_weight"].copy()}, p)
        k = "seed%d_" % seed
        assert list(map(list, json.loads(str(g[k + "crop_param"])))) == [list(x) for x in s["RandomCrop_Param"]]
        assert json.loads(str(g[k + "flip_param"])) == s["RandomFlip_Param"]
        for name in ("image", "label", "label_prob", "pixel_weight"):
            np.testing.assert_array_equal(s[name], g[k + name], err_msg="%s seed %d" % (name, seed))


EV_CASES = (("l12", [1, 2], False), ("l1", [1], False), ("fuse12", [1, 2], True))


def test_evaluation_scores_against_reference_fixture(golden_dir):
    """oracle ev_* vs get_multi_class_evaluation_score of the reference (tests/golden/make_golden_eval.py).
    dice / rve / volume are bit-identical; iou to 1e-6 (float32 `+ 1e-5` under NumPy 2 in the reference, see np_ref)."""
    from oracle import np_ref as R
    g = _load(golden_dir, "evaluation.npz")
    sp = tuple(g["spacing"])
    for metric in ("dice", "iou", "rve", "volume"):
        for tag, labels, fuse in EV_CASES:
            want = g["%s_%s" % (metric, tag)]
            got = np.array([R.ev_multi_class(g["s%d" % i][None], g["g%d" % i][None], labels, fuse, sp, metric)
                            for i in range(6)], np.float64)
            if metric == "iou":
                np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9)
            else:
                np.testing.assert_array_equal(got, want)


def test_surface_metric_restatement_known_answers():
    """ev_edge_points / ev_raster_scan / ev_binary_assd (GeodisTK restated - PARITY UNPINNED, the extension is absent):
    hand-checkable answers, the one-sweep convergence the GPU kernel's closed form rests on, and the sandwich
    Euclidean <= lattice distance (<= 1.15 x Euclidean at unit spacing) against scipy's exact distance transform."""
    from scipy import ndimage
    from oracle import np_ref as R
    cube = np.zeros((5, 5, 5), bool)
    cube[1:4, 1:4, 1:4] = True
    e = R.ev_edge_points(cube)
    assert e.sum() == 26 and e[2, 2, 2] == 0
    full = np.ones((3, 4, 4), bool)                            # outside the volume is background: every border voxel
    assert R.ev_edge_points(full).sum() == 3 * 16 - 1 * 2 * 2
    seed = np.zeros((6, 7, 8), np.uint8)
    seed[1, 2, 3] = 1
    d = R.ev_raster_scan(seed, (1.0, 1.0, 1.0), 2)
    np.testing.assert_allclose(d[4, 4, 4], np.sqrt(3) + np.sqrt(2) + 1.0, rtol=1e-6)      # offsets (3, 2, 1)
    np.testing.assert_allclose(d[1, 2, 7], 4.0, rtol=1e-6)
    sp = (1.5, 0.41, 0.41)
    d = R.ev_raster_scan(seed, sp, 2)
    np.testing.assert_allclose(d[3, 2, 3], 3.0, rtol=1e-6)
    np.testing.assert_allclose(d[2, 3, 4], np.sqrt(1.5 ** 2 + 2 * 0.41 ** 2), rtol=1e-6)
    rs = np.random.RandomState(3)
    seeds = (rs.rand(7, 9, 10) < 0.02).astype(np.uint8)
    seeds[3, 4, 5] = 1
    one, two = R.ev_raster_scan(seeds, sp, 1), R.ev_raster_scan(seeds, sp, 2)
    np.testing.assert_allclose(one, two, rtol=2e-6)                                       # converged after one sweep pair
    edt = ndimage.distance_transform_edt(seeds == 0, sampling=sp)
    assert (two >= edt * (1 - 1e-6)).all()                     # a lattice path is never shorter than the straight line
    iso = R.ev_raster_scan(seeds, (1.0, 1.0, 1.0), 2)          # (anisotropic spacing: up to ~25 % longer, by design)
    edt = ndimage.distance_transform_edt(seeds == 0)
    assert (iso >= edt * (1 - 1e-6)).all() and (iso <= edt * 1.15 + 1e-6).all()
    # assd: a cube against itself shifted by one voxel along x; 2D squeeze; the cap and the 0 / 0 of empty volumes
    a = np.zeros((7, 9, 9), bool)
    a[2:5, 2:6, 2:6] = True
    b = np.roll(a, 1, axis=2)
    v = R.ev_binary_assd(a, b, (1.0, 1.0, 1.0))
    assert 0.0 < v < 1.0
    assert R.ev_binary_assd(a, a, None) == 0 and R.ev_binary_hd95(a, b, None) == 1.0
    assert R.ev_multi_class(a[None, 3:4].astype(np.uint8), b[None, 3:4].astype(np.uint8), [1], False, (1, 1, 1), "hd95") == [1.0]
    assert R.ev_binary_assd(a, np.zeros_like(a), None) == 50

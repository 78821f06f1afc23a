"""-m gpu parity of fplx.transform (csrc/sample.hip) against the reference-generated fixture tests/golden/transforms.npz
and against the numpy oracle on other shapes.  Data movement (pad / crop / flip / one-hot) and the random decisions are
bit-exact; the normalisation is float32 arithmetic on a float32 mean / std that the GPU reduces in float64 while numpy
uses pairwise float32 sums, so it is checked to rtol 2e-6 / atol 2e-6 (values are O(1))."""
import json
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NORM_RTOL, NORM_ATOL = 2e-6, 2e-6


def _fx(golden_dir):
    return np.load(os.path.join(golden_dir, "transforms.npz"), allow_pickle=False)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _params(g):
    p = json.loads(str(g["params_json"]))
    p["task"] = "segmentation"
    return p


def test_transform_chain_matches_reference_fixture(golden_dir):
    import fplx
    from fplx import transform as T
    g = _fx(golden_dir)
    p = _params(g)
    names = ["NormalizeWithMeanStd", "Pad", "RandomCrop", "RandomFlip", "LabelToProbability"]
    # stage by stage on the first seed
    s = {"image": _dev(g["image"]), "label": _dev(g["label"]), "pixel_weight": _dev(g["pixel_weight"])}
    s = T.NormalizeWithMeanStd(dict(p))(s)
    np.testing.assert_allclose(s["image"].cpu().numpy(), g["after_NormalizeWithMeanStd_image"], rtol=NORM_RTOL,
                               atol=NORM_ATOL)
    s["image"] = _dev(g["after_NormalizeWithMeanStd_image"])           # exact input for the bit-exact stages
    s = T.Pad(dict(p))(s)
    for k in ("image", "label", "pixel_weight"):
        assert np.array_equal(s[k].cpu().numpy(), g["after_Pad_" + k]), k
    seeds = sorted(int(k[4:-6]) for k in g.files if k.startswith("seed") and k.endswith("_image"))
    assert len(seeds) >= 5
    for seed in seeds:
        random.seed(seed)
        s = {"image": _dev(g["image"]), "label": _dev(g["label"]), "pixel_weight": _dev(g["pixel_weight"])}
        s = T.apply_transforms(T.build_transforms(names, dict(p)), s)
        k = "seed%d_" % seed
        assert json.loads(s["RandomCrop_Param"]) == json.loads(str(g[k + "crop_param"])), seed
        assert json.loads(s["RandomFlip_Param"]) == json.loads(str(g[k + "flip_param"])), seed
        assert json.loads(s["Pad_Param"]) == json.loads(str(g[k + "pad_param"])), seed
        np.testing.assert_allclose(s["image"].cpu().numpy(), g[k + "image"], rtol=NORM_RTOL, atol=NORM_ATOL)
        for name in ("label", "label_prob", "pixel_weight"):
            assert np.array_equal(s[name].cpu().numpy(), g[k + name]), (name, seed)
        assert s["label_prob"].dtype == torch.float32 and s["label"].dtype == torch.uint8


@pytest.mark.parametrize("shape,out,focus", [
    ((2, 9, 33, 21), [16, 48, 32], True),        # two channels, every axis padded (odd margins)
    ((1, 20, 30, 40), [8, 16, 24], False),       # no padding, crop only
    ((1, 5, 7, 6), [16, 24, 20], True),          # pad wider than the volume: multiple reflections
    ((1, 12, 20, 20), [None, 16, 16], True),     # depth not cropped
])
def test_transforms_match_oracle(shape, out, focus):
    from fplx import transform as T
    from oracle import np_ref as R
    rs = np.random.RandomState(hash(shape) % 1000)
    img = (rs.randn(*shape) * 20 + 100).astype(np.float32)
    lab = (rs.rand(1, *shape[1:]) > 0.93).astype(np.uint8) * rs.randint(1, 3, (1,) + shape[1:]).astype(np.uint8)
    pw = rs.rand(1, *shape[1:]).astype(np.float32)
    pad_out = [o if o is not None else shape[1] for o in out]
    p = {"task": "segmentation", "normalizewithmeanstd_channels": None, "pad_output_size": pad_out,
         "randomcrop_output_size": out, "randomcrop_foreground_focus": focus, "randomcrop_foreground_ratio": 0.7,
         "randomcrop_mask_label": [2], "randomflip_flip_depth": True, "randomflip_flip_height": True,
         "randomflip_flip_width": True, "labeltoprobability_class_num": 3}
    names = ["NormalizeWithMeanStd", "Pad", "RandomCrop", "RandomFlip", "LabelToProbability"]
    for seed in range(6):
        random.seed(100 + seed)
        want = R.tf_train_chain({"image": img.copy(), "label": lab.copy(), "pixel_weight": pw.copy()}, p)
        random.seed(100 + seed)
        got = T.apply_transforms(T.build_transforms(names, dict(p)),
                                 {"image": _dev(img), "label": _dev(lab), "pixel_weight": _dev(pw)})
        assert json.loads(got["RandomCrop_Param"]) == [list(x) for x in want["RandomCrop_Param"]]
        assert json.loads(got["RandomFlip_Param"]) == want["RandomFlip_Param"]
        np.testing.assert_allclose(got["image"].cpu().numpy(), want["image"], rtol=NORM_RTOL, atol=NORM_ATOL)
        for name in ("label", "label_prob", "pixel_weight"):
            assert np.array_equal(got[name].cpu().numpy(), want[name]), (name, seed)


def test_normalize_given_mean_std_and_moments():
    from fplx import ops
    rs = np.random.RandomState(3)
    x = (rs.randn(37, 41, 29) * 55 + 300).astype(np.float32)
    y = ops.normalize_mean_std(_dev(x), (250.0, 40.0))
    assert np.array_equal(y.cpu().numpy(), (x - np.float32(250.0)) / np.float32(40.0))   # same fp32 ops: exact
    y, ms = ops.normalize_mean_std(_dev(x), None, want_moments=True)
    m, s = ms.tolist()
    assert abs(m - x.mean(dtype=np.float64)) < 1e-4 and abs(s - x.std(dtype=np.float64)) < 1e-4


def test_label_bbox_and_empty_mask():
    from fplx import ops
    lab = np.zeros((1, 10, 12, 14), np.uint8)
    lab[0, 2:5, 3:9, 6:7] = 1
    lab[0, 7, 11, 13] = 2
    cnt, lo, hi = ops.label_bbox(_dev(lab), [1])
    assert (cnt, lo, hi) == (3 * 6 * 1, [0, 2, 3, 6], [1, 5, 9, 7])
    cnt, lo, hi = ops.label_bbox(_dev(lab), [1, 2])
    assert (cnt, lo, hi) == (19, [0, 2, 3, 6], [1, 8, 12, 14])
    cnt, _, _ = ops.label_bbox(_dev(lab), [5])
    assert cnt == 0


def test_pad_and_flip_inverse_for_prediction():
    """Pad_inverse = True (config_dual/data_vs/vs_t1s_g.cfg): the test-time prediction is cropped back (pad.py:165-191)"""
    from fplx import transform as T
    p = {"task": "segmentation", "pad_output_size": [16, 48, 48], "randomflip_flip_depth": True,
         "randomflip_flip_height": True, "randomflip_flip_width": True}
    rs = np.random.RandomState(0)
    img = rs.randn(1, 11, 40, 45).astype(np.float32)
    s = T.Pad(dict(p))({"image": _dev(img)})
    assert tuple(s["image"].shape) == (1, 16, 48, 48)
    pred = rs.randn(1, 2, 16, 48, 48).astype(np.float32)
    s["predict"] = _dev(pred)
    s = T.Pad(dict(p)).inverse_transform_for_prediction(s)
    lo, up = json.loads(s["Pad_Param"])
    want = pred[:, :, lo[0]:16 - up[0], lo[1]:48 - up[1], lo[2]:48 - up[2]]
    assert np.array_equal(s["predict"].cpu().numpy(), want)
    random.seed(4)
    f = T.RandomFlip(dict(p))
    s2 = f({"image": _dev(img)})
    axes = json.loads(s2["RandomFlip_Param"])
    assert np.array_equal(s2["image"].cpu().numpy(), np.flip(img, axes) if axes else img)
    s2["predict"] = _dev(pred)
    s2 = f.inverse_transform_for_prediction(s2)
    assert np.array_equal(s2["predict"].cpu().numpy(), np.flip(pred, axes) if axes else pred)


def test_transform_errors():
    from fplx import transform as T
    with pytest.raises(ValueError):
        T.build_transforms(["NoSuchTransform"], {"task": "segmentation"})
    with pytest.raises(ValueError):
        T.Pad({"task": "segmentation", "pad_output_size": [4, 4, 4]})({"image": torch.zeros(1, 2, 2, 2)})   # host tensor
    from fplx import ops
    with pytest.raises(ValueError):
        ops.crop_flip(torch.zeros(1, 4, 4, 4, device="cuda:0"), (2, 0, 0), (4, 4, 4))   # box outside the volume


def _write_cases(tmp_path, n=3):
    """n synthetic cases as .nii.gz (image float64, label int16, pixel weight float32) + the csv NiftyDataset reads"""
    from fplx import nifti
    rs = np.random.RandomState(11)
    rows, arrays = [], []
    for i in range(n):
        shp = (10 + i, 36, 44)
        img = rs.randn(*shp) * 30 + 200
        lab = np.zeros(shp, np.int16)
        lab[3:7, 10:20, 12:30] = 1
        pw = rs.choice([0.5, 1.0], size=shp).astype(np.float32)
        nifti.write_nifti(str(tmp_path / ("img%d.nii.gz" % i)), img, (0.4, 0.4, 1.5), (1.0, 2.0, 3.0))
        nifti.write_nifti(str(tmp_path / ("lab%d.nii.gz" % i)), lab, (0.4, 0.4, 1.5), (1.0, 2.0, 3.0))
        if i != 1:                                         # case 1 has no weight file: the 0.5 fall-back
            nifti.write_nifti(str(tmp_path / ("pw%d.nii.gz" % i)), pw)
        rows.append("img%d.nii.gz,lab%d.nii.gz,pw%d.nii.gz,%s" % (i, i, i, [0.8, 0.6, 1.0][i % 3]))
        arrays.append((img, lab, pw))
    csv = tmp_path / "train.csv"
    csv.write_text("image,label,pixel_weight,image_weight\n" + "\n".join(rows) + "\n")
    return str(csv), arrays


def test_nifty_dataset_samples_and_set_weight(tmp_path):
    import fplx
    csv, arrays = _write_cases(tmp_path)
    ds = fplx.NiftyDataset(str(tmp_path), csv, modal_num=1, with_label=True, transform=None)
    assert len(ds) == 3
    for i, (img, lab, pw) in enumerate(arrays):
        s = ds[i]
        iw = [0.8, 0.6, 1.0][i]
        assert s["names"] == "img%d.nii.gz" % i and s["image_weight"] == iw
        np.testing.assert_allclose(s["spacing"], (1.5, 0.4, 0.4), rtol=1e-6)
        assert np.array_equal(s["image"].cpu().numpy(), img.astype(np.float32)[None])
        assert s["label"].dtype == torch.uint8 and np.array_equal(s["label"].cpu().numpy(), lab[None])
        if i == 1:
            want = np.full(img.shape, 0.5, np.float32)[None]
        else:                                              # nifty_dataset.py:165-168
            want = pw.copy()
            want[want < 1] = 0
            want = (want * iw)[None]
        assert np.array_equal(s["pixel_weight"].cpu().numpy(), want.astype(np.float32)), i
    # csv without a pixel_weight column: all-ones weights scaled by the image weight (nifty_dataset.py:206-209)
    csv2 = tmp_path / "t2.csv"
    csv2.write_text("image,label,image_weight\nimg0.nii.gz,lab0.nii.gz,0.7\n")
    s = fplx.NiftyDataset(str(tmp_path), str(csv2), with_label=True)[0]
    assert np.array_equal(s["pixel_weight"].cpu().numpy(), np.full((1,) + arrays[0][0].shape, np.float32(0.7)))


def test_files_to_train_step_end_to_end(tmp_path):
    """.nii.gz files -> NiftyDataset -> GPU transforms -> collate -> TrainStep.step_all: the loss is finite, goes down and
    the run is reproducible under the same `random` seed (same crops, same flips, fixed-order reductions)."""
    import fplx
    from fplx import transform as T
    from fplx.dataset import collate
    csv, _ = _write_cases(tmp_path, 3)
    p = {"task": "segmentation", "normalizewithmeanstd_channels": [0], "pad_output_size": [16, 32, 32],
         "randomcrop_output_size": [16, 32, 32], "randomcrop_foreground_focus": True, "randomcrop_foreground_ratio": 0.5,
         "randomcrop_mask_label": [1], "randomflip_flip_depth": False, "randomflip_flip_height": True,
         "randomflip_flip_width": True, "labeltoprobability_class_num": 2}
    names = ["NormalizeWithMeanStd", "Pad", "RandomCrop", "RandomFlip", "LabelToProbability"]
    cfg = dict(in_chns=1, feature_chns=[8, 16, 32, 32, 32], dropout=[0.0, 0.0, 0.0, 0.0, 0.0], conv_dims=[3, 3, 3, 3, 3],
               class_num=2, bilinear=False, num_domains=2, precision="fp32")

    def run():
        random.seed(7)
        torch.manual_seed(7)
        ds = fplx.NiftyDataset(str(tmp_path), csv, with_label=True, transform=T.Compose(T.build_transforms(names, dict(p))))
        net = fplx.UNet2D5_dsbn(dict(cfg)).cuda()
        ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-2, weight_decay=1e-5)
        losses = []
        for it in range(8):
            batches = [collate([ds[(it + k) % 3], ds[(it + k + 1) % 3]]) for k in range(2)]      # one batch per domain
            assert tuple(batches[0]["image"].shape) == (2, 1, 16, 32, 32)
            assert tuple(batches[0]["label_prob"].shape) == (2, 2, 16, 32, 32)
            outs = ts.step_all(batches)
            losses.append(float(sum(o[0] for o in outs).item()) / 2)
        return losses, net.flat_params.clone()

    la, pa = run()
    lb, pb = run()
    assert all(np.isfinite(la)) and la == lb and torch.equal(pa, pb)
    assert min(la[4:]) < la[0]


def test_pseudo_label_stage_files_in_files_out(tmp_path):
    """The FPL+ stage-2 hand-off end to end on files: test csv -> NiftyDataset -> [Normalize, Pad] -> sliding-window
    inference -> inverse Pad -> uint8 .nii.gz masks with the input geometry (agent_seg.py:944-953, 1022-1083); then the
    uncertainty list -> weight csv (data/get image_weight.py) and two mask folders -> pixel-weight volumes
    (merge_pixelw.py), and that csv feeds NiftyDataset again (stage 3)."""
    import fplx
    from fplx import nifti, filter as flt
    rs = np.random.RandomState(2)
    root = tmp_path / "data"
    (root / "img").mkdir(parents=True)
    shapes = [(11, 30, 37), (9, 33, 40)]
    for i, shp in enumerate(shapes):
        nifti.write_nifti(str(root / "img" / ("c%d.nii.gz" % i)), rs.randn(*shp) * 40 + 150, (0.5, 0.6, 1.2), (3.0, -4.0, 5.0))
    (tmp_path / "test.csv").write_text("image\nimg/c0.nii.gz\nimg/c1.nii.gz\n")
    net_cfg = dict(net_type="UNet2D5_dsbn", in_chns=1, feature_chns=[8, 16, 32, 32, 32], dropout=[0.0, 0.0, 0.2, 0.2, 0.2],
                   conv_dims=[3, 3, 3, 3, 3], class_num=2, bilinear=False, num_domains=2, precision="fp32")
    config = {
        "dataset": {"root_dir": str(root), "test_csv": str(tmp_path / "test.csv"), "tensor_type": "float",
                    "test_transform": ["NormalizeWithMeanStd", "Pad"], "normalizewithmeanstd_channels": [0],
                    "pad_output_size": [16, 32, 48], "pad_inverse": True},
        "network": net_cfg,
        "training": {"ckpt_save_dir": "model/vs_t1s_g", "random_seed": 1},
        "testing": {"gpus": [0], "domian_label": 1, "output_dir": str(tmp_path / "results"), "evaluation_mode": True,
                    "sliding_window_enable": True, "sliding_window_size": [16, 32, 32], "sliding_window_stride": [16, 16, 16],
                    "tta_mode": 1},
    }
    torch.manual_seed(0)
    agent = fplx.SegmentationAgent(config, "test")
    agent.create_dataset()
    agent.create_network()
    out = agent.infer()
    out_dir = tmp_path / "results" / "vs_t1s_g_test"
    assert sorted(os.listdir(str(out_dir))) == ["c0.nii.gz", "c1.nii.gz"]
    for i, shp in enumerate(shapes):
        d = nifti.load_nifty_volume_as_4d_array(str(out_dir / ("c%d.nii.gz" % i)))
        ref = nifti.load_nifty_volume_as_4d_array(str(root / "img" / ("c%d.nii.gz" % i)))
        assert d["data_array"].dtype == np.uint8 and d["data_array"].shape == (1,) + shp      # inverse Pad applied
        assert d["spacing"] == ref["spacing"] and d["origin"] == ref["origin"] and d["direction"] == ref["direction"]
        assert np.array_equal(d["data_array"][0], out["img/c%d.nii.gz" % i].cpu().numpy())
        # the same mask by hand: pad on the host like the reference, run, crop, argmax
        x = ref["data_array"].astype(np.float32)
        x = (x - x.mean()) / x.std()
        margin = [max(0, [16, 32, 48][k] - shp[k]) for k in range(3)]
        lo = [int(m / 2) for m in margin]
        xp = np.pad(x, [(0, 0)] + [(lo[k], margin[k] - lo[k]) for k in range(3)], "reflect")
        with torch.no_grad():
            pred = agent.inferer.run(agent.net, torch.from_numpy(xp[None]).cuda(), torch.ones(1, dtype=torch.long))
        pred = pred[0, :, lo[0]:lo[0] + shp[0], lo[1]:lo[1] + shp[1], lo[2]:lo[2] + shp[2]].cpu().numpy()
        want = np.argmax(pred, 0).astype(np.uint8)
        assert (want != d["data_array"][0]).mean() < 1e-3          # normalisation differs by 1 ulp: ties only

    # FPL branch: MC-dropout uncertainty list, sorted ascending, saved as the reference's npy
    config["testing"]["fpl"] = True
    config["testing"]["fpl_uncertainty_sorted"] = str(tmp_path / "uncertainty_sorted.npy")
    srt = agent.infer(mc_passes=4)
    rows = np.load(str(tmp_path / "uncertainty_sorted.npy"), allow_pickle=True)
    assert len(rows) == 2 and [r[1] for r in rows] == [r[1] for r in srt]
    assert rows[0][0][0] <= rows[1][0][0]
    # weight csv (get image_weight.py) from a hand-made uncertainty list with distinct values
    fake = [([0.2], "./dataset/hrT2_train/img/a.nii.gz"), ([0.5], "./dataset/hrT2_train/img/b.nii.gz"),
            ([1], "./dataset/hrT2_train/img/c.nii.gz")]
    w = flt.write_weight_csv(fake, str(tmp_path / "w.csv"), "./dataset/hrT2_train/img", "./results/masks", "dataset/pw")
    assert w == pytest.approx([1.01, 0.01, 0.01])
    lines = (tmp_path / "w.csv").read_text().strip().splitlines()
    assert lines[0] == "image,label,pixel_weight,image_weight"
    assert lines[1].split(",")[:3] == ["./dataset/hrT2_train/img/a.nii.gz", "./results/masks/a.nii.gz", "dataset/pw/a.nii.gz"]
    # pixel-weight volumes from two prediction folders
    other = tmp_path / "results_cyc"
    other.mkdir()
    for i, shp in enumerate(shapes):
        m = nifti.load_nifty_volume_as_4d_array(str(out_dir / ("c%d.nii.gz" % i)))["data_array"][0]
        m2 = m.copy()
        m2[2:5, 4:9, 6:20] ^= 1
        nifti.write_nifti(str(other / ("c%d.nii.gz" % i)), m2)
    names = flt.write_pixel_weight_volumes(str(out_dir), str(other), str(tmp_path / "pw"))
    assert names == ["c0.nii.gz", "c1.nii.gz"]
    pw = nifti.load_nifty_volume_as_4d_array(str(tmp_path / "pw" / "c0.nii.gz"))["data_array"][0]
    want = np.ones(shapes[0], np.float32)
    want[2:5, 4:9, 6:20] = 0.5
    assert np.array_equal(pw, want)
    # stage 3 reads what stage 2 wrote
    (tmp_path / "s3.csv").write_text("image,label,pixel_weight,image_weight\n"
                                     "data/img/c0.nii.gz,results/vs_t1s_g_test/c0.nii.gz,pw/c0.nii.gz,0.9\n")
    s = fplx.NiftyDataset(str(tmp_path), str(tmp_path / "s3.csv"), with_label=True)[0]
    got = s["pixel_weight"].cpu().numpy()[0]
    assert np.array_equal(got, np.where(want < 1, 0, want) * np.float32(0.9))
    assert tuple(s["label"].shape) == (1,) + shapes[0]


@pytest.mark.parametrize("case", ["auto", "given"])
def test_normalize_ignore_non_positive_matches_reference_fixture(golden_dir, case):
    """NormalizeWithMeanStd_ignore_non_positive (normalize.py:39, 55-66) against the reference's own output
    (tests/golden/make_golden_normalize_np.py): moments over the voxels > 0 (or the given ones), the non-positive voxels
    replaced by numpy.random.normal(0, 1) drawn from numpy's global generator in the reference's order - bit-exact there, the
    normalised voxels within the float32 rounding of the moments (numpy sums pairwise in float32, the kernel in float64)."""
    from fplx import transform as T
    g = np.load(os.path.join(golden_dir, "normalize_np.npz"))
    given = case == "given"
    p = {"task": "segmentation", "normalizewithmeanstd_channels": [0, 1],
         "normalizewithmeanstd_mean": [float(v) for v in g["given_mean"]] if given else None,
         "normalizewithmeanstd_std": [float(v) for v in g["given_std"]] if given else None,
         "normalizewithmeanstd_ignore_non_positive": True}
    np.random.seed(int(g["seed"]))
    got = T.NormalizeWithMeanStd(p)({"image": _dev(g["image"])})["image"].cpu().numpy()
    want, bg = g["out_" + case], g["image"] <= 0
    assert bg.any() and (~bg).any()
    assert np.array_equal(got[bg], want[bg])                                  # the host draw, cast to float32
    np.testing.assert_allclose(got[~bg], want[~bg], rtol=NORM_RTOL, atol=NORM_ATOL)
    # the moments really are those of the positive voxels (a whole-image mean would shift every value by > 0.5)
    if not given:
        x = g["image"][0].astype(np.float64)
        m, sd = x[x > 0].mean(), x[x > 0].std()
        np.testing.assert_allclose(got[0][~bg[0]], ((x - m) / sd)[~bg[0]], rtol=1e-5, atol=1e-5)

"""-m gpu parity of fplx.evaluation (csrc/sample.hip overlap_counts_k + float64 host formulas) against the fixtures the
reference's own evaluation code produced (tests/golden/make_golden_eval.py): scores and the csv reports."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CASES = (("l12", [1, 2], False), ("l1", [1], False), ("fuse12", [1, 2], True))


def _fx(golden_dir):
    return np.load(os.path.join(golden_dir, "evaluation.npz"))


def test_scores_match_reference_fixture(golden_dir):
    from fplx import evaluation as E
    g = _fx(golden_dir)
    sp = tuple(g["spacing"])
    for metric in ("dice", "iou", "rve", "volume"):
        for tag, labels, fuse in CASES:
            want = g["%s_%s" % (metric, tag)]
            got = np.array([E.get_multi_class_evaluation_score(g["s%d" % i][None], g["g%d" % i][None], labels, fuse, sp,
                                                               metric) for i in range(6)], np.float64)
            if metric == "iou":        # the reference's float32 union sum + NumPy-2 promotion: see fplx/evaluation.py
                np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9)
            else:
                np.testing.assert_array_equal(got, want)                        # bit-identical float64
    s, gt = g["s1"] == 1, g["g1"] == 1
    assert E.binary_dice(s, gt) == g["dice_l1"][1, 0]
    assert E.binary_dice(torch.from_numpy(s).cuda(), torch.from_numpy(gt).cuda()) == g["dice_l1"][1, 0]
    assert E.binary_relative_volume_error(s, gt) == g["rve_l1"][1, 0]
    assert E.get_binary_evaluation_score(s[None], gt[None], sp, "dice") == g["dice_l1"][1, 0]


def test_counts_at_full_volume_size_and_errors():
    from fplx import evaluation as E, ops
    rs = np.random.RandomState(1)
    s = rs.randint(0, 4, (48, 160, 272)).astype(np.uint8)
    g = rs.randint(0, 4, (48, 160, 272)).astype(np.uint8)
    got = ops.overlap_counts(torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda(), [1, 2, 3])
    want = [[int(((s == l) & (g == l)).sum()), int((s == l).sum()), int((g == l).sum())] for l in (1, 2, 3)]
    assert got == want
    fused = ops.overlap_counts(torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda(), [1, 3], fuse=True)
    sm, gm = np.isin(s, [1, 3]), np.isin(g, [1, 3])
    assert fused == [[int((sm & gm).sum()), int(sm.sum()), int(gm.sum())]]
    assert E.get_multi_class_evaluation_score(np.zeros((4, 4, 4)), np.zeros((4, 4, 4)), [1], False, (1, 1, 1), "dice") == [1.0]
    with pytest.raises(ValueError):
        E.get_multi_class_evaluation_score(s, g, [1], False, (1, 1, 1), "assd")
    with pytest.raises(ValueError):
        E.get_multi_class_evaluation_score(s, g, [1], False, (1, 1, 1), "f1")
    with pytest.raises(ValueError):
        ops.overlap_counts(torch.zeros(4, dtype=torch.uint8).cuda(), torch.zeros(5, dtype=torch.uint8).cuda(), [1])


def test_evaluation_1_csv_reports_match_reference(tmp_path, golden_dir):
    """files in, csv out: the same masks as .nii.gz, evaluation_1 -> byte-identical csv text for dice (two labels +
    average, mean / std rows); the iou report to 1e-6."""
    from fplx import evaluation as E, nifti
    g = _fx(golden_dir)
    want = json.load(open(os.path.join(golden_dir, "evaluation_csv.json")))
    sp = tuple(float(v) for v in g["spacing"])
    seg_dir = tmp_path / "seg" / "vs_t1s_g_test"
    (tmp_path / "gt").mkdir()
    seg_dir.mkdir(parents=True)
    for i in range(6):
        nifti.write_nifti(str(tmp_path / "gt" / ("lab%d.nii.gz" % i)), g["g%d" % i].astype(np.int16), sp[::-1])
        nifti.write_nifti(str(seg_dir / ("case%d.nii.gz" % i)), g["s%d" % i], sp[::-1])
    (tmp_path / "test_pair.csv").write_text("ground_truth,segmentation\n" + "".join(
        "lab%d.nii.gz,case%d.nii.gz\n" % (i, i) for i in range(4)))
    (tmp_path / "valid_pair.csv").write_text("ground_truth,segmentation\n" + "".join(
        "lab%d.nii.gz,case%d.nii.gz\n" % (i, i) for i in (4, 5)))
    for metric, labels in (("dice", [1, 2]), ("iou", [1])):
        cfg = {"evaluation": {"metric_1": metric, "label_list": labels, "organ_name": "vs",
                              "ground_truth_folder_root": str(tmp_path / "gt"),
                              "test_evaluation_image_pair": str(tmp_path / "test_pair.csv"),
                              "valid_evaluation_image_pair": str(tmp_path / "valid_pair.csv")},
               "testing": {"output_dir": str(tmp_path / "seg")}, "training": {"ckpt_save_dir": "model/vs_t1s_g"},
               "dataset": {"test_csv": "config/test.csv"}}
        res = E.evaluation_1(cfg)
        for part in ("test", "valid"):
            got = open(str(seg_dir / ("%s_vs_%s_all.csv" % (part, metric))), newline="").read()
            if metric == "dice":
                assert got == want["%s_%s" % (part, metric)], part
            else:
                gl, wl = got.strip().splitlines(), want["%s_%s" % (part, metric)].strip().splitlines()
                assert gl[0] == wl[0] and len(gl) == len(wl)
                for a, b in zip(gl[1:], wl[1:]):
                    assert a.split(",")[0] == b.split(",")[0]
                    np.testing.assert_allclose([float(v) for v in a.split(",")[1:]], [float(v) for v in b.split(",")[1:]],
                                               rtol=1e-6, atol=1e-8)
        assert res["test"][0].shape == ((3,) if metric == "dice" else (1,))

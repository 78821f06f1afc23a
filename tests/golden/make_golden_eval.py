"""Generates tests/golden/evaluation.npz + evaluation_csv.json by RUNNING the reference's evaluation code
(/root/reference/PyMIC/pymic/util/evaluation_seg_train.py: binary_dice 21-50, binary_iou 68-81, rve 171-186,
get_binary_evaluation_score 188-229, get_multi_class_evaluation_score 231-262, evaluation_1 263-420), imported with the
stub modules of _ref_import.py.  SimpleITK is absent, so for evaluation_1 the module's `load_image_as_nd_array` is pointed
at an in-memory table of the same synthetic masks (declared IO stand-in: the scoring and the csv writing are the reference's
own code).  ASSD / HD95 need GeodisTK and are not generated.  Build-container only; the GPU box reads the fixtures."""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install()
import pymic.util.evaluation_seg_train as E  # noqa: E402

SPACING = (1.5, 0.41, 0.41)
SHAPE = (14, 40, 52)


def masks():
    """6 (segmentation, ground truth) pairs with labels {0,1,2}: shifted / scaled ellipsoids, one empty segmentation"""
    zz, yy, xx = np.meshgrid(*[np.arange(n) for n in SHAPE], indexing="ij")
    out = []
    for i in range(6):
        def blob(cz, cy, cx, r):
            return ((zz - cz) ** 2 * 6 + (yy - cy) ** 2 + (xx - cx) ** 2) < r * r
        g = np.zeros(SHAPE, np.uint8)
        g[blob(6, 18, 22, 9)] = 1
        g[blob(8, 26, 36, 6)] = 2
        s = np.zeros(SHAPE, np.uint8)
        if i != 4:
            s[blob(6 + (i % 2), 18 + i, 22 - i, 9 - 0.5 * i)] = 1
            s[blob(8, 26 - i, 36 + (i % 3), 6 + 0.4 * i)] = 2
        out.append((s, g))
    return out


def main():
    pairs = masks()
    out = {"spacing": np.array(SPACING)}
    for i, (s, g) in enumerate(pairs):
        out["s%d" % i], out["g%d" % i] = s, g
    for metric in ("dice", "iou", "rve", "volume"):
        for tag, labels, fuse in (("l12", [1, 2], False), ("l1", [1], False), ("fuse12", [1, 2], True)):
            sc = [E.get_multi_class_evaluation_score(s[None], g[None], labels, fuse, SPACING, metric) for s, g in pairs]
            out["%s_%s" % (metric, tag)] = np.asarray(sc, np.float64)
    np.savez_compressed(os.path.join(HERE, "evaluation.npz"), **out)

    # evaluation_1 end to end (csv format), masks served from memory
    table = {}
    for i, (s, g) in enumerate(pairs):
        table["gt/lab%d.nii.gz" % i] = g
        table["seg/vs_t1s_g_test/case%d.nii.gz" % i] = s
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "seg", "vs_t1s_g_test"))

    def loader(name):
        rel = os.path.relpath(name, tmp)
        return {"data_array": table[rel][None], "spacing": SPACING, "origin": (0, 0, 0), "direction": tuple(np.eye(3).ravel())}
    E.load_image_as_nd_array = loader
    with open(os.path.join(tmp, "test_pair.csv"), "w") as f:
        f.write("ground_truth,segmentation\n" + "".join("lab%d.nii.gz,case%d.nii.gz\n" % (i, i) for i in range(4)))
    with open(os.path.join(tmp, "valid_pair.csv"), "w") as f:
        f.write("ground_truth,segmentation\n" + "".join("lab%d.nii.gz,case%d.nii.gz\n" % (i, i) for i in (4, 5)))
    csvs = {}
    for metric, labels in (("dice", [1, 2]), ("iou", [1])):
        cfg = {"evaluation": {"metric_1": metric, "label_list": labels, "organ_name": "vs",
                              "ground_truth_folder_root": os.path.join(tmp, "gt"),
                              "test_evaluation_image_pair": os.path.join(tmp, "test_pair.csv"),
                              "valid_evaluation_image_pair": os.path.join(tmp, "valid_pair.csv")},
               "testing": {"output_dir": os.path.join(tmp, "seg")},
               "training": {"ckpt_save_dir": "model/vs_t1s_g"},
               "dataset": {"test_csv": "config/test.csv"}}
        E.evaluation_1(cfg)
        for part in ("test", "valid"):
            p = os.path.join(tmp, "seg", "vs_t1s_g_test", "%s_vs_%s_all.csv" % (part, metric))
            csvs["%s_%s" % (part, metric)] = open(p, newline="").read()
    json.dump(csvs, open(os.path.join(HERE, "evaluation_csv.json"), "w"), indent=1)
    print(csvs["test_dice"])
    print(out["rve_l12"], out["volume_l1"])


if __name__ == "__main__":
    main()

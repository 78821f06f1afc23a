"""Generates tests/golden/normalize_np.npz by RUNNING the reference's NormalizeWithMeanStd with
NormalizeWithMeanStd_ignore_non_positive = True (/root/reference/PyMIC/pymic/transform/normalize.py:39, 55-66, imported with
the stub modules of _ref_import.py) on a deterministic two-channel volume whose background is zero / negative, with numpy's
global generator seeded: the host mirror (fplx/transform.py) draws the replacement noise from the same generator in the same
order.  Two cases: moments computed (mean = std = None) and moments given.  Build-container only; the GPU box reads the .npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install()
import detdata  # noqa: E402
from pymic.transform.normalize import NormalizeWithMeanStd  # noqa: E402

SHAPE = (2, 6, 20, 24)
SEED = 20240
GIVEN = ([180.0, 95.5], [41.0, 17.25])


def image():
    img = (detdata.normal("npn.image", SHAPE) * 60.0 + 120.0).astype(np.float32)
    zz, yy, xx = np.meshgrid(np.arange(SHAPE[1]), np.arange(SHAPE[2]), np.arange(SHAPE[3]), indexing="ij")
    img[:, ((yy - 10) ** 2 + (xx - 12) ** 2) > 81] = 0.0                   # zero background outside a disc; some negatives inside
    return img


def main():
    out = {"image": image(), "seed": np.array(SEED), "given_mean": np.array(GIVEN[0]), "given_std": np.array(GIVEN[1])}
    for name, mean, std in (("auto", None, None), ("given", list(GIVEN[0]), list(GIVEN[1]))):
        p = {"task": "segmentation", "normalizewithmeanstd_channels": [0, 1], "normalizewithmeanstd_mean": mean,
             "normalizewithmeanstd_std": std, "normalizewithmeanstd_ignore_non_positive": True}
        np.random.seed(SEED)
        s = NormalizeWithMeanStd(p)({"image": image()})
        out["out_" + name] = s["image"]
        assert s["image"].dtype == np.float32
    np.savez_compressed(os.path.join(HERE, "normalize_np.npz"), **out)
    print({k: (v.shape, v.dtype) for k, v in out.items()})


if __name__ == "__main__":
    main()

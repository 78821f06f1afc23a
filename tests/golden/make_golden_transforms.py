"""Generates tests/golden/transforms.npz by RUNNING the reference's numpy transforms
(/root/reference/PyMIC/pymic/transform/{normalize,pad,crop,flip,label_convert}.py, imported with the stub modules of
_ref_import.py) on deterministic inputs, in the order of the shipped configs
(config_dual/data_vs/vs_t1s_g.cfg:21  train_transform = [NormalizeWithMeanStd, Pad, RandomCrop, RandomFlip,
LabelToProbability]).  Python's `random` module is seeded per case: the host mirror (fplx/transform.py) draws from the
same generator in the same order, so crop positions and flip axes are reproduced, not just the data movement.
Build-container only; the GPU box reads the .npz."""
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install()
import detdata  # noqa: E402
from pymic.transform.normalize import NormalizeWithMeanStd  # noqa: E402
from pymic.transform.pad import Pad  # noqa: E402
from pymic.transform.crop import RandomCrop  # noqa: E402
from pymic.transform.flip import RandomFlip  # noqa: E402
from pymic.transform.label_convert import LabelToProbability  # noqa: E402

PARAMS = {
    "task": "segmentation",
    "normalizewithmeanstd_channels": [0], "normalizewithmeanstd_mean": None, "normalizewithmeanstd_std": None,
    "pad_output_size": [16, 48, 48], "pad_ceil_mode": False,
    "randomcrop_output_size": [16, 32, 32], "randomcrop_foreground_focus": True, "randomcrop_foreground_ratio": 0.5,
    "randomcrop_mask_label": [1],
    "randomflip_flip_depth": False, "randomflip_flip_height": True, "randomflip_flip_width": True,
    "labeltoprobability_class_num": 2,
}
SHAPE = (12, 40, 50)            # D, H, W of the synthetic volume (smaller than Pad_output_size along every axis or not)
SEEDS = [1, 2, 3, 5, 8, 13]


def inputs():
    img = (detdata.normal("tf.image", (1,) + SHAPE) * 37.0 + 210.0).astype(np.float32)      # MR-like intensities
    zz, yy, xx = np.meshgrid(np.arange(SHAPE[0]), np.arange(SHAPE[1]), np.arange(SHAPE[2]), indexing="ij")
    lab = (((zz - 7) ** 2 * 4 + (yy - 25) ** 2 + (xx - 31) ** 2) < 36).astype(np.uint8)[None]
    pw = (detdata.uniform("tf.pw", (1,) + SHAPE) > 0.3).astype(np.float32) * 0.73
    return img, lab, pw


def main():
    img, lab, pw = inputs()
    out = {"image": img, "label": lab, "pixel_weight": pw, "params_json": np.array(json.dumps(PARAMS))}
    chain = lambda: [NormalizeWithMeanStd(dict(PARAMS)), Pad(dict(PARAMS)), RandomCrop(dict(PARAMS)),
                     RandomFlip(dict(PARAMS)), LabelToProbability(dict(PARAMS))]
    for seed in SEEDS:
        random.seed(seed)
        np.random.seed(seed)
        s = {"image": img.copy(), "label": lab.copy(), "pixel_weight": pw.copy()}
        for t in chain():
            s = t(s)
            name = type(t).__name__
            if name in ("NormalizeWithMeanStd", "Pad") and seed == SEEDS[0]:
                out["after_%s_image" % name] = s["image"].copy()
                if name == "Pad":
                    out["after_Pad_label"] = s["label"].copy()
                    out["after_Pad_pixel_weight"] = s["pixel_weight"].copy()
        k = "seed%d_" % seed
        out[k + "image"] = s["image"]
        out[k + "label"] = s["label"]
        out[k + "label_prob"] = s["label_prob"]
        out[k + "pixel_weight"] = s["pixel_weight"]
        out[k + "crop_param"] = np.array(s["RandomCrop_Param"])
        out[k + "flip_param"] = np.array(s["RandomFlip_Param"])
        out[k + "pad_param"] = np.array(s["Pad_Param"])
    np.savez_compressed(os.path.join(HERE, "transforms.npz"), **out)
    for seed in SEEDS:
        print(seed, out["seed%d_crop_param" % seed], out["seed%d_flip_param" % seed])


if __name__ == "__main__":
    main()

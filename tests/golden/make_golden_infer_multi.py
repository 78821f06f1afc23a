"""Fixture for the multi-output (deep supervision) path of the Inferer: RUNS the reference's Inferer
(/root/reference/PyMIC/pymic/net_run_dsbn/infer_func.py:113-140, 188-222) on CPU with a weight-free toy network that
returns two tensors at scales 1 and 1/2, sliding window with overlap, tta_mode 0 and 1.
    python tests/golden/make_golden_infer_multi.py        (build container only: imports the reference)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402
import detdata  # noqa: E402

_ref_import.install()
from pymic.net_run_dsbn.infer_func import Inferer  # noqa: E402


def toy_multi(x, domain_label=None):
    """two class channels from the one input channel; second output at half resolution"""
    a = torch.cat([torch.sin(x) + 0.5 * x, torch.cos(2.0 * x) - 0.25 * x * x], 1)
    return [a, torch.nn.functional.avg_pool3d(a, 2, 2) * 1.5]


def main():
    x = torch.from_numpy(detdata.normal("infmulti.x", (1, 1, 24, 40, 40)))
    dl = torch.ones(1, dtype=torch.long)
    out = {}
    for tta in (0, 1):
        cfg = dict(sliding_window_enable=True, sliding_window_size=[16, 16, 24], sliding_window_stride=[8, 12, 16], tta_mode=tta,
                   class_num=2)
        r = Inferer(cfg).run(toy_multi, x, dl)
        assert isinstance(r, list) and len(r) == 2
        out["tta%d.out0" % tta], out["tta%d.out1" % tta] = r[0].numpy(), r[1].numpy()
    np.savez_compressed(os.path.join(HERE, "inferer_multi.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

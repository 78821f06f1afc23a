"""Inferer host logic on CPU: the generic (any-model, several-outputs) path against a fixture produced by RUNNING the
reference's Inferer (tests/golden/make_golden_infer_multi.py), and the tile plan."""
import os

import numpy as np
import pytest
import torch

import detdata


def _toy_multi(x, domain_label=None):
    a = torch.cat([torch.sin(x) + 0.5 * x, torch.cos(2.0 * x) - 0.25 * x * x], 1)
    return [a, torch.nn.functional.avg_pool3d(a, 2, 2) * 1.5]


def test_multi_output_inferer_matches_reference_run(golden_dir):
    import fplx
    g = np.load(os.path.join(golden_dir, "inferer_multi.npz"))
    x = torch.from_numpy(detdata.normal("infmulti.x", (1, 1, 24, 40, 40)))
    dl = torch.ones(1, dtype=torch.long)
    for tta in (0, 1):
        cfg = dict(sliding_window_enable=True, sliding_window_size=[16, 16, 24], sliding_window_stride=[8, 12, 16], tta_mode=tta,
                   class_num=2)
        r = fplx.Inferer(cfg).run(_toy_multi, x, dl)
        assert isinstance(r, list) and len(r) == 2
        for i in range(2):
            np.testing.assert_allclose(r[i].numpy(), g["tta%d.out%d" % (tta, i)], rtol=0, atol=1e-6)


def test_single_output_generic_path_and_plan():
    import fplx
    from fplx.infer import _axis_starts
    assert _axis_starts(40, 16, 12) == [0, 12, 24, 24]            # the clamped duplicate is a tile of its own
    assert _axis_starts(32, 32, 32) == [0]
    x = torch.from_numpy(detdata.normal("infplan.x", (2, 1, 8, 24, 24)))

    def model(t, domain_label=None):
        return torch.cat([t, -t], 1)
    cfg = dict(sliding_window_enable=True, sliding_window_size=[8, 16, 16], sliding_window_stride=[8, 8, 8], tta_mode=1, class_num=2)
    out = fplx.Inferer(cfg).run(model, x, torch.zeros(2, dtype=torch.long))
    np.testing.assert_allclose(out.numpy(), torch.cat([x, -x], 1).numpy(), atol=1e-6)      # identity network: averaging is exact-ish
    with pytest.raises(ValueError):
        fplx.Inferer({"tta_mode": 5}).run(model, x, None)

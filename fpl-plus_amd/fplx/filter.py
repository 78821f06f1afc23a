"""Pseudo-label uncertainty filter and weight arithmetic of FPL+ on the GPU.

Reference call sites (paths under /root/reference):
  PyMIC/pymic/net_run_dsbn/agent_seg.py:897-931, 954-961   MC / TTA statistics -> image uncertainty
  PyMIC/pymic/net_run_dsbn/agent_seg.py:1049-1050          hard pseudo-label (softmax -> argmax -> uint8)
  data/get_pixel_weight.py:21-26, merge_pixelw.py:21-27    mask disagreement -> pixel weight
  PyMIC/pymic/io/nifty_dataset.py:165-168                  set_weight_
  data/get image_weight.py:10-28                           uncertainty -> image weight (host, 100 floats)
"""
import torch

from . import ops

fpl_filter = ops.mc_filter
hard_label = ops.hard_label
pixel_weight_from_masks = ops.pixel_weight


def fpl_uncertainty(logits_tcv, thr=0.01):
    """-> python dict(uncer_one, vars, boundary, hards) for one volume (one host sync)."""
    r = ops.mc_filter(logits_tcv, thr)
    st = r["stats"].cpu()
    b = int(st[1].item())
    return dict(uncer_one=(1 if b < 50 else float(st[2].item())), vars=float(st[0].item()), boundary=b,
                hards=r["hards"])


def sort_uncertainty(uncer_by_name):
    """agent_seg.py:957-959: ascending list of ([uncertainty], name)."""
    return sorted(zip([[v] for v in uncer_by_name.values()], uncer_by_name.keys()), reverse=False)


def image_weights(rows):
    """`data/get image_weight.py`:10-28 on [(uncertainty, path)]: host arithmetic in float64."""
    allw = [u for u, _ in rows if u != 1]
    mx, mn = max(allw), min(allw)
    out = []
    for u, _ in rows:
        u = mx if u > mx else u
        out.append(abs((mx - u) / (mx - mn)) + 0.01)
    return out

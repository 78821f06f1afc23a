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


def write_pixel_weight_volumes(pseudo_target_root, pseudo_fake_source_root, out_dir):
    """merge_pixelw.py:6-30 / data/get_pixel_weight.py:6-30: for every mask name present in both prediction folders the
    weight volume 1 - 0.5 * (a XOR b), computed on the GPU, written as a float32 .nii.gz WITHOUT geometry (the reference
    writes sitk.GetImageFromArray output as is).  Returns the names written."""
    import os
    import numpy as np
    from .nifti import load_nifty_volume_as_4d_array, write_nifti
    a_names = sorted(n for n in os.listdir(pseudo_target_root) if '.nii.gz' in n)
    b_names = sorted(n for n in os.listdir(pseudo_fake_source_root) if '.nii.gz' in n)
    assert len(a_names) == len(b_names)
    os.makedirs(out_dir, exist_ok=True)
    for name in a_names:
        a = load_nifty_volume_as_4d_array(os.path.join(pseudo_target_root, name))['data_array'][0]
        b = load_nifty_volume_as_4d_array(os.path.join(pseudo_fake_source_root, name))['data_array'][0]
        assert a.shape == b.shape
        ta = torch.from_numpy(np.ascontiguousarray(a.astype(np.uint8))).cuda()
        tb = torch.from_numpy(np.ascontiguousarray(b.astype(np.uint8))).cuda()
        w = ops.pixel_weight(ta, tb)
        write_nifti(os.path.join(out_dir, name), w.cpu().numpy().reshape(a.shape))
    return a_names


def write_weight_csv(sorted_rows, out_csv, img_dir, label_dir, weight_dir):
    """`data/get image_weight.py`:19-40: one csv row (image, label, pixel_weight, image_weight) per case of the sorted
    uncertainty list; the label / weight paths are the image path with img_dir replaced."""
    import csv
    rows = [(u[0] if isinstance(u, (list, tuple)) else u, n) for u, n in sorted_rows]
    weights = image_weights(rows)
    with open(out_csv, mode='w') as f:
        wr = csv.writer(f, delimiter=',', quotechar='"', quoting=csv.QUOTE_MINIMAL)
        wr.writerow(['image', 'label', 'pixel_weight', 'image_weight'])
        for (u, name), w in zip(rows, weights):
            wr.writerow([name, name.replace(img_dir, label_dir), name.replace(img_dir, weight_dir), w])
    return weights

"""Segmentation evaluation on the GPU behind PyMIC's evaluation functions.

Mirrors PyMIC/pymic/util/evaluation_seg_train.py: binary_dice (21-50), binary_iou (68-81),
binary_relative_volume_error (171-186), get_binary_evaluation_score (188-229), get_multi_class_evaluation_score
(231-262) and evaluation_1 (263-420, the csv reports `test_<organ>_<metric>_all.csv` / `valid_..._all.csv`).
The voxel counting runs in one HIP kernel for all labels (exact 64-bit integers, csrc/sample.hip overlap_counts_k); the
scores are the reference's float64 formulas on those integers, so Dice / RVE / volume are bit-identical to the numpy
results.  IoU: the reference sums a float32 union map; under NumPy >= 2 its `+ 1e-5` then stays float32, under the
NumPy 1.x it was written for it is float64 - the float64 form is used here (difference ~1e-8 relative).
ASSD / HD95 (binary_assd 137-171, binary_hd95 101-135, evaluation_2 420-560): the reference gets its distance maps from
GeodisTK.geodesic3d_raster_scan(zeros, edge, spacing, 0.0, 2) - a third-party C++ extension (version unpinned) that is
absent here.  On a constant image with lambda = 0 that raster scan converges (in one forward / backward sweep) to the
shortest 26-neighbour lattice path with Euclidean step lengths under the spacing; csrc/sample.hip computes exactly that
metric in closed form between the two edge-point sets (edge extraction and the all-pairs minimum are HIP kernels).
PARITY UNPINNED against GeodisTK itself: the tests pin the kernels to oracle/np_ref.py's literal raster scan.
"""
import csv
import os

import numpy as np
import pandas as pd
import torch

from . import ops
from .nifti import load_image_as_nd_array


def _dev_u8(a, device="cuda:0"):
    if torch.is_tensor(a):
        t = a
    else:
        a = np.asarray(a)
        if a.dtype == np.bool_:
            a = a.astype(np.uint8)
        if a.min() < 0 or a.max() > 255:
            raise ValueError("fplx.evaluation: label values outside [0, 255]")
        t = torch.from_numpy(np.ascontiguousarray(a.astype(np.uint8)))
    if t.dtype == torch.bool:
        t = t.to(torch.uint8)
    if t.dtype != torch.uint8:
        t = t.to(torch.uint8)
    return t.to(device) if not t.is_cuda else t


def _score(counts, metric, spacing):
    s0, s1, s2 = counts
    m = metric.lower()
    if m == "dice":
        return (2.0 * s0 + 1e-5) / (s1 + s2 + 1e-5)
    if m == "iou":
        return (s0 + 1e-5) / (float(s1 + s2 - s0) + 1e-5)
    if m == "rve":
        assert s2 > 0
        return abs(float(s1) - float(s2)) / float(s2)
    if m == "volume":
        voxel_size = 1.0
        for dim in range(len(spacing)):
            voxel_size = voxel_size * spacing[dim]
        return s2 * voxel_size
    if m in ("assd", "hd95"):
        return None                                            # surface metrics: not a function of the counts
    raise ValueError("unsupported evaluation metric: {0:}".format(metric))


def binary_dice(s, g, resize=False):
    assert len(s.shape) == len(g.shape)
    return _score(ops.overlap_counts(_dev_u8(s), _dev_u8(g), [1])[0], "dice", None)


def binary_iou(s, g):
    assert len(s.shape) == len(g.shape)
    return _score(ops.overlap_counts(_dev_u8(s), _dev_u8(g), [1])[0], "iou", None)


def binary_relative_volume_error(s, g):
    return _score(ops.overlap_counts(_dev_u8(s), _dev_u8(g), [1])[0], "rve", None)


def _squeeze_like_reference(t):
    """get_binary_evaluation_score 203-209: [1, D, H, W] -> [D, H, W]; a leading 1 again -> [H, W]"""
    if t.dim() == 4:
        assert t.shape[0] == 1
        t = t.reshape(t.shape[1:])
    if t.shape[0] == 1:
        t = t.reshape(t.shape[1:])
    return t


def _surface_distances(s, g, spacing):
    """-> (distances of g's edge voxels to s's edge, of s's edge voxels to g's edge) as fp32 device vectors.
    s, g: binary uint8 device volumes, 2D or 3D (the reference asserts nothing else reaches GeodisTK)."""
    s, g = _squeeze_like_reference(s), _squeeze_like_reference(g)
    dim = s.dim()
    assert dim == g.dim() and dim in (2, 3)
    if dim == 2:
        sp = (1.0, 1.0, 1.0)                                   # geodesic2d_raster_scan takes no spacing (122-123)
    else:
        sp = (1.0, 1.0, 1.0) if spacing is None else tuple(float(v) for v in spacing)
        assert len(sp) == 3
    pts = []
    for m in (s, g):
        e = torch.nonzero(ops.edge_points((m != 0).to(torch.uint8))).to(torch.int32)
        if dim == 2:
            e = torch.cat([torch.zeros((e.shape[0], 1), dtype=torch.int32, device=e.device), e], 1)
        pts.append(e.contiguous())
    s_pts, g_pts = pts
    return ops.surface_min_dist(g_pts, s_pts, sp), ops.surface_min_dist(s_pts, g_pts, sp)


def binary_assd(s, g, spacing=None):
    """evaluation_seg_train.py:137-171: (sum of s_dis over g's edge + sum of g_dis over s's edge) / (ns + ng), capped
    at 50; nan when neither volume has foreground (the reference's 0 / 0)"""
    d_g, d_s = _surface_distances(_dev_u8(s), _dev_u8(g), spacing)
    n = d_g.numel() + d_s.numel()
    if n == 0:
        return float("nan")
    assd = (float(d_g.double().sum()) + float(d_s.double().sum())) / n
    return 50 if assd > 50 else assd


def binary_hd95(s, g, spacing=None):
    """evaluation_seg_train.py:101-135: max over the two directions of sorted(dist)[int(len * 0.95)]"""
    d_g, d_s = _surface_distances(_dev_u8(s), _dev_u8(g), spacing)
    if d_g.numel() == 0 or d_s.numel() == 0:
        raise IndexError("binary_hd95: a volume without foreground has no edge points (list index out of range)")
    d1 = torch.sort(d_g)[0][int(d_g.numel() * 0.95)]
    d2 = torch.sort(d_s)[0][int(d_s.numel() * 0.95)]
    return float(max(d1, d2))


def get_binary_evaluation_score(s_volume, g_volume, spacing, metric):
    m = metric.lower()
    if m == "assd":
        return binary_assd(s_volume, g_volume, spacing)
    if m == "hd95":
        return binary_hd95(s_volume, g_volume, spacing)
    return _score(ops.overlap_counts(_dev_u8(s_volume), _dev_u8(g_volume), [1])[0], metric, spacing)


def get_multi_class_evaluation_score(s_volume, g_volume, label_list, fuse_label, spacing, metric):
    """one kernel launch for the whole label list; -> list of scores (one entry when fuse_label)"""
    _score((1, 1, 1), metric, spacing)                       # unknown metric: raise before any GPU work
    s, g = _dev_u8(s_volume), _dev_u8(g_volume)
    if metric.lower() in ("assd", "hd95"):
        fn = binary_assd if metric.lower() == "assd" else binary_hd95
        if fuse_label:
            lab = torch.tensor([int(v) for v in label_list], dtype=torch.uint8, device=s.device)
            return [fn(torch.isin(s, lab).to(torch.uint8), torch.isin(g, lab).to(torch.uint8), spacing)]
        return [fn((s == int(l)).to(torch.uint8), (g == int(l)).to(torch.uint8), spacing) for l in label_list]
    counts = ops.overlap_counts(s, g, label_list, bool(fuse_label))
    return [_score(c, metric, spacing) for c in counts]


def convert_label(label, source_list, target_list):
    """util/image_process.py convert_label: labels not in source_list become 0"""
    label = np.asarray(label)
    out = np.zeros_like(label)
    for s, t in zip(source_list, target_list):
        out[label == s] = t
    return out


def _evaluate_pairs(items, gt_root, seg_root, label_list, label_fuse, metric, conv):
    score_all, rows = [], []
    for i in range(len(items)):
        gt_name, seg_name = items.iloc[i, 0], items.iloc[i, 1]
        s_dict = load_image_as_nd_array(seg_root + '/' + seg_name)
        g_dict = load_image_as_nd_array(gt_root + '/' + gt_name)
        s_volume, g_volume = s_dict["data_array"], g_dict["data_array"]
        if conv[0] is not None and conv[1] is not None:
            g_volume = convert_label(g_volume, conv[0], conv[1])
        if conv[2] is not None and conv[3] is not None:
            s_volume = convert_label(s_volume, conv[2], conv[3])
        vec = get_multi_class_evaluation_score(s_volume, g_volume, label_list, label_fuse, s_dict["spacing"], metric)
        if len(label_list) > 1:
            vec.append(np.asarray(vec).mean())
        score_all.append(vec)
        rows.append([seg_name] + vec)
    score_all = np.asarray(score_all)
    mean, std = score_all.mean(axis=0), score_all.std(axis=0)
    rows.append(['mean'] + list(mean))
    rows.append(['std'] + list(std))
    return rows, mean, std


def evaluation_1(config):
    """evaluation_seg_train.py:263-420: score every (ground truth, segmentation) pair of the test and the valid csv with
    `metric_1` and write `<seg_root>/{test,valid}_<organ>_<metric>_all.csv`; returns {'test': (mean, std), 'valid': ..}"""
    return _evaluation(config, 'metric_1')


def evaluation_2(config):
    """evaluation_seg_train.py:420-560: the same report for `metric_2` (assd in every shipped cfg)"""
    return _evaluation(config, 'metric_2')


def _evaluation(config, metric_key):
    ev = config['evaluation']
    metric, label_list, organ_name = ev[metric_key], ev['label_list'], ev['organ_name']
    label_fuse = config.get('label_fuse', False)
    gt_root = ev['ground_truth_folder_root']
    ckpt_dir = config['training']['ckpt_save_dir'].split('/')[-1]
    subset = config['dataset']['test_csv'].split('/')[-1][:-4]
    seg_root = os.path.join(config['testing']['output_dir'], ckpt_dir + '_' + subset)
    conv = (config.get('ground_truth_label_convert_source', None), config.get('ground_truth_label_convert_target', None),
            config.get('segmentation_label_convert_source', None), config.get('segmentation_label_convert_target', None))
    out = {}
    for part in ('test', 'valid'):
        items = pd.read_csv(ev[part + '_evaluation_image_pair'])
        rows, mean, std = _evaluate_pairs(items, gt_root, seg_root, label_list, label_fuse, metric, conv)
        score_csv = "{0:}/{1:}_{2:}_{3:}_all.csv".format(seg_root, part, organ_name, metric)
        with open(score_csv, mode='w') as f:
            wr = csv.writer(f, delimiter=',', quotechar='"', quoting=csv.QUOTE_MINIMAL)
            head = ['image'] + ["class_{0:}".format(i) for i in label_list]
            if len(label_list) > 1:
                head = head + ["average"]
            wr.writerow(head)
            for item in rows:
                wr.writerow(item)
        out[part] = (mean, std)
    return out

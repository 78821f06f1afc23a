"""Segmentation evaluation on the GPU behind PyMIC's evaluation functions.

Mirrors PyMIC/pymic/util/evaluation_seg_train.py: binary_dice (21-50), binary_iou (68-81),
binary_relative_volume_error (171-186), get_binary_evaluation_score (188-229), get_multi_class_evaluation_score
(231-262) and evaluation_1 (263-420, the csv reports `test_<organ>_<metric>_all.csv` / `valid_..._all.csv`).
The voxel counting runs in one HIP kernel for all labels (exact 64-bit integers, csrc/sample.hip overlap_counts_k); the
scores are the reference's float64 formulas on those integers, so Dice / RVE / volume are bit-identical to the numpy
results.  IoU: the reference sums a float32 union map; under NumPy >= 2 its `+ 1e-5` then stays float32, under the
NumPy 1.x it was written for it is float64 - the float64 form is used here (difference ~1e-8 relative).
ASSD / HD95 need GeodisTK's raster-scan geodesic distance (third-party, absent): asking for them raises ValueError.
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
        raise ValueError("fplx.evaluation: {0:} needs GeodisTK's geodesic distance, which this build does not "
                         "restate".format(metric))
    raise ValueError("unsupported evaluation metric: {0:}".format(metric))


def binary_dice(s, g, resize=False):
    assert len(s.shape) == len(g.shape)
    return _score(ops.overlap_counts(_dev_u8(s), _dev_u8(g), [1])[0], "dice", None)


def binary_iou(s, g):
    assert len(s.shape) == len(g.shape)
    return _score(ops.overlap_counts(_dev_u8(s), _dev_u8(g), [1])[0], "iou", None)


def binary_relative_volume_error(s, g):
    return _score(ops.overlap_counts(_dev_u8(s), _dev_u8(g), [1])[0], "rve", None)


def get_binary_evaluation_score(s_volume, g_volume, spacing, metric):
    return _score(ops.overlap_counts(_dev_u8(s_volume), _dev_u8(g_volume), [1])[0], metric, spacing)


def get_multi_class_evaluation_score(s_volume, g_volume, label_list, fuse_label, spacing, metric):
    """one kernel launch for the whole label list; -> list of scores (one entry when fuse_label)"""
    _score((1, 1, 1), metric, spacing)                       # unknown / unsupported metric: raise before any GPU work
    counts = ops.overlap_counts(_dev_u8(s_volume), _dev_u8(g_volume), label_list, bool(fuse_label))
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
    """evaluation_seg_train.py:263-420: score every (ground truth, segmentation) pair of the test and the valid csv
    and write `<seg_root>/{test,valid}_<organ>_<metric>_all.csv`; returns {'test': (mean, std), 'valid': (mean, std)}"""
    ev = config['evaluation']
    metric, label_list, organ_name = ev['metric_1'], ev['label_list'], ev['organ_name']
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

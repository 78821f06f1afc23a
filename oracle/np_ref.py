"""CPU ORACLE (test infrastructure, NOT product code) - numpy restatement of the
pseudo-label uncertainty filter and weight arithmetic of FPL+.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Pinned by tests/test_oracle_golden.py against fixtures produced by running the reference
(tests/golden/fpl_filter.npz, pixel_weight.npz) and against the reference's own
known-answer data pair (tests/golden/image_weight_kat.json).
Reference paths are relative to /root/reference.
PARITY UNPINNED for ev_raster_scan / ev_binary_assd / ev_binary_hd95 only: they restate GeodisTK (third-party, absent,
version unpinned by the reference) from its published algorithm; everything else here is pinned as above.
"""
import numpy as np


def softmax(x, axis):
    """scipy.special.softmax as the reference calls it (PyMIC/pymic/net_run_dsbn/agent_seg.py:911,
    1049): exp(x - max) / sum, in the input dtype (float32)."""
    m = np.amax(x, axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / np.sum(e, axis=axis, keepdims=True)


def fpl_filter(logits_stack):
    """FPL branch of SegmentationAgent.infer (agent_seg.py:897-931) for ONE volume.

    logits_stack: float32 [T, C, D, H, W] - the T (=6 in the reference, line 898) MC/TTA
    predictions, each one the [1, C, D, H, W] array `pred` of line 905-909.
    Returns dict with maps [T,C,D,H,W] f32, hards [T,D,H,W] u8, vars, means, uncertainty,
    boundary (int) and uncer_one."""
    maps, hards = None, None
    for i in range(logits_stack.shape[0]):
        pred = logits_stack[i:i + 1]
        prob = softmax(pred, axis=1)                                   # line 911
        hard = np.asarray(np.argmax(prob, axis=1), np.uint8)           # line 914
        if i == 0:
            maps, hards = prob, hard
        else:
            maps = np.concatenate((maps, prob), axis=0)                # lines 919-920
            hards = np.concatenate((hards, hard), axis=0)
    vars_ = maps.var(axis=0).sum()                                     # line 921
    means = np.mean(maps[:, 1], axis=0)                                # line 922
    uncertainty = -1.0 * (means * np.log(means + 1e-6))                # line 923
    boundary = np.where(uncertainty > 0.01, 1, 0).sum()                # line 924
    uncer_one = 1 if boundary < 50 else vars_ / boundary               # lines 926-929
    return dict(maps=maps, hards=hards, vars=vars_, means=means, uncertainty=uncertainty,
                boundary=int(boundary), uncer_one=uncer_one)


def sort_uncertainty(uncer_by_name):
    """agent_seg.py:957-959: ascending sort of ([uncer], name) tuples."""
    pairs = list(zip([[v] for v in uncer_by_name.values()], uncer_by_name.keys()))
    return sorted(pairs, reverse=False)


def hard_label(logits):
    """save_outputs (agent_seg.py:1049-1050): softmax -> argmax -> uint8, [N,C,...] -> [N,...]."""
    return np.asarray(np.argmax(softmax(logits, axis=1), axis=1), np.uint8)


def pixel_weight_from_masks(a, b):
    """data/get_pixel_weight.py:21-26 (= merge_pixelw.py:21-27): union, intersection, xor ->
    weight 1 where the two masks agree, 0.5 where they differ (float64)."""
    both = a + b
    both[both > 1] = 1
    and_arr = b * a
    sub = both - and_arr
    return np.ones_like(sub) - sub * 0.5


def set_weight(img_weight, pixel_weight):
    """NiftyDataset.set_weight_ (PyMIC/pymic/io/nifty_dataset.py:165-168): disagreeing voxels
    (w < 1) are zeroed, the rest scaled by the image weight.  Modifies its argument like the
    reference does."""
    pixel_weight[pixel_weight < 1] = 0
    return pixel_weight * img_weight


def image_weights(rows):
    """`data/get image_weight.py`:10-28.  rows: [(uncertainty, path)] in file order.
    Returns the image_weight column (python floats), same order."""
    allw = [u for u, _ in rows if u != 1]
    mx, mn = max(allw), min(allw)
    out = []
    for u, _ in rows:
        if u > mx:
            u = mx
        out.append(abs((mx - u) / (mx - mn)) + 0.01)
    return out


# ---------------------------------------------------------------------------------------
# Philox4x32-10 keep-mask: the counter-based dropout stream of the HIP path
# (fpl-plus_amd/csrc/philox.h).  The reference draws dropout from torch's global RNG; no other
# implementation can replay that, so the GPU path defines its own stream and this function
# reproduces it bit-for-bit for the parity tests.
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = [np.asarray(c, np.uint32).copy() for c in (c0, c1, c2, c3)]
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * _M0
            p1 = c2.astype(np.uint64) * _M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0, k1 = np.uint32(k0 + _W0), np.uint32(k1 + _W1)
    return c0, c1, c2, c3


def philox_keep_mask(seed, stream, n_elems, p):
    """keep[i] for the flat NDHWC element index i: word (i & 3) of
    philox(counter=(i>>2, 0, stream, 0), key=(seed_lo, seed_hi)) >= floor(float32(p) * 2^32)."""
    n4 = (n_elems + 3) // 4
    idx = np.arange(n4, dtype=np.uint32)
    z = np.zeros(n4, np.uint32)
    r = philox4x32_10(idx, z, np.full(n4, stream, np.uint32), z,
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    words = np.stack(r, axis=1).reshape(-1)[:n_elems]
    thr = np.uint32(min(int(float(np.float32(p)) * 4294967296.0), 0xFFFFFFFF))  # p travels as fp32
    return words >= thr


def dropout_masks_ncdhw(seed, step, net_params, in_shape, ps):
    """Keep-masks (NCDHW bool tensors) for the 9 ConvBlockND dropout sites of one forward,
    matching fplx's stream numbering: stream = step * 16 + block_index."""
    n, _, D, H, W = in_shape
    ft = net_params["feature_chns"]
    lv = [0, 1, 2, 3, 4, 3, 2, 1, 0]
    out = []
    for b in range(9):
        p = ps[b]
        if p <= 0:
            out.append(None)
            continue
        L = lv[b]
        d, h, w, c = D >> L, H >> L, W >> L, ft[L]
        keep = philox_keep_mask(seed, step * 16 + b, n * d * h * w * c, p)
        out.append(np.ascontiguousarray(keep.reshape(n, d, h, w, c).transpose(0, 4, 1, 2, 3)))
    return out


# ---------------------------------------------------------------------------------------------
# Training-sample transforms.  numpy restatements of the reference chain
# `train_transform = [NormalizeWithMeanStd, Pad, RandomCrop, RandomFlip, LabelToProbability]`
# (config_dual/data_vs/vs_t1s_g.cfg:21); sample = dict with 'image' [C,D,H,W], 'label' [1,D,H,W], 'pixel_weight'.
def tf_normalize(sample, chns=None, mean=None, std=None):
    """PyMIC/pymic/transform/normalize.py:43-68 (ignore_non_positive = False)"""
    image = sample['image']
    chns = chns if chns is not None else range(image.shape[0])
    for i, chn in enumerate(chns):
        m = image[chn].mean() if mean is None or mean[i] is None else mean[i]
        s = image[chn].std() if std is None or std[i] is None else std[i]
        image[chn] = (image[chn] - m) / s
    sample['image'] = image
    return sample


def tf_pad(sample, output_size, ceil_mode=False):
    """pad.py:126-163: reflect padding up to output_size, lower margin = int(margin / 2)"""
    import math
    shape = sample['image'].shape
    dim = len(shape) - 1
    if ceil_mode:
        output_size = [int(math.ceil(float(shape[1 + i]) / output_size[i])) * output_size[i] for i in range(dim)]
    margin = [max(0, output_size[i] - shape[1 + i]) for i in range(dim)]
    lo = [int(margin[i] / 2) for i in range(dim)]
    hi = [margin[i] - lo[i] for i in range(dim)]
    sample['Pad_Param'] = (lo, hi)
    pad = tuple([(0, 0)] + [(lo[i], hi[i]) for i in range(dim)])
    if max(margin) > 0:
        for k in ('image', 'label', 'pixel_weight'):
            if k in sample:
                sample[k] = np.pad(sample[k], pad, 'reflect')
    return sample


def tf_random_crop(sample, output_size, fg_focus=False, fg_ratio=0.5, mask_label=(1,), rng=None):
    """crop.py:201-236 (parameters, incl. the order of the random draws) + crop.py:27-49 (the slicing)"""
    import random as _random
    rng = rng or _random
    shape = sample['image'].shape
    dim = len(shape) - 1
    size = list(output_size)
    if dim == 3 and size[0] is None:
        size = [shape[1]] + size[1:]
    margin = [shape[i + 1] - size[i] for i in range(dim)]
    cmin = [0 if m == 0 else rng.randint(0, m) for m in margin]
    if fg_focus and rng.random() < fg_ratio:
        label = sample['label']
        mask = np.zeros_like(label)
        for lab in mask_label:
            mask = np.maximum(mask, label == lab)
        if mask.sum() == 0:
            bb_min, bb_max = [0] * (dim + 1), list(mask.shape)
        else:
            idx = np.nonzero(mask)                          # util/image_process.py:8-34
            bb_min = [int(idx[i].min()) for i in range(dim + 1)]
            bb_max = [int(idx[i].max()) + 1 for i in range(dim + 1)]
        bb_min, bb_max = bb_min[1:], bb_max[1:]
        cmin = [rng.randint(bb_min[i], bb_max[i]) - int(size[i] / 2) for i in range(dim)]
        cmin = [max(0, c) for c in cmin]
        cmin = [min(cmin[i], shape[i + 1] - size[i]) for i in range(dim)]
    cmax = [cmin[i] + size[i] for i in range(dim)]
    sample['RandomCrop_Param'] = (list(shape), [0] + cmin, [shape[0]] + cmax)
    sl = tuple(slice(cmin[i], cmax[i]) for i in range(dim))
    for k in ('image', 'label', 'pixel_weight'):
        if k in sample:
            sample[k] = sample[k][(slice(None),) + sl]
    return sample


def tf_random_flip(sample, flip_depth, flip_height, flip_width, rng=None):
    """flip.py:34-62: one draw per enabled axis, in the order width, height, depth"""
    import random as _random
    rng = rng or _random
    dim = sample['image'].ndim - 1
    axes = []
    if flip_width and rng.random() > 0.5:
        axes.append(-1)
    if flip_height and rng.random() > 0.5:
        axes.append(-2)
    if dim == 3 and flip_depth and rng.random() > 0.5:
        axes.append(-3)
    sample['RandomFlip_Param'] = axes
    if axes:
        for k in ('image', 'label', 'pixel_weight'):
            if k in sample:
                sample[k] = np.flip(sample[k], axes).copy()
    return sample


def tf_label_to_probability(sample, class_num):
    """label_convert.py:82-94 (segmentation)"""
    label = sample['label'][0]
    prob = np.zeros((class_num,) + label.shape, dtype=np.float32)
    for i in range(class_num):
        prob[i] = label == i * np.ones_like(label)
    sample['label_prob'] = prob
    return sample


def tf_train_chain(sample, p, rng=None):
    """the five transforms with the reference's parameter keys (lower-cased, as parse_config delivers them)"""
    s = tf_normalize(sample, p.get('normalizewithmeanstd_channels'), p.get('normalizewithmeanstd_mean'),
                     p.get('normalizewithmeanstd_std'))
    s = tf_pad(s, p['pad_output_size'], p.get('pad_ceil_mode', False))
    s = tf_random_crop(s, p['randomcrop_output_size'], p.get('randomcrop_foreground_focus', False),
                       p.get('randomcrop_foreground_ratio', 0.5), p.get('randomcrop_mask_label', [1]), rng)
    s = tf_random_flip(s, p['randomflip_flip_depth'], p['randomflip_flip_height'], p['randomflip_flip_width'], rng)
    return tf_label_to_probability(s, p['labeltoprobability_class_num'])


# ---------------------------------------------------------------------------------------------------------------------
# evaluation (PyMIC/pymic/util/evaluation_seg_train.py) - SURVEY 8f #3
def ev_binary_dice(s, g):
    """evaluation_seg_train.py:21-50"""
    prod = np.multiply(s, g)
    return (2.0 * prod.sum() + 1e-5) / (s.sum() + g.sum() + 1e-5)


def ev_binary_iou(s, g):
    """evaluation_seg_train.py:68-81 in float64 (NumPy 1.x promotion, which the reference was written for)"""
    inter = np.multiply(s, g)
    union = np.asarray(s + g > 0, np.float64)
    return (inter.sum() + 1e-5) / (union.sum() + 1e-5)


def ev_binary_rve(s, g):
    """evaluation_seg_train.py:171-186"""
    s_v, g_v = float(s.sum()), float(g.sum())
    assert g_v > 0
    return abs(s_v - g_v) / g_v


def ev_multi_class(s_volume, g_volume, label_list, fuse_label, spacing, metric):
    """evaluation_seg_train.py:188-262"""
    if fuse_label:
        s_sub, g_sub = np.zeros_like(s_volume), np.zeros_like(g_volume)
        for lab in label_list:
            s_sub = s_sub + np.asarray(s_volume == lab, np.uint8)
            g_sub = g_sub + np.asarray(g_volume == lab, np.uint8)
        label_list = [1]
        s_volume, g_volume = np.asarray(s_sub > 0, np.uint8), np.asarray(g_sub > 0, np.uint8)
    out = []
    for label in label_list:
        s, g = s_volume == label, g_volume == label
        m = metric.lower()
        if m == "dice":
            out.append(ev_binary_dice(s, g))
        elif m == "iou":
            out.append(ev_binary_iou(s, g))
        elif m == "rve":
            out.append(ev_binary_rve(s, g))
        elif m == "volume":
            voxel_size = 1.0
            for dim in range(len(spacing)):
                voxel_size = voxel_size * spacing[dim]
            out.append(g.sum() * voxel_size)
        elif m == "assd":
            out.append(ev_binary_assd(_ev_squeeze(s), _ev_squeeze(g), spacing))
        elif m == "hd95":
            out.append(ev_binary_hd95(_ev_squeeze(s), _ev_squeeze(g), spacing))
        else:
            raise ValueError("unsupported evaluation metric: {0:}".format(metric))
    return out


def _ev_squeeze(v):
    """get_binary_evaluation_score, evaluation_seg_train.py:203-209"""
    if v.ndim == 4:
        assert v.shape[0] == 1
        v = v.reshape(v.shape[1:])
    if v.shape[0] == 1:
        v = v.reshape(v.shape[1:])
    return v


def ev_edge_points(img):
    """evaluation_seg_train.py:84-99 (scipy's binary_erosion with the cross structuring element, border_value 0)"""
    from scipy import ndimage
    strt = ndimage.generate_binary_structure(img.ndim, 1)
    ero = ndimage.binary_erosion(img, strt)
    return np.asarray(img, np.uint8) - np.asarray(ero, np.uint8)


def ev_raster_scan(seeds, spacing, iterations=2):
    """What the reference calls as GeodisTK.geodesic3d_raster_scan(zeros, seeds, spacing, 0.0, iterations)
    (geodesic2d_raster_scan for 2D, unit spacing; evaluation_seg_train.py:121-126, 157-162).

    PARITY UNPINNED: GeodisTK (github.com/taigw/GeodisTK, a C++ extension; the reference imports it without pinning a
    version) is not in /root/reference and not installed, so this restates the published algorithm it implements -
    the raster-scan geodesic distance transform of Toivanen (1996) / Criminisi et al. (2008): distances start at 0 on
    the seeds and 1e10 elsewhere; a forward sweep relaxes every voxel against its already-visited half of the 26
    (2D: 8) neighbourhood, a backward sweep against the other half; the step cost is sqrt(l_euc^2 + lambda^2 * dI^2),
    here with lambda = 0 and a constant image simply the Euclidean step length under `spacing`.  Pure-Python loops:
    small volumes only.  float32 like GeodisTK's buffers."""
    seeds = np.asarray(seeds)
    if seeds.ndim == 2:
        seeds, spacing = seeds[None], (1.0, 1.0, 1.0)
        return ev_raster_scan(seeds, spacing, iterations)[0]
    D, H, W = seeds.shape
    dis = np.where(seeds > 0, np.float32(0.0), np.float32(1.0e10)).astype(np.float32)
    fwd = [(dd, dh, dw) for dd in (-1, 0) for dh in (-1, 0, 1) for dw in (-1, 0, 1)
           if (dd, dh, dw) < (0, 0, 0)]                        # the 13 neighbours a forward raster has already visited
    cost = {o: np.float32(np.sqrt((o[0] * spacing[0]) ** 2 + (o[1] * spacing[1]) ** 2 + (o[2] * spacing[2]) ** 2))
            for o in fwd}
    for _ in range(iterations):
        for sign in (1, -1):
            zs = range(D) if sign > 0 else range(D - 1, -1, -1)
            ys = range(H) if sign > 0 else range(H - 1, -1, -1)
            xs = range(W) if sign > 0 else range(W - 1, -1, -1)
            for z in zs:
                for y in ys:
                    for x in xs:
                        best = dis[z, y, x]
                        for o in fwd:
                            zz, yy, xx = z + sign * o[0], y + sign * o[1], x + sign * o[2]
                            if 0 <= zz < D and 0 <= yy < H and 0 <= xx < W:
                                v = np.float32(dis[zz, yy, xx] + cost[o])
                                if v < best:
                                    best = v
                        dis[z, y, x] = best
    return dis


def ev_binary_assd(s, g, spacing=None):
    """evaluation_seg_train.py:137-171"""
    s_edge, g_edge = ev_edge_points(s), ev_edge_points(g)
    dim = s.ndim
    assert dim == g.ndim
    if spacing is None:
        spacing = [1.0] * dim
    s_dis, g_dis = ev_raster_scan(s_edge, spacing, 2), ev_raster_scan(g_edge, spacing, 2)
    ns, ng = s_edge.sum(), g_edge.sum()
    with np.errstate(invalid="ignore", divide="ignore"):
        assd = ((s_dis * g_edge).sum() + (g_dis * s_edge).sum()) / (ns + ng)
    return 50 if assd > 50 else assd


def ev_binary_hd95(s, g, spacing=None):
    """evaluation_seg_train.py:101-135"""
    s_edge, g_edge = ev_edge_points(s), ev_edge_points(g)
    if spacing is None:
        spacing = [1.0] * s.ndim
    s_dis, g_dis = ev_raster_scan(s_edge, spacing, 2), ev_raster_scan(g_edge, spacing, 2)
    l1, l2 = sorted(s_dis[g_edge > 0]), sorted(g_dis[s_edge > 0])
    return max(l1[int(len(l1) * 0.95)], l2[int(len(l2) * 0.95)])

"""Training-sample transforms on the GPU, behind PyMIC's transform interface.

Mirrors the classes FPL+'s configs name (config_dual/data_vs/vs_t1s_g.cfg:21-23:
train_transform = [NormalizeWithMeanStd, Pad, RandomCrop, RandomFlip, LabelToProbability]):
same class names, same lower-cased parameter keys (PyMIC/pymic/transform/*.py), same `__call__(sample) -> sample`
contract and the same `<Name>_Param` json strings in the sample, so `TransformDict[name](params)` drops in for
PyMIC/pymic/transform/trans_dict.py:42.  The difference is where the volumes live: `sample['image']` (float32
[C,D,H,W]), `sample['label']` (uint8 [1,D,H,W]) and `sample['pixel_weight']` (float32 [1,D,H,W]) are device tensors
and every gather / reduction is a HIP kernel (csrc/sample.hip).  The random decisions are drawn on the host from
Python's `random` in exactly the reference's order, so a seeded run picks the same crops and flips as the reference.
"""
import json
import math
import random

import numpy as np

import torch

from . import ops

_SPATIAL_KEYS = ("label", "pixel_weight", "image1")


class AbstractTransform(object):
    """PyMIC/pymic/transform/abstract_transform.py:4-27"""

    def __init__(self, params):
        self.task = params['task']

    def __call__(self, sample):
        return sample

    def inverse_transform_for_prediction(self, sample):
        raise ValueError("not implemented")

    def _others(self, sample):
        if self.task != 'segmentation':
            return []
        return [k for k in _SPATIAL_KEYS if k in sample]


def _check_volume(t, name):
    if not (torch.is_tensor(t) and t.is_cuda and t.dim() == 4):
        raise ValueError("fplx.transform: sample['{0:}'] must be a [C,D,H,W] device tensor".format(name))
    return t.contiguous()


class NormalizeWithMeanStd(AbstractTransform):
    """normalize.py:34-68.  mean/std None -> each channel's own float32 mean and population std."""

    def __init__(self, params):
        super(NormalizeWithMeanStd, self).__init__(params)
        self.chns = params['normalizewithmeanstd_channels']
        self.mean = params.get('normalizewithmeanstd_mean', None)
        self.std = params.get('normalizewithmeanstd_std', None)
        self.ignore_np = params.get('normalizewithmeanstd_ignore_non_positive', False)
        self.inverse = params.get('normalizewithmeanstd_inverse', False)

    def __call__(self, sample):
        image = _check_volume(sample['image'], 'image')
        chns = self.chns if self.chns is not None else range(image.shape[0])
        if self.mean is None:
            self.mean = [None] * len(chns)
            self.std = [None] * len(chns)
        for i, chn in enumerate(chns):
            ms = None if self.mean[i] is None else (self.mean[i], self.std[i])
            if self.ignore_np:
                # normalize.py:55-66: moments over the positive voxels (only when none are given), the others replaced by
                # numpy.random.normal(0, 1) - drawn on the host from numpy's global generator for the WHOLE channel, as the
                # reference does, so that a seeded run reproduces its numbers
                noise = torch.from_numpy(np.random.normal(0, 1, size=tuple(image[chn].shape)).astype(np.float32))
                noise = noise.to(image.device)
                if ms is None:
                    ops.normalize_positive(image[chn], noise, out=image[chn])
                else:
                    keep = ~(image[chn] <= 0)
                    ops.normalize_mean_std(image[chn], ms, out=image[chn])
                    image[chn] = torch.where(keep, image[chn], noise)
                continue
            ops.normalize_mean_std(image[chn], ms, out=image[chn])      # in place, like the reference
        sample['image'] = image
        return sample


class Pad(AbstractTransform):
    """pad.py:117-191: reflect padding to max(image_size, output_size), lower margin int(margin / 2)."""

    def __init__(self, params):
        super(Pad, self).__init__(params)
        self.output_size = params['pad_output_size']
        self.ceil_mode = params.get('pad_ceil_mode', False)
        self.inverse = params.get('pad_inverse', True)

    def __call__(self, sample):
        image = _check_volume(sample['image'], 'image')
        shape = image.shape
        assert len(self.output_size) == 3
        if self.ceil_mode:
            out = [int(math.ceil(float(shape[1 + i]) / self.output_size[i])) * self.output_size[i] for i in range(3)]
        else:
            out = self.output_size
        margin = [max(0, out[i] - shape[1 + i]) for i in range(3)]
        lower = [int(margin[i] / 2) for i in range(3)]
        upper = [margin[i] - lower[i] for i in range(3)]
        sample['Pad_Param'] = json.dumps((lower, upper))
        if max(margin) == 0:
            return sample
        size = [shape[1 + i] + margin[i] for i in range(3)]
        sample['image'] = ops.pad_reflect(image, lower, size)
        for k in self._others(sample):
            sample[k] = ops.pad_reflect(_check_volume(sample[k], k), lower, size)
        return sample

    def inverse_transform_for_prediction(self, sample):
        p = sample['Pad_Param']
        lower, upper = json.loads(p[0] if isinstance(p, (list, tuple)) else p)

        def crop(pred):                                  # [N,C,D,H,W]
            n, c = pred.shape[:2]
            size = [pred.shape[2 + i] - lower[i] - upper[i] for i in range(3)]
            flat = pred.contiguous().view(n * c, *pred.shape[2:])
            return ops.crop_flip(flat, lower, size).view(n, c, *size)

        predict = sample['predict']
        sample['predict'] = [crop(q) for q in predict] if isinstance(predict, (tuple, list)) else crop(predict)
        return sample


class RandomCrop(AbstractTransform):
    """crop.py:165-245.  Draw order: one randint per axis with a margin, then random() for the foreground focus,
    then one randint per axis inside the label's bounding box."""

    def __init__(self, params):
        self.output_size = params['randomcrop_output_size']
        self.fg_focus = params.get('randomcrop_foreground_focus', False)
        self.fg_ratio = params.get('randomcrop_foreground_ratio', 0.5)
        self.mask_label = params.get('randomcrop_mask_label', [1])
        self.inverse = params.get('randomcrop_inverse', True)
        self.task = params['task']
        assert isinstance(self.output_size, (list, tuple))
        if self.mask_label is not None:
            assert isinstance(self.mask_label, (list, tuple))

    def _get_crop_param(self, sample):
        shape = list(sample['image'].shape)
        assert len(self.output_size) == 3
        size = list(self.output_size)
        if size[0] is None:
            size[0] = shape[1]
        margin = [shape[i + 1] - size[i] for i in range(3)]
        crop_min = [0 if m == 0 else random.randint(0, m) for m in margin]
        if self.fg_focus and random.random() < self.fg_ratio:
            count, bb_min, bb_max = ops.label_bbox(_check_volume(sample['label'], 'label'), self.mask_label)
            if count == 0:
                bb_min, bb_max = [0] * 4, list(sample['label'].shape)
            bb_min, bb_max = bb_min[1:], bb_max[1:]
            crop_min = [random.randint(bb_min[i], bb_max[i]) - int(size[i] / 2) for i in range(3)]
            crop_min = [max(0, v) for v in crop_min]
            crop_min = [min(crop_min[i], shape[i + 1] - size[i]) for i in range(3)]
        crop_max = [crop_min[i] + size[i] for i in range(3)]
        sample['RandomCrop_Param'] = json.dumps((shape, [0] + crop_min, shape[0:1] + crop_max))
        return sample, crop_min, size

    def __call__(self, sample):
        image = _check_volume(sample['image'], 'image')
        sample, crop_min, size = self._get_crop_param(sample)
        sample['image'] = ops.crop_flip(image, crop_min, size)
        for k in self._others(sample):
            sample[k] = ops.crop_flip(_check_volume(sample[k], k), crop_min, size)
        return sample


class RandomFlip(AbstractTransform):
    """flip.py:14-62: one draw per enabled axis in the order width, height, depth; flip when the draw is > 0.5."""

    def __init__(self, params):
        super(RandomFlip, self).__init__(params)
        self.flip_depth = params['randomflip_flip_depth']
        self.flip_height = params['randomflip_flip_height']
        self.flip_width = params['randomflip_flip_width']
        self.inverse = params.get('randomflip_inverse', True)

    def __call__(self, sample):
        image = _check_volume(sample['image'], 'image')
        flip_axis = []
        if self.flip_width and random.random() > 0.5:
            flip_axis.append(-1)
        if self.flip_height and random.random() > 0.5:
            flip_axis.append(-2)
        if self.flip_depth and random.random() > 0.5:
            flip_axis.append(-3)
        sample['RandomFlip_Param'] = json.dumps(flip_axis)
        if flip_axis:
            mask = sum(1 << (-a - 1) for a in flip_axis)
            sample['image'] = ops.crop_flip(image, (0, 0, 0), image.shape[1:], mask)
            for k in self._others(sample):
                t = _check_volume(sample[k], k)
                sample[k] = ops.crop_flip(t, (0, 0, 0), t.shape[1:], mask)
        return sample

    def inverse_transform_for_prediction(self, sample):
        p = sample['RandomFlip_Param']
        flip_axis = json.loads(p[0] if isinstance(p, (list, tuple)) else p)
        if flip_axis:
            mask = sum(1 << (-a - 1) for a in flip_axis)
            pred = sample['predict']
            n, c = pred.shape[:2]
            flat = pred.contiguous().view(n * c, *pred.shape[2:])
            sample['predict'] = ops.crop_flip(flat, (0, 0, 0), pred.shape[2:], mask).view_as(pred)
        return sample


class LabelToProbability(AbstractTransform):
    """label_convert.py:64-101 (segmentation): one-hot fp32 [class_num, D, H, W] in sample['label_prob']."""

    def __init__(self, params):
        super(LabelToProbability, self).__init__(params)
        self.class_num = params['labeltoprobability_class_num']
        self.inverse = params.get('labeltoprobability_inverse', False)

    def __call__(self, sample):
        if self.task != 'segmentation':
            raise ValueError("fplx.transform: LabelToProbability supports the segmentation task only")
        label = _check_volume(sample['label'], 'label')
        if label.dtype != torch.uint8:
            raise ValueError("fplx.transform: sample['label'] must be uint8")
        sample['label_prob'] = ops.label_to_probability(label[0], self.class_num)
        return sample


TransformDict = {
    'NormalizeWithMeanStd': NormalizeWithMeanStd,
    'Pad': Pad,
    'RandomCrop': RandomCrop,
    'RandomFlip': RandomFlip,
    'LabelToProbability': LabelToProbability,
}


def build_transforms(names, params):
    """The list PyMIC/pymic/net_run/agent_seg.py:48-61 builds from `<stage>_transform`: unknown names raise."""
    out = []
    for name in names:
        if name not in TransformDict:
            raise ValueError("Undefined transform {0:}".format(name))
        out.append(TransformDict[name](params))
    return out


def apply_transforms(transforms, sample):
    for t in transforms:
        sample = t(sample)
    return sample


class Compose(object):
    """torchvision.transforms.Compose as the reference uses it (agent_seg.py:61): a callable chain"""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, sample):
        return apply_transforms(self.transforms, sample)

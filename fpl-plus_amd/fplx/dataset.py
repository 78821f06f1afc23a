"""NiftyDataset on device volumes: the caller side of the transforms.

Mirrors PyMIC/pymic/io/nifty_dataset.py:111-237 (`NiftyDataset`): same constructor, same csv columns
(`image..., label, pixel_weight, image_weight`, nifty_dataset.py:133-141), same sample keys, the same `set_weight_`
rule (165-168) and the same 0.5 fall-back weight when a pixel-weight file cannot be read (216-220).  The files are
parsed on the host (fplx/nifti.py) and uploaded once; from there the sample stays in HBM: image float32 [C,D,H,W],
label uint8 [1,D,H,W] (the reference keeps int32; class indices fit a byte and the loss kernels take bytes),
pixel_weight float32 [1,D,H,W].  `collate` is torch's default collate for these dictionaries (stack tensors, list strings).

MI355X-first: with `cache=True` every case is parsed and uploaded ONCE and the whole training set stays resident in HBM
(the VS set is ~200 volumes x 5 MB against 288 GB); an iteration then costs a handful of small kernels and no host IO,
which is what keeps the train step fed without the reference's 16 DataLoader worker processes.
`BatchLoader` is the DataLoader of agent_abstract.py:269-281 for such a dataset: shuffled index order from a seeded
torch.Generator, batches of `batch_size` collated samples, the last short batch kept (drop_last = False).
"""
import numpy as np
import pandas as pd
import torch

from . import ops
from .nifti import load_image_as_nd_array


class NiftyDataset(object):
    def __init__(self, root_dir, csv_file, modal_num=1, with_label=False, transform=None, device="cuda:0", cache=False):
        self.cache = {} if cache else None
        self.root_dir = root_dir
        self.csv_items = pd.read_csv(csv_file)
        self.modal_num = modal_num
        self.with_label = with_label
        self.transform = transform
        self.device = torch.device(device)
        keys = list(self.csv_items.keys())
        self.image_weight_idx = keys.index('image_weight') if 'image_weight' in keys else None
        self.pixel_weight_idx = keys.index('pixel_weight') if 'pixel_weight' in keys else None
        self.image1 = keys.index('image1') if 'image1' in keys else None

    def __len__(self):
        return len(self.csv_items)

    def _path(self, idx, col):
        return "{0:}/{1:}".format(self.root_dir, self.csv_items.iloc[idx, col])

    def _upload(self, a, dtype):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype))).to(self.device)

    def __getlabel__(self, idx):
        col = list(self.csv_items.keys()).index('label')
        label = load_image_as_nd_array(self._path(idx, col))['data_array']
        if label.min() < 0 or label.max() > 255:
            raise ValueError("fplx.NiftyDataset: label values outside [0, 255] in {0:}".format(self._path(idx, col)))
        return self._upload(label, np.uint8)

    def set_weight_(self, img_weight, pixel_weight):
        return ops.set_weight_(pixel_weight, img_weight)

    def __getitem__(self, idx):
        if self.cache is None:
            sample = self._load(idx)
        else:
            if idx not in self.cache:
                self.cache[idx] = self._load(idx)
            # the transforms replace entries of the dictionary and normalise the image in place: hand out a copy
            sample = dict(self.cache[idx])
            sample['image'] = sample['image'].clone()
        if self.transform:
            sample = self.transform(sample)
        return sample

    def _load(self, idx):
        names_list, image_list = [], []
        for i in range(self.modal_num):
            image_name = self.csv_items.iloc[idx, i]
            image_dict = load_image_as_nd_array("{0:}/{1:}".format(self.root_dir, image_name))
            names_list.append(image_name)
            image_list.append(image_dict['data_array'])
        image = self._upload(np.concatenate(image_list, axis=0), np.float32)
        sample = {'image': image, 'names': names_list[0], 'origin': image_dict['origin'],
                  'spacing': image_dict['spacing'], 'direction': image_dict['direction']}
        if self.with_label:
            sample['label'] = self.__getlabel__(idx)
            assert image.shape[1:] == sample['label'].shape[1:]
        if self.image_weight_idx is not None:
            sample['image_weight'] = float(self.csv_items.iloc[idx, self.image_weight_idx])
            if self.pixel_weight_idx is None:
                sample['pixel_weight'] = self.set_weight_(sample['image_weight'], torch.ones_like(image))
        if self.pixel_weight_idx is not None:
            try:
                w = load_image_as_nd_array(self._path(idx, self.pixel_weight_idx))['data_array']
                sample['pixel_weight'] = self.set_weight_(sample['image_weight'], self._upload(w, np.float32))
            except (OSError, ValueError, KeyError):
                sample['pixel_weight'] = torch.full_like(image, 0.5)
            assert image.shape[1:] == sample['pixel_weight'].shape[1:]
        if self.image1 is not None:
            try:
                w = load_image_as_nd_array(self._path(idx, self.image1))['data_array']
                sample['image1'] = self._upload(w, np.float32)
            except (OSError, ValueError):
                sample['image1'] = image
        return sample


def collate(samples):
    """default-collate of sample dictionaries: tensors stacked along a new batch axis, numbers to tensors, strings listed"""
    out = {}
    for k in samples[0]:
        v = [s[k] for s in samples]
        if torch.is_tensor(v[0]):
            out[k] = torch.stack(v, 0)
        elif isinstance(v[0], (int, float)):
            out[k] = torch.tensor(v, dtype=torch.float64 if isinstance(v[0], float) else torch.int64)
        else:
            out[k] = v
    return out


class BatchLoader(object):
    """iterable of collated batches over a dataset (agent_abstract.py:263-281: batch_size, shuffle, seeded generator).
    rank / world: data parallelism with one process per GPU - `batch_size` stays the GLOBAL batch of the config and every
    rank collates its chunk of each batch, chunk r of `world` along the batch axis: what nn.DataParallel's scatter hands to
    replica r (agent_seg.py:692-698).  All ranks draw the same permutation (same seeded generator)."""

    def __init__(self, dataset, batch_size=1, shuffle=False, generator=None, rank=0, world=1):
        self.dataset, self.batch_size, self.shuffle, self.generator = dataset, int(batch_size), shuffle, generator
        self.rank, self.world = int(rank), int(world)
        if self.world > 1 and self.batch_size % self.world:
            raise ValueError("fplx BatchLoader: batch size {0:} is not a multiple of the {1:} ranks".format(batch_size, world))

    def __len__(self):
        if self.world > 1:                               # the ragged last batch is dropped on all ranks (__iter__)
            return len(self.dataset) // self.batch_size
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        per = self.batch_size // self.world
        for i in range(0, n, self.batch_size):
            full = order[i:i + self.batch_size]
            if self.world > 1:
                if len(full) < self.batch_size:          # a ragged last batch cannot be split evenly: dropped on all ranks
                    return
                full = full[self.rank * per:(self.rank + 1) * per]
            yield collate([self.dataset[j] for j in full])

"""Inferer - the reference's plugin class (PyMIC/pymic/net_run_dsbn/infer_func.py:7-222) re-designed for the GPU: same
constructor config (`sliding_window_enable`, `sliding_window_size`, `sliding_window_stride`, `tta_mode`, `class_num`), same
`run(model, image, domain_label)`, same result - but ONE plan instead of nested loops of forwards:

  plan      per-axis tile starts min(k * stride, size - window), flips (none | H | W | H+W for tta_mode 1);
  extract   every tile of every flip (and of every Monte-Carlo pass, `run_mc`) gathered into one batch on the device
            (fplx_sw_extract);
  forward   the network runs on chunks of that batch (eval-mode networks: BatchNorm uses running statistics, samples are
            independent; a network in train mode keeps the reference's one-forward-per-tile batches - and run_mc its
            pass-by-pass loop - because its batch statistics would change);
  merge     every output voxel is formed by the reference's additions in the reference's order (tile order w, h, d; flips
            ((o1 + o2) + o3 + o4) / 4) by a gather kernel (fplx_sw_merge) - independent of the chunking.

Networks that return several tensors (deep supervision) take the generic path at the end of this file, which follows
infer_func.py:113-140 literally, including its counter (incremented once per output and tile, lines 132-137).
"""
import ctypes

import torch

from . import ops
from ._lib import call

_FLIPS_TTA = (0, 2, 1, 3)          # infer_func.py:201-204: image, flip [-2] (H), flip [-1] (W), flip [-2, -1]


def _axis_starts(size, window, stride):
    """infer_func.py:75-84 for one axis: range(0, size, stride) clamped to size - window (duplicates kept)"""
    return [min(s, size - window) for s in range(0, size, stride)]


class Inferer(object):
    def __init__(self, config):
        self.config = config
        self.model = None
        # samples per network forward when tiles are batched: bounded by voxels so that activations stay in the GBs
        self.max_batch_voxels = int(config.get('infer_batch_voxels', 1 << 23))

    # ------------------------------------------------------------------ plan
    def _plan(self, image):
        img_shape = list(image.shape[2:])
        if len(img_shape) != 3:
            raise ValueError("Inference using sliding window only supports 2D and 3D images")   # infer_func.py:63-64
        if not self.config.get('sliding_window_enable', False):
            window, stride = list(img_shape), list(img_shape)
        else:
            window = [x for x in self.config['sliding_window_size']]
            stride = [x for x in self.config['sliding_window_stride']]
            for d in range(3):                                               # infer_func.py:66-70
                if window[d] is None or window[d] > img_shape[d]:
                    window[d] = img_shape[d]
                if stride[d] is None or stride[d] > window[d]:
                    stride[d] = window[d]
        if all(window[d] >= img_shape[d] for d in range(3)):                 # infer_func.py:72-74: one plain forward
            starts = [[0], [0], [0]]
        else:
            starts = [_axis_starts(img_shape[d], window[d], stride[d]) for d in range(3)]
        tta_mode = self.config.get('tta_mode', 0)
        if tta_mode == 0:
            flips = (0,)
        elif tta_mode == 1:
            flips = _FLIPS_TTA
        else:
            raise ValueError("Undefined tta_mode {0:}".format(tta_mode))
        return window, starts, flips

    @staticmethod
    def _c_args(image_shape, window, starts, flips):
        n, c, d, h, w = image_shape
        arr = [(ctypes.c_int * len(s))(*s) for s in starts]
        fl = (ctypes.c_int * len(flips))(*flips)
        return (n, c, d, h, w, arr[0], len(starts[0]), arr[1], len(starts[1]), arr[2], len(starts[2]),
                window[0], window[1], window[2], fl, len(flips))

    # ------------------------------------------------------------------ batched path (single-output networks)
    def _forward_chunks(self, patches, n, domain_label, passes):
        """patches [nb, C, wd, wh, ww] with nb = flips * tiles * n -> (logits, chunk): the network runs on chunks of `chunk`
        consecutive patches and each call leaves all passes of its chunk, pass-major, in its slice of the one buffer
        [chunks][passes][patches of the chunk][classes, wd, wh, ww] - the layout fplx_sw_merge_mc reads.
        Monte-Carlo passes see the same patches; their dropout masks differ per sample of the batch.  A network that offers
        forward_mc (fplx.UNet2D5_dsbn in eval mode) computes the part above its first active dropout once for all passes
        and writes its logits straight into the slice."""
        model = self.model
        nb = patches.shape[0]
        per = patches[0, 0].numel()
        if getattr(model, 'training', False):
            chunk = n                                     # train-mode BatchNorm: the reference's batches, one tile each
        else:
            chunk = max(n, (self.max_batch_voxels // max(per * passes, 1)) // n * n)
        chunk = min(chunk, nb)
        dom = int(domain_label[0]) if domain_label is not None else 0
        shared = (hasattr(model, "forward_mc") and not getattr(model, 'training', False)
                  and (passes > 1 or not torch.is_grad_enabled()))

        def buffer(classes):
            return torch.empty((passes * nb, classes) + tuple(patches.shape[2:]), dtype=torch.float32, device=patches.device)

        out = buffer(model.n_class) if shared else None
        for b0 in range(0, nb, chunk):
            b1 = min(nb, b0 + chunk)
            m = b1 - b0
            if shared:
                model.forward_mc(patches[b0:b1], torch.full((m,), dom, dtype=torch.long), passes,
                                 out=out[b0 * passes:(b0 + m) * passes])
                continue
            xin = patches[b0:b1] if passes == 1 else patches[b0:b1].repeat(passes, 1, 1, 1, 1)
            o = model(xin, domain_label=torch.full((m * passes,), dom, dtype=torch.long))
            if isinstance(o, (tuple, list)):
                return None, chunk                        # several outputs: the generic path handles it
            if out is None:
                out = buffer(o.shape[1])
            out[b0 * passes:(b0 + m) * passes] = o
        return out, chunk

    def _run_batched(self, image, domain_label, passes):
        window, starts, flips = self._plan(image)
        ops.require_gpu(image)
        image = image.float().contiguous()
        n = image.shape[0]
        tiles = len(starts[0]) * len(starts[1]) * len(starts[2])
        nb = len(flips) * tiles * n
        cargs = self._c_args(tuple(image.shape), window, starts, flips)
        one = torch.empty((nb, image.shape[1]) + tuple(window), dtype=torch.float32, device=image.device)
        call("fplx_sw_extract", ops.ptr(image), *cargs, ops.ptr(one), ops.stream())
        logits, chunk = self._forward_chunks(one, n, domain_label, passes)
        if logits is None:
            return None
        classes = logits.shape[1]
        outs = torch.empty((passes, n, classes) + tuple(image.shape[2:]), dtype=torch.float32, device=image.device)
        margs = self._c_args((n, classes) + tuple(image.shape[2:]), window, starts, flips)
        call("fplx_sw_merge_mc", ops.ptr(logits), passes, chunk, *margs, ops.ptr(outs), ops.stream())
        return outs

    # ------------------------------------------------------------------ public surface
    def run(self, model, image, domain_label):
        """infer_func.py:188-222"""
        self.model = model
        self._plan(image)                                 # raises for an undefined tta_mode / unsupported rank first
        if image.is_cuda:
            outs = self._run_batched(image, domain_label, 1)
            if outs is not None:
                return outs[0]
        return self._run_generic(image, domain_label)

    def run_mc(self, model, image, domain_label, passes):
        """`passes` stochastic runs of `run` (test-time dropout: agent_seg.py:898-909 calls run() six times) in one batch.
        -> [passes, N, classes, D, H, W]"""
        self.model = model
        if getattr(model, 'training', False) and passes > 1:
            # train-mode BatchNorm (testing.evaluation_mode = False): batch statistics - and the running-statistics update -
            # belong to ONE pass's tile batch, so the passes run one after the other exactly as the reference's loop does
            return torch.stack([self.run(model, image, domain_label) for _ in range(passes)])
        outs = self._run_batched(image, domain_label, passes) if image.is_cuda else None
        if outs is None:
            outs = torch.stack([self._run_generic(image, domain_label) for _ in range(passes)])
        return outs

    # ------------------------------------------------------------------ generic path (any model, several outputs)
    def _infer_generic(self, image, domain_label):
        window, starts, _ = self._plan(image)
        model = self.model
        shape = list(image.shape[2:])
        tiles = [(d0, h0, w0) for w0 in starts[2] for h0 in starts[1] for d0 in starts[0]]
        if len(tiles) == 1:
            return model(image, domain_label=domain_label)
        class_num = self.config['class_num']
        n = image.shape[0]
        first = model(torch.ones((n, image.shape[1]) + tuple(window), device=image.device), domain_label=domain_label)
        if not isinstance(first, (tuple, list)):                              # infer_func.py:96-112
            output = torch.zeros([n, class_num] + shape, device=image.device)
            counter = torch.zeros([n, class_num] + shape, device=image.device)
            for (a, b, c) in tiles:
                sl = (slice(None), slice(None), slice(a, a + window[0]), slice(b, b + window[1]), slice(c, c + window[2]))
                o = model(image[sl], domain_label=domain_label)
                output[sl] += o[0] if isinstance(o, (tuple, list)) else o
                counter[sl] += 1.0
            return output / counter
        # several outputs at different scales (infer_func.py:113-140); the shared counter is incremented once per output
        from torch.nn.functional import interpolate
        s0 = list(first[0].shape[2:])
        scales = [[(list(f.shape[2:])[d] + 0.0) / s0[d] for d in range(3)] for f in first]
        scales[0] = [1.0, 1.0, 1.0]
        outs = [torch.zeros([n, class_num] + [int(shape[d] * sc[d]) for d in range(3)], device=image.device) for sc in scales]
        counter = torch.zeros([n, class_num] + shape, device=image.device)
        for (a, b, c) in tiles:
            c0, c1 = (a, b, c), (a + window[0], b + window[1], c + window[2])
            o = model(image[:, :, c0[0]:c1[0], c0[1]:c1[1], c0[2]:c1[2]], domain_label=domain_label)
            for i, sc in enumerate(scales):
                lo = [int(c0[d] * sc[d]) for d in range(3)]
                hi = [int(c1[d] * sc[d]) for d in range(3)]
                outs[i][:, :, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] += o[i]
                counter[:, :, c0[0]:c1[0], c0[1]:c1[1], c0[2]:c1[2]] += 1.0
        for i, sc in enumerate(scales):
            outs[i] = outs[i] / interpolate(counter, scale_factor=sc)
        return outs

    def _run_generic(self, image, domain_label):
        _, _, flips = self._plan(image)
        if len(flips) == 1:
            return self._infer_generic(image, domain_label)
        axes = {0: None, 2: [-2], 1: [-1], 3: [-2, -1]}
        res = []
        for f in flips:
            x = image if axes[f] is None else torch.flip(image, axes[f])
            o = self._infer_generic(x, domain_label)
            if isinstance(o, (tuple, list)):
                o = [t if axes[f] is None else torch.flip(t, axes[f]) for t in o]
            elif axes[f] is not None:
                o = torch.flip(o, axes[f])
            res.append(o)
        if isinstance(res[0], (tuple, list)):
            return [(res[0][i] + res[1][i] + res[2][i] + res[3][i]) / 4 for i in range(len(res[0]))]
        return (res[0] + res[1] + res[2] + res[3]) / 4

"""Segmentation losses behind the reference's SegLossDict surface
(reference: PyMIC/pymic/loss/loss_dict_seg.py:31-41; classes loss/seg/dice.py:9-57 DiceLoss,
dice.py:95-128 DiceLoss_weight, loss/seg/ce.py:9-44 CrossEntropyLoss, loss/seg/combined.py:8-39
CombinedLoss; base class loss/seg/abstract.py:7-37).

Same construction (`params` dict, `loss_softmax` key) and same call: forward(loss_input_dict) with
keys 'prediction', 'ground_truth', optional 'pixel_weight' [N,1,D,H,W] and 'image_weight' [N];
returns a scalar tensor that supports .backward().  One fused HIP pass evaluates softmax, every
requested term and the hard-Dice train metric; there is no CPU path.
"""
import torch
import torch.nn as nn

from . import ops


class _FusedSegLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, label, pw, iw, weights, softmax, holder):
        ops.require_gpu(logits, label, pw, iw)
        if logits.dim() != 5:
            raise ValueError("{0:}D tensor not supported".format(logits.dim()))        # loss/seg/util.py:46-47
        if logits.shape != label.shape:
            raise ValueError("fplx loss: prediction {0:} and ground_truth {1:} differ in shape".format(
                tuple(logits.shape), tuple(label.shape)))
        logits = logits.float().contiguous()
        label = label.float().contiguous()
        n, c = logits.shape[0], logits.shape[1]
        v = logits[0, 0].numel()
        if pw is not None:
            pw = pw.float().contiguous()
            if pw.numel() != n * v:
                raise ValueError("fplx loss: pixel_weight must be [N,1,D,H,W]")
        if iw is not None:
            iw = iw.float().contiguous()
        dev = logits.device
        part = torch.empty((n, ops.loss_rows(v), ops.loss_k(c)), dtype=torch.float32, device=dev)
        out = torch.empty(4 + c, dtype=torch.float32, device=dev)
        coef = torch.empty(n * c * 2 + 2, dtype=torch.float32, device=dev)
        group = getattr(holder, "dist_group", None) if holder is not None else None
        if holder is not None and getattr(holder, "dist_sync", False):
            ops.seg_loss_fwd_dist(logits, label, pw, iw, weights, softmax, part, out, coef, group)
        else:
            ops.seg_loss_fwd(logits, label, pw, iw, weights, softmax, part, out, coef)
        ctx.save_for_backward(logits, label, pw, coef)
        ctx.weights, ctx.softmax = weights, softmax
        if holder is not None:
            holder.last_out = out
        return out[0]

    @staticmethod
    def backward(ctx, g):
        logits, label, pw, coef = ctx.saved_tensors
        dl = torch.empty_like(logits)
        ops.seg_loss_bwd(logits, label, pw, coef, g.float().contiguous(), ctx.weights, ctx.softmax, dl)
        return dl, None, None, None, None, None, None


class AbstractSegLoss(nn.Module):
    """loss/seg/abstract.py:7-21.  `terms` = weights of (Dice, CE, image-weighted Dice, entropy)."""
    terms = (0.0, 0.0, 0.0, 0.0)
    needs_weights = False

    def __init__(self, params=None):
        super(AbstractSegLoss, self).__init__()
        self.softmax = True if params is None else params.get('loss_softmax', True)
        self.last_out = None      # device tensor [4 + C]: total, dice, ce, entropy, hard class Dice[C]
        # data parallelism (fplx.ddp.attach): evaluate the loss over the FULL batch of all ranks, as the reference's
        # nn.DataParallel does on its gathered logits; the ranks' gradients then add up to the full-batch gradient
        self.dist_sync, self.dist_group = False, None

    def _run(self, loss_input_dict, terms):
        predict = loss_input_dict['prediction']
        if isinstance(predict, (list, tuple)):
            predict = predict[0]                                             # dice.py:26-27
        pw = loss_input_dict.get('pixel_weight', None)
        iw = loss_input_dict.get('image_weight', None)
        if terms[2] != 0.0 and (pw is None or iw is None):
            raise KeyError('pixel_weight')                                   # dice.py:109-110 index the dict
        if terms[2] == 0.0:
            iw = None
        return _FusedSegLoss.apply(predict, loss_input_dict['ground_truth'], pw, iw, tuple(float(t) for t in terms),
                                   bool(self.softmax), self)

    def forward(self, loss_input_dict):
        return self._run(loss_input_dict, self.terms)


class DiceLoss(AbstractSegLoss):
    terms = (1.0, 0.0, 0.0, 0.0)


class CrossEntropyLoss(AbstractSegLoss):
    terms = (0.0, 1.0, 0.0, 0.0)


class DiceLoss_weight(AbstractSegLoss):
    terms = (0.0, 0.0, 1.0, 0.0)
    needs_weights = True


class EntropyTerm(AbstractSegLoss):
    """the regulariser SegmentationAgent.training adds (net_run_dsbn/agent_seg.py:352-354)"""
    terms = (0.0, 0.0, 0.0, 1.0)


SegLossDict = {
    'DiceLoss': DiceLoss,
    'CrossEntropyLoss': CrossEntropyLoss,
    'DiceLoss_weight': DiceLoss_weight,
}


class CombinedLoss(AbstractSegLoss):
    """loss/seg/combined.py:20-39: weighted sum of registered losses - evaluated in ONE pass."""

    def __init__(self, params, loss_dict, extra_entropy=0.0):
        super(CombinedLoss, self).__init__(params)
        loss_names = params['loss_type']
        self.loss_weight = params['loss_weight']
        assert (len(loss_names) == len(self.loss_weight))
        terms = [0.0, 0.0, 0.0, float(extra_entropy)]
        for name, w in zip(loss_names, self.loss_weight):
            if name not in loss_dict:
                raise ValueError("{0:} is not defined, or has not been added to the \
                    loss dictionary".format(name))
            cls = loss_dict[name]
            if not (isinstance(cls, type) and issubclass(cls, AbstractSegLoss)):
                raise ValueError("fplx CombinedLoss fuses fplx losses only; {0:} is foreign".format(name))
            for i, t in enumerate(cls.terms):
                terms[i] += w * t
        self.terms = tuple(terms)


def make_loss(training_cfg, loss_dict=None, entropy_weight=0.0):
    """create_loss_calculator (net_run_dsbn/agent_seg.py:113-132) for the fused losses."""
    loss_dict = SegLossDict if loss_dict is None else loss_dict
    name = training_cfg['loss_type']
    if isinstance(name, (list, tuple)):
        return CombinedLoss(training_cfg, loss_dict, entropy_weight)
    if name not in loss_dict:
        raise ValueError("Undefined loss function {0:}".format(name))          # agent_seg.py:120-121
    if entropy_weight == 0.0:
        return loss_dict[name](training_cfg)
    cfg = dict(training_cfg)
    cfg['loss_type'], cfg['loss_weight'] = [name], [1.0]
    return CombinedLoss(cfg, loss_dict, entropy_weight)

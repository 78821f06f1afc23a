"""DomainSpecificBatchNorm3d - mirror of reference PyMIC/pymic/net_run_dsbn/dsbn.py:35-64.

`bns` is a ModuleList of torch.nn.BatchNorm3d used as parameter / buffer containers (same
state_dict keys: bns.{d}.weight|bias|running_mean|running_var|num_batches_tracked).  Inside
UNet2D5_dsbn the layer is fused into the convolution epilogue + the BN-apply/PReLU pass; a
stand-alone call `layer(x, domain_label)` runs the same HIP kernels (statistics, finalize,
apply) and returns `(y, domain_label)` like the reference.  Forward only (no autograd).
"""
import torch
from torch import nn

from . import ops


class _DomainSpecificBatchNorm3d(nn.Module):
    _version = 2

    def __init__(self, num_features, num_domains):
        super(_DomainSpecificBatchNorm3d, self).__init__()
        self.bns = nn.ModuleList([nn.BatchNorm3d(num_features) for _ in range(num_domains)])

    def reset_running_stats(self):
        for bn in self.bns:
            bn.reset_running_stats()

    def reset_parameters(self):
        for bn in self.bns:
            bn.reset_parameters()

    def _check_input_dim(self, input):
        raise NotImplementedError

    def forward(self, x, domain_label):
        self._check_input_dim(x)
        bn = self.bns[domain_label[0]]            # one BN set for the whole batch (dsbn.py:56)
        ops.require_gpu(x)
        n, c = x.shape[0], x.shape[1]
        if c != bn.num_features:
            raise ValueError("fplx DSBN: expected {0:} channels, got {1:}".format(bn.num_features, c))
        xin = x.detach().float().permute(0, 2, 3, 4, 1).contiguous().view(-1, c)     # NDHWC view (plumbing)
        vox = xin.shape[0]
        bnbuf = torch.empty((4, c), dtype=torch.float32, device=x.device)
        if bn.training:
            rows = ops.num_partials(vox)
            stats = torch.empty((rows, 2, c), dtype=torch.float32, device=x.device)
            ops.call("fplx_channel_stats", ops.ptr(xin), c, vox, c, ops.F32, ops.ptr(stats), ops.stream())
            ops.bn_train_finalize(stats, rows, c, vox, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                  bn.num_batches_tracked, bnbuf, bn.momentum, bn.eps)
        else:
            ops.bn_eval_prepare(bn.weight, bn.bias, bn.running_mean, bn.running_var, bnbuf, bn.eps)
        one = torch.ones(1, dtype=torch.float32, device=x.device)                   # PReLU slope 1 = identity
        out = torch.empty_like(xin)
        ops.bn_act_fwd(xin, out, bnbuf, one, 0.0, 0, 0, c)
        y = out.view(n, x.shape[2], x.shape[3], x.shape[4], c).permute(0, 4, 1, 2, 3)
        return y, domain_label


class DomainSpecificBatchNorm3d(_DomainSpecificBatchNorm3d):
    def _check_input_dim(self, input):
        if input.dim() != 5:
            raise ValueError('expected 5D input (got {}D input)'.format(input.dim()))

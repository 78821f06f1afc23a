"""helpers shared by the GPU parity tests"""
import numpy as np
import torch

import detdata


def load_det_weights(net, params, device):
    sd = detdata.state_dict_3d(params)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net.to(device)
    return sd


def max_rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))

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


def plan_kernel(n, d, h, w, cin, cout, k=(3, 3, 3), x_dt=1, y_dt=1, full=False):
    """fplx_conv3d_plan_query: the kernel family (FPLX_KERNEL_*: 0 generic, 1 direct, 2 tile, 3 stream, 4 march, 5 brick,
    6 stem, 7 out_conv) fplx_conv3d_fwd dispatches a layer to; full=True -> (kernel, geometry, ksplit, stats_rows)"""
    import ctypes
    from fplx import _lib
    kern, g, ks, r = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(_lib.lib().fplx_conv3d_plan_query(n, d, h, w, cin, cout, k[0], k[1], k[2], x_dt, y_dt, ctypes.byref(kern),
                                                 ctypes.byref(g), ctypes.byref(ks), ctypes.byref(r)))
    return (kern.value, g.value, ks.value, r.value) if full else kern.value

"""Deterministic, platform-independent synthetic tensors for the parity fixtures.

Used by BOTH the fixture generator (tests/golden/make_golden.py, build container,
imports the reference) and the tests / smoke / bench (GPU box, no reference).
Every array is a pure function of (name, shape): numpy Philox keyed by crc32(name).
"""
import zlib
import numpy as np


def _rng(name):
    return np.random.Generator(np.random.Philox(key=zlib.crc32(name.encode()) + 0x5EED))


def normal(name, shape, scale=1.0, shift=0.0):
    return (_rng(name).standard_normal(size=tuple(shape)) * scale + shift).astype(np.float32)


def uniform(name, shape, lo=0.0, hi=1.0):
    return _rng(name).uniform(lo, hi, size=tuple(shape)).astype(np.float32)


def ball_label(shape_dhw, radius, n=1, class_num=2, offsets=None):
    """One-hot float label [n, class_num, D, H, W]: a centred (or offset) ball = class 1."""
    D, H, W = shape_dhw
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    out = np.zeros((n, class_num, D, H, W), np.float32)
    for i in range(n):
        off = (0, 0, 0) if offsets is None else offsets[i]
        m = ((zz - D / 2 - off[0]) ** 2 + (yy - H / 2 - off[1]) ** 2 +
             (xx - W / 2 - off[2]) ** 2) <= radius ** 2
        out[i, 1][m] = 1.0
        out[i, 0][~m] = 1.0
    return out


def state_dict_3d(net_params, prefix="w"):
    """Deterministic weights for the live members of UNet2D5_dsbn (conv_dims[l] = 3: the 3D ones; 2: the 2D ones).

    Key names and shapes follow the reference's state_dict for the 3D branch
    (PyMIC/pymic/net/net3d/unet2d5_dsbn.py:48-83,131-154,265-294 and
    PyMIC/pymic/net_run_dsbn/dsbn.py:35-57).  Returned as {key: np.ndarray}.
    """
    ft = net_params["feature_chns"]
    cin = net_params["in_chns"]
    nd = net_params["num_domains"]
    ncls = net_params["class_num"]
    sd = {}

    def conv(key, co, ci, k):
        fan_in = ci * int(np.prod(k))
        sd[key + ".weight"] = normal(prefix + key + ".w", (co, ci) + tuple(k), scale=(2.0 / fan_in) ** 0.5)
        sd[key + ".bias"] = normal(prefix + key + ".b", (co,), scale=0.1)

    def dsbn(key, c):
        for d in range(nd):
            k = "%s.bns.%d" % (key, d)
            sd[k + ".weight"] = normal(prefix + k + ".w", (c,), 0.2, 1.0)
            sd[k + ".bias"] = normal(prefix + k + ".b", (c,), 0.2)
            sd[k + ".running_mean"] = normal(prefix + k + ".rm", (c,), 0.2)
            sd[k + ".running_var"] = uniform(prefix + k + ".rv", (c,), 0.5, 1.5)
            sd[k + ".num_batches_tracked"] = np.zeros((), np.int64)

    dims = list(net_params.get("conv_dims", [3] * 5))

    def block(key, ci, co, dim=3):
        k = (3,) * dim
        conv(key + ".conv%dd_1" % dim, co, ci, k)
        conv(key + ".conv%dd_2" % dim, co, co, k)
        dsbn(key + ".bn%dd1" % dim, co)
        dsbn(key + ".bn%dd2" % dim, co)
        sd[key + ".relu_1.weight"] = uniform(prefix + key + ".r1", (1,), 0.1, 0.4)
        sd[key + ".relu_2.weight"] = uniform(prefix + key + ".r2", (1,), 0.1, 0.4)

    chans = [cin] + list(ft)
    for i in range(5):
        block("block%d.conv" % i, chans[i], chans[i + 1], dims[i])
    for i, (c1, c2) in enumerate([(ft[4], ft[3]), (ft[3], ft[2]), (ft[2], ft[1]), (ft[1], ft[0])]):
        key = "up%d" % (i + 1)
        dim = dims[3 - i]
        if net_params.get("bilinear", False):
            # bilinear = True: 1x1(x1) convolution `conv{2,3}d` + (tri/bi)linear upsampling (unet2d5_dsbn.py:148-149, 175-176)
            sd[key + ".conv%dd.weight" % dim] = normal(prefix + key + ".c1.w", (c2, c1) + (1,) * dim, scale=(1.0 / c1) ** 0.5)
            sd[key + ".conv%dd.bias" % dim] = normal(prefix + key + ".c1.b", (c2,), scale=0.1)
        else:
            # ConvTranspose{2,3}d weight is [Cin, Cout, 2, 2(, 2)]
            sd[key + ".trans%dd.weight" % dim] = normal(prefix + key + ".t.w", (c1, c2) + (2,) * dim, scale=(1.0 / c1) ** 0.5)
            sd[key + ".trans%dd.bias" % dim] = normal(prefix + key + ".t.b", (c2,), scale=0.1)
        block(key + ".conv", 2 * c2, c2, dim)
    conv("out_conv", ncls, ft[0], (1, 3, 3))
    return sd

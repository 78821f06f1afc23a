"""Network configs / shapes / labels shared by the fixture generator and the tests."""
import numpy as np
import detdata

NETS = {
    "tiny": dict(in_chns=1, feature_chns=[8, 16, 32, 64, 128], dropout=[0, 0, 0, 0, 0],
                 conv_dims=[3, 3, 3, 3, 3], class_num=2, bilinear=False, num_domains=2,
                 net_type="UNet2D5_dsbn"),
    "cfg1": dict(in_chns=1, feature_chns=[16, 32, 64, 128, 256], dropout=[0, 0, 0, 0, 0],
                 conv_dims=[3, 3, 3, 3, 3], class_num=2, bilinear=False, num_domains=2,
                 net_type="UNet2D5_dsbn"),
    "c4": dict(in_chns=4, feature_chns=[8, 16, 32, 64, 128], dropout=[0, 0, 0, 0, 0],
               conv_dims=[3, 3, 3, 3, 3], class_num=3, bilinear=False, num_domains=2,
               net_type="UNet2D5_dsbn"),
}
# the shipped configs' dimensionality pattern (config_dual/data_vs/vs_t1s_g.cfg:58): 2D convolutions at levels 0-1
NETS["tiny25"] = dict(NETS["tiny"], conv_dims=[2, 2, 3, 3, 3])
# bilinear = True: UpBlock = 1x1(x1) convolution + (tri / bi)linear upsampling with align_corners (unet2d5_dsbn.py:148-149, 175-176)
NETS["tinybl"] = dict(NETS["tiny"], bilinear=True)
NETS["tinybl25"] = dict(NETS["tiny"], bilinear=True, conv_dims=[2, 2, 3, 3, 3])
SHAPES = {"tinybl": (2, 1, 16, 32, 32), "tinybl25": (2, 1, 16, 32, 32), "tiny25": (2, 1, 16, 32, 32), "tiny": (2, 1, 16, 32, 32), "cfg1": (1, 1, 32, 64, 64), "c4": (2, 4, 16, 32, 32)}


def label_for(name):
    n, _, D, H, W = SHAPES[name]
    cls = NETS[name]["class_num"]
    lab = detdata.ball_label((D, H, W), radius=min(D, H, W) / 4.0, n=n, class_num=2,
                             offsets=[(0, 1, -2), (1, -3, 2)][:n])
    if cls == 3:  # split the ball into two classes by x
        out = np.zeros((n, 3, D, H, W), np.float32)
        out[:, 0] = lab[:, 0]
        half = np.zeros((D, H, W), bool)
        half[:, :, W // 2:] = True
        out[:, 1] = lab[:, 1] * half
        out[:, 2] = lab[:, 1] * (~half)
        lab = out
    return lab


_LEVEL_OF = {"block0": 0, "block1": 1, "block2": 2, "block3": 3, "block4": 4, "up1": 3, "up2": 2, "up3": 1, "up4": 0}


def key_for(name, params):
    """state_dict key of a parameter given by its 3D-style name: the members of a conv_dims = 2 level are conv2d_* /
    bn2d* / trans2d"""
    lvl = _LEVEL_OF.get(name.split(".")[0])
    if lvl is None or params["conv_dims"][lvl] == 3:
        return name
    return name.replace("conv3d_", "conv2d_").replace("bn3d", "bn2d").replace("trans3d", "trans2d").replace(".conv3d.", ".conv2d.")

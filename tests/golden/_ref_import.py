"""Import helper for the READ-ONLY reference tree (/root/reference).

Only used by tests/golden/make_golden.py in the build container to GENERATE
fixtures.  Nothing here (and nothing under /root/reference) is available on the
GPU box; tests read the committed .npz/.json files instead.

The reference cannot be imported as published: it needs
torchvision, SimpleITK, GeodisTK, cv2, skimage, tensorboardX and a package
`pymic.net.net2d` that does not exist.  We inject empty stub modules for those.
"""
import sys
import types

REF_ROOT = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    for p in (REF_ROOT, REF_ROOT + "/PyMIC"):
        if p not in sys.path:
            sys.path.insert(0, p)

    class _Compose(object):
        def __init__(self, transforms):
            self.transforms = transforms

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms", Compose=_Compose)
    tv.utils = _stub("torchvision.utils")
    tv.models = _stub("torchvision.models")
    _stub("SimpleITK")
    _stub("GeodisTK")
    _stub("cv2")
    sk = _stub("skimage")
    sk.measure = _stub("skimage.measure")
    sk.draw = _stub("skimage.draw")
    sk.morphology = _stub("skimage.morphology")
    _stub("tensorboardX", SummaryWriter=object)

    class _Dummy(object):
        def __init__(self, *a, **k):
            raise RuntimeError("stubbed 2D network")

    import importlib
    pymic_net = importlib.import_module("pymic.net")
    net2d = _stub("pymic.net.net2d")
    pymic_net.net2d = net2d
    for sub, cls in [("unet2d", "UNet2D"), ("unet2d_dual_branch", "UNet2D_DualBranch"),
                     ("unet2d_urpc", "UNet2D_URPC"), ("unet2d_cct", "UNet2D_CCT"),
                     ("cople_net", "COPLENet"), ("unet2d_attention", "AttentionUNet2D"),
                     ("unet2d_nest", "NestedUNet2D"), ("unet2d_scse", "UNet2D_ScSE")]:
        _stub("pymic.net.net2d." + sub, **{cls: _Dummy})

"""Generates tests/golden/net_tinybl.npz and net_tinybl25.npz by RUNNING the reference network with bilinear = True
(UpBlock = Conv{2,3}d(kernel 1) + nn.Upsample(scale 2, bi/trilinear, align_corners=True),
PyMIC/pymic/net/net3d/unet2d5_dsbn.py:148-150, 172-176), all-3D and in the shipped dimensionality pattern.
Same recipe and deterministic inputs as make_golden.py / make_golden_25d.py.  Build-container only."""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (installs the stubs, imports the reference)
import detdata  # noqa: E402
from make_golden_cfg import NETS, SHAPES, label_for, key_for  # noqa: E402


def build_ref_net(name):
    params = copy.deepcopy(NETS[name])
    torch.manual_seed(1)
    net = MG.UNet2D5_dsbn(params).float()
    sd = detdata.state_dict_3d(params)
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    dead = [k for k in net.state_dict() if k not in sd]
    assert set(missing) <= set(dead)
    return net, params


def gen(name):
    p = NETS[name]
    x = torch.from_numpy(detdata.normal("x." + name, SHAPES[name]))
    y = torch.from_numpy(label_for(name))
    n = x.shape[0]
    out = {}
    net, _ = build_ref_net(name)
    net.eval()
    with torch.no_grad():
        for d in (0, 1):
            out["logits_eval_d%d" % d] = net(x, domain_label=d * torch.ones(n, dtype=torch.long)).numpy()
    for d in (0, 1):
        net, _ = build_ref_net(name)
        net.train()
        logits = net(x, domain_label=d * torch.ones(n, dtype=torch.long))
        out["logits_train_d%d" % d] = logits.detach().numpy()
        loss = MG.DiceLoss()({"prediction": logits, "ground_truth": y})
        out["loss_dice_d%d" % d] = np.float32(loss.item())
        loss.backward()
        named = dict(net.named_parameters())
        norms = {k: float(t.grad.norm()) for k, t in named.items() if t.grad is not None}
        out["gradnorm_keys_d%d" % d] = np.array(sorted(norms.keys()))
        out["gradnorm_vals_d%d" % d] = np.array([norms[k] for k in sorted(norms.keys())], np.float64)
        for k3 in ("up1.conv3d.weight", "up1.conv3d.bias", "up4.conv3d.weight", "up4.conv3d.bias", "up3.conv.conv3d_1.weight",
                   "block4.conv.conv3d_2.weight", "out_conv.weight", "block0.conv.conv3d_1.weight"):
            k = key_for(k3, p)
            g = named[k].grad.numpy()
            if g.size > 20000:                          # keep the fixture small: strided sample of the big tensors
                stride = g.size // 10000
                out["gradsub%d_d%d.%s" % (stride, d, k)] = g.reshape(-1)[::stride].copy()
            else:
                out["grad_d%d.%s" % (d, k)] = g.copy()
    np.savez_compressed(os.path.join(HERE, "net_%s.npz" % name), **out)
    print(name, "ok", {k: v.shape for k, v in out.items() if k.startswith("logits")})


if __name__ == "__main__":
    with MG.quiet():
        pass
    for nm in ("tinybl", "tinybl25"):
        gen(nm)

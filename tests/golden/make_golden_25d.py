"""Generates tests/golden/net_tiny25.npz and train_step25.npz by RUNNING the reference network in the dimensionality
pattern of its shipped configs, conv_dims = [2, 2, 3, 3, 3] (config_dual/data_vs/vs_t1s_g.cfg:58): 2D convolutions /
MaxPool2d / ConvTranspose2d on depth-folded tensors at levels 0-1 (PyMIC/pymic/net/net3d/unet2d5_dsbn.py:66-73, 110-127,
160-188), 3D below.  Same recipe as make_golden.py (gen_net / gen_train_step), same deterministic inputs.
Build-container only."""
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

NAME = "tiny25"


def build_ref_net(name):
    params = copy.deepcopy(NETS[name])
    torch.manual_seed(1)
    net = MG.UNet2D5_dsbn(params).float()
    sd = detdata.state_dict_3d(params)
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    dead = [k for k in net.state_dict() if k not in sd]
    assert len(dead) == 240 and set(missing) <= set(dead)   # twins of the other dimensionality + the bilinear 1x1 convs
    return net, params


def gen_net():
    p = NETS[NAME]
    x = torch.from_numpy(detdata.normal("x." + NAME, SHAPES[NAME]))
    y = torch.from_numpy(label_for(NAME))
    n = x.shape[0]
    out = {}
    net, _ = build_ref_net(NAME)
    net.eval()
    with torch.no_grad():
        for d in (0, 1):
            out["logits_eval_d%d" % d] = net(x, domain_label=d * torch.ones(n, dtype=torch.long)).numpy()
    for d in (0, 1):
        net, _ = build_ref_net(NAME)
        net.train()
        logits = net(x, domain_label=d * torch.ones(n, dtype=torch.long))
        out["logits_train_d%d" % d] = logits.detach().numpy()
        loss = MG.DiceLoss()({"prediction": logits, "ground_truth": y})
        out["loss_dice_d%d" % d] = np.float32(loss.item())
        loss.backward()
        sd = net.state_dict()
        for bn in ("block0.conv.bn3d1", "block1.conv.bn3d2", "block4.conv.bn3d2", "up4.conv.bn3d2"):
            for dd in (0, 1):
                for s in ("running_mean", "running_var", "num_batches_tracked"):
                    k = key_for("%s.bns.%d.%s" % (bn, dd, s), p)
                    out["d%d.%s" % (d, k)] = sd[k].numpy().copy()
        named = dict(net.named_parameters())
        norms = {k: float(t.grad.norm()) for k, t in named.items() if t.grad is not None}
        out["gradnorm_keys_d%d" % d] = np.array(sorted(norms.keys()))
        out["gradnorm_vals_d%d" % d] = np.array([norms[k] for k in sorted(norms.keys())], np.float64)
        for k3 in MG.GRAD_KEYS + ["up3.trans3d.weight", "up3.conv.conv3d_1.weight", "block1.conv.conv3d_2.weight"]:
            k = key_for(k3, p)
            g = named[k].grad.numpy()
            if g.size > 100000:
                stride = g.size // 50000
                out["gradsub%d_d%d.%s" % (stride, d, k)] = g.reshape(-1)[::stride].copy()
            else:
                out["grad_d%d.%s" % (d, k)] = g.copy()
        for bn in ("block0.conv.bn3d1", "block2.conv.bn3d2", "up4.conv.bn3d2"):
            for s in ("weight", "bias"):
                k = key_for("%s.bns.%d.%s" % (bn, d, s), p)
                out["grad_d%d.%s" % (d, k)] = named[k].grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "net_%s.npz" % NAME), **out)
    print("net", NAME, {k: v.shape for k, v in out.items() if k.startswith("logits")},
          [float(out["loss_dice_d%d" % d]) for d in (0, 1)])


def gen_train_step():
    MG.build_ref_net = build_ref_net                     # make_agent builds the network through this hook
    p = NETS[NAME]
    agent, cfg = MG.make_agent(NAME)
    with MG.quiet():
        agent.create_optimizer(agent.get_parameters_to_update())
        agent.create_loss_calculator()
    n, _, D, H, W = SHAPES[NAME]
    batches = []
    for dom in (0, 1):
        x = detdata.normal("ts25.x.d%d" % dom, SHAPES[NAME])
        lab = detdata.ball_label((D, H, W), 5.0, n=n, offsets=[(dom, 1, -2), (1, -3, 2 + dom)])
        b = {"image": torch.from_numpy(x), "label_prob": torch.from_numpy(lab)}
        if dom == 1:
            pw = (detdata.uniform("ts25.pw", (n, 1, D, H, W)) > 0.25).astype(np.float32)
            iw = np.array([0.8, 0.45], np.float32)
            b["pixel_weight"] = torch.from_numpy(pw * iw[:, None, None, None, None])
            b["image_weight"] = torch.from_numpy(iw)
        batches.append(b)
    agent.train_loader_1, agent.train_loader_2 = [batches[0]], [batches[1]]
    watch = [key_for(k, p) for k in ("out_conv.weight", "block0.conv.conv3d_1.weight", "block4.conv.conv3d_2.bias",
                                     "up1.trans3d.weight", "up4.trans3d.weight", "block0.conv.bn3d1.bns.0.weight",
                                     "block1.conv.bn3d1.bns.1.bias", "block0.conv.relu_1.weight",
                                     "up4.conv.bn3d2.bns.1.running_var", "up3.conv.conv3d_2.weight")]
    out, lrs = {}, []
    for step in range(1, 6):
        with MG.quiet():
            sc = agent.training_all()
        out["step%d.loss" % step] = np.float64(sc["loss"])
        out["step%d.class_dice" % step] = np.asarray(sc["class_dice"], np.float64)
        lrs.append(agent.optimizer.param_groups[0]["lr"])
        if step in (1, 3):
            sd = agent.net.state_dict()
            for k in watch:
                out["step%d.%s" % (step, k)] = sd[k].numpy().copy()
    out["lrs"] = np.array(lrs, np.float64)
    np.savez_compressed(os.path.join(HERE, "train_step25.npz"), **out)
    print("train_step25", {k: float(v) for k, v in out.items() if k.endswith("loss")})


if __name__ == "__main__":
    gen_net()
    gen_train_step()

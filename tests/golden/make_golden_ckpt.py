"""Generates tests/golden/ref_ckpt_3.pt + ref_ckpt.npz + ref_state_keys.json by RUNNING the reference
(PyMIC/pymic/net_run_dsbn/agent_seg.py: training_all 415-508, the save_dict of train_valid 786-799,
get_optimizer.py: torch.optim.Adam + MultiStepLR): a small UNet2D5_dsbn trains 3 iterations, its checkpoint dictionary
{'iteration', 'valid_pred', 'model_state_dict', 'optimizer_state_dict'} is written with torch.save exactly as the
reference does, then training continues 2 more iterations and the losses / parameters after iteration 5 are recorded.
A build that loads the .pt and resumes must land on the same numbers.  ref_state_keys.json holds the ordered
state_dict keys (484) and named_parameters (268) of the reference network: the save side must reproduce them.
Build-container only."""
import copy
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (installs the stubs, imports the reference)
import detdata  # noqa: E402
from make_golden_cfg import NETS  # noqa: E402

NET = dict(in_chns=1, feature_chns=[4, 4, 8, 8, 8], dropout=[0, 0, 0, 0, 0], conv_dims=[3, 3, 3, 3, 3], class_num=2,
           bilinear=False, num_domains=2, net_type="UNet2D5_dsbn")
SHAPE = (2, 1, 16, 32, 32)
WATCH = ["out_conv.weight", "block0.conv.conv3d_1.weight", "block4.conv.conv3d_2.bias", "up1.trans3d.weight",
         "block0.conv.bn3d1.bns.0.weight", "block0.conv.bn3d1.bns.1.bias", "block0.conv.relu_1.weight",
         "up4.conv.bn3d2.bns.1.running_var", "up4.conv.bn3d2.bns.0.num_batches_tracked"]


def batches():
    n, _, D, H, W = SHAPE
    out = []
    for dom in (0, 1):
        x = detdata.normal("ck.x.d%d" % dom, SHAPE)
        lab = detdata.ball_label((D, H, W), 5.0, n=n, offsets=[(dom, 1, -2), (1, -3, 2 + dom)])
        out.append({"image": torch.from_numpy(x), "label_prob": torch.from_numpy(lab)})
    return out


def main():
    NETS["ckpt"] = NET
    agent, cfg = MG.make_agent("ckpt", {"lr_milestones": [2, 4]})
    with MG.quiet():
        agent.create_optimizer(agent.get_parameters_to_update())
        agent.create_loss_calculator()
    b = batches()
    agent.train_loader_1, agent.train_loader_2 = [b[0]], [b[1]]
    out = {"x0": b[0]["image"].numpy(), "x1": b[1]["image"].numpy(), "lab0": b[0]["label_prob"].numpy(),
           "lab1": b[1]["label_prob"].numpy()}
    keys = {"state_dict": list(agent.net.state_dict().keys()),
            "named_parameters": [n for n, _ in agent.net.named_parameters()],
            "shapes": {k: list(v.shape) for k, v in agent.net.state_dict().items()}}
    json.dump(keys, open(os.path.join(HERE, "ref_state_keys.json"), "w"))
    for step in range(1, 6):
        with MG.quiet():
            sc = agent.training_all()
        out["step%d.loss" % step] = np.float64(sc["loss"])
        out["step%d.lr" % step] = np.float64(agent.optimizer.param_groups[0]["lr"])
        if step == 3:
            save_dict = {'iteration': 3, 'valid_pred': 0.4321, 'model_state_dict': agent.net.state_dict(),
                         'optimizer_state_dict': agent.optimizer.state_dict()}
            torch.save(save_dict, os.path.join(HERE, "ref_ckpt_3.pt"))
        if step in (3, 5):
            sd = agent.net.state_dict()
            for k in WATCH:
                out["step%d.%s" % (step, k)] = sd[k].numpy().copy()
    osd = agent.optimizer.state_dict()
    out["opt.n_state"] = np.int64(len(osd["state"]))
    out["opt.steps"] = np.array([float(osd["state"][i]["step"]) for i in sorted(osd["state"])])
    np.savez_compressed(os.path.join(HERE, "ref_ckpt.npz"), **out)
    print({k: float(v) for k, v in out.items() if k.endswith("loss") or k.endswith("lr")})
    print("state entries", len(osd["state"]), "group keys", sorted(osd["param_groups"][0].keys()))
    print(os.path.getsize(os.path.join(HERE, "ref_ckpt_3.pt")))


if __name__ == "__main__":
    main()

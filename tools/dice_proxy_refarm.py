"""The REFERENCE arm of the Dice proxy (tools/dice_proxy.py): the reference's own code - pymic UNet2D5_dsbn, DiceLoss,
SegmentationAgent.training_all (agent_seg.py:415-508: both domains, loss = (l0 + l1) / 2, torch.optim.Adam) - trained on the CPU
in the BUILD container from the same initial weights on the same batches in the same order as the fplx arms, then the same
held-out volumes segmented (eval-mode BatchNorm, argmax) and scored with the reference's binary_dice
(util/evaluation_seg_train.py:21-50).  Writes profiles/r03_dice_proxy_reference_arm.json (per batch order: per-volume Dice in
percent points, loss trajectory), which tools/dice_proxy.py --reference-arm reads on the GPU box to report
"fplx fp32 - reference" and "fplx bf16 - reference", paired per batch order.  Nothing here runs on the GPU box (the reference
does not travel); this script is the committed generator of that data file.

    python tools/dice_proxy_refarm.py [--seeds 24] [--iters 300] [--base 16] [--out profiles/r03_dice_proxy_reference_arm.json]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fpl-plus_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import _ref_import  # noqa: E402

_ref_import.install()
from pymic.net.net3d.unet2d5_dsbn import UNet2D5_dsbn as RefNet  # noqa: E402
from pymic.net_run_dsbn import agent_seg as ref_agent_seg  # noqa: E402
from pymic.util.evaluation_seg_train import binary_dice as ref_binary_dice  # noqa: E402

import fplx  # noqa: E402  (initial weights only: the same constructor call as tools/dice_proxy.py, on the CPU)
from dice_proxy import make_case, to_batch  # noqa: E402


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--base", type=int, default=16)
    ap.add_argument("--dims", default="33333")
    ap.add_argument("--shape", default="32,64,64")
    ap.add_argument("--seeds", type=int, default=24)
    ap.add_argument("--first-seed", type=int, default=0)
    ap.add_argument("--held-out", type=int, default=16)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_dice_proxy_reference_arm.json"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    shape = tuple(int(t) for t in a.shape.split(","))
    b = a.base
    p = dict(in_chns=1, feature_chns=[b, 2 * b, 4 * b, 8 * b, 16 * b], dropout=[0.0] * 5, conv_dims=[int(c) for c in a.dims],
             class_num=2, bilinear=False, num_domains=2, net_type="UNet2D5_dsbn")
    rs = np.random.RandomState(7)                                              # the data of tools/dice_proxy.py, same generator
    train = [[make_case(rs, shape, d) for _ in range(13)] for d in (0, 1)]
    test = [[make_case(rs, shape, d) for _ in range(a.held_out)] for d in (0, 1)]
    res = {}
    if os.path.exists(a.out):
        res = json.load(open(a.out)).get("orders", {})
    for seed in range(a.first_seed, a.first_seed + a.seeds):
        if str(seed) in res:
            continue
        t0 = time.time()
        ors = np.random.RandomState(100 + seed)
        order = [[ors.permutation(13)[:2] for _ in range(a.iters)] for _ in (0, 1)]
        torch.manual_seed(1 + seed)
        init = fplx.UNet2D5_dsbn(dict(p)).state_dict()
        cfg = {"dataset": {"tensor_type": "float", "task_type": "seg", "root_dir": "/", "train_batch_size": 2},
               "network": dict(p),
               "training": {"dis": False, "train_fpl_uda": False, "loss_type": "DiceLoss", "optimizer": "Adam",
                            "learning_rate": 1e-3, "momentum": 0.9, "weight_decay": 1e-5, "lr_scheduler": "MultiStepLR",
                            "lr_gamma": 0.5, "lr_milestones": [10 ** 9], "iter_valid": a.iters,
                            "ckpt_save_dir": "/tmp/fplx_refarm_ckpt", "gpus": [0]},
               "testing": {}}
        with quiet():
            agent = ref_agent_seg.SegmentationAgent(cfg, "train")
            net = RefNet(dict(p)).float()
            missing, unexpected = net.load_state_dict(init, strict=False)
            assert not unexpected, unexpected
            for k in missing:                       # the reference's dead twins (2D members of an all-3D net, bilinear branch)
                assert ("2d" in k) or (".conv3d." in k and k.startswith("up")), k
            agent.set_network(net)
            agent.create_network()
            agent.device = torch.device("cpu")
            agent.checkpoint = None
            agent.fpl_uda = False
            agent.create_optimizer(agent.get_parameters_to_update())
            agent.create_loss_calculator()
        loaders = []
        for d in (0, 1):
            bl = []
            for it in range(a.iters):
                x, y = to_batch([train[d][i] for i in order[d][it]])
                bl.append({"image": x, "label_prob": y})
            loaders.append(bl)
        agent.train_loader_1, agent.train_loader_2 = loaders
        with quiet():
            sc = agent.training_all()
        agent.net.eval()
        dice = []
        with torch.no_grad():
            for d in (0, 1):
                for img, lab in test[d]:
                    lg = agent.net(torch.from_numpy(img[None, None]), domain_label=d * torch.ones(1, dtype=torch.long))
                    seg = torch.argmax(lg, dim=1)[0].numpy().astype(np.uint8)
                    dice.append(float(ref_binary_dice(seg, lab)))
        res[str(seed)] = {"dice_percent": [100.0 * v for v in dice], "train_loss_mean": float(sc["loss"]),
                          "seconds": round(time.time() - t0, 1)}
        print("order %d: reference mean Dice %.2f, mean train loss %.4f, %.0f s" %
              (seed, np.mean(res[str(seed)]["dice_percent"]), sc["loss"], time.time() - t0), flush=True)
        json.dump({"what": "reference arm of tools/dice_proxy.py: pymic UNet2D5_dsbn + DiceLoss + SegmentationAgent.training_all "
                           "(torch CPU, %d threads), generated by tools/dice_proxy_refarm.py" % a.threads,
                   "config": {"base": b, "dims": a.dims, "shape": list(shape), "iters": a.iters, "held_out": a.held_out,
                              "dropout": [0.0] * 5, "lr": 1e-3, "weight_decay": 1e-5},
                   "orders": res}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()

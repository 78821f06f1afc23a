"""Dice proxy for the north_star target "Dice on VS hrT2 within +-0.5 of the reference" (no VS data, no trained reference
checkpoint and no network on the box, so the real comparison cannot be run): the SAME initial weights and the SAME batches
are trained three ways -

    fp32    fplx parity mode (fp32 activations; logits match the reference's CPU path to 1e-3, tests/test_gpu_net_parity.py)
    bf16    fplx performance mode (bf16 activations, the benchmarked kernels)
    oracle  the CPU restatement of the reference for the first --oracle-iters iterations (trajectory check)

on a synthetic two-domain tumour task (domain 1 = contrast-shifted, noisier images), `training_all` iterations (both domains,
one Adam step), then the held-out volumes of both domains are segmented (eval-mode BatchNorm) and scored with
fplx.evaluation.binary_dice - the reference's evaluation function (util/evaluation_seg_train.py:21-50).
Prints per-precision mean Dice in percent points and the bf16 - fp32 difference.

    python tools/dice_proxy.py [--iters 300] [--base 16] [--oracle-iters 3] [--dims 33333]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fpl-plus_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fplx  # noqa: E402
from fplx import evaluation  # noqa: E402


def make_case(rs, shape, domain):
    D, H, W = shape
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    c = np.array([rs.uniform(0.3, 0.7) * D, rs.uniform(0.3, 0.7) * H, rs.uniform(0.3, 0.7) * W])
    r = np.array([rs.uniform(3, 6), rs.uniform(5, 10), rs.uniform(5, 10)])
    lab = (((zz - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((xx - c[2]) / r[2]) ** 2) <= 1.0
    # smooth background structure + tumour contrast + noise; domain 1: weaker contrast, stronger noise, intensity shift
    bg = np.sin(zz / 5.0 + rs.uniform(0, 6)) * np.cos(yy / 9.0 + rs.uniform(0, 6)) + 0.5 * np.sin(xx / 7.0 + rs.uniform(0, 6))
    contrast, noise, shift = ((1.6, 0.6, 0.0), (1.0, 0.9, 0.7))[domain]
    img = bg + contrast * lab + noise * rs.randn(D, H, W) + shift
    img = (img - img.mean()) / img.std()
    return img.astype(np.float32), lab.astype(np.uint8)


def to_batch(cases):
    x = torch.from_numpy(np.stack([c[0] for c in cases])[:, None])
    l = np.stack([c[1] for c in cases])
    y = torch.from_numpy(np.stack([1 - l, l], 1).astype(np.float32))
    return x, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--base", type=int, default=16)
    ap.add_argument("--oracle-iters", type=int, default=3)
    ap.add_argument("--dims", default="33333")
    ap.add_argument("--shape", default="32,64,64")
    ap.add_argument("--seeds", type=int, default=1, help="repeat with this many batch orders: the spread between them is the scale any fp32-vs-bf16 difference has to be read on")
    ap.add_argument("--dropout", default="0,0,0,0,0", help="e.g. 0,0,0.3,0.4,0.5 (then the CPU oracle trajectory is not comparable)")
    a = ap.parse_args()
    shape = tuple(int(t) for t in a.shape.split(","))
    b = a.base
    p = dict(in_chns=1, feature_chns=[b, 2 * b, 4 * b, 8 * b, 16 * b], dropout=[float(t) for t in a.dropout.split(",")],
             conv_dims=[int(c) for c in a.dims], class_num=2, bilinear=False, num_domains=2, net_type="UNet2D5_dsbn")
    rs = np.random.RandomState(7)
    train = [[make_case(rs, shape, d) for _ in range(13)] for d in (0, 1)]          # 13 cases per domain, like the sample data
    test = [[make_case(rs, shape, d) for _ in range(8)] for d in (0, 1)]
    per_seed = []
    for seed in range(a.seeds):
        rs = np.random.RandomState(100 + seed) if seed else rs
        order = [[rs.permutation(13)[:2] for _ in range(a.iters)] for _ in (0, 1)]      # the same batches for every run
        torch.manual_seed(1)
        init = fplx.UNet2D5_dsbn(dict(p)).state_dict()
        results, traj = {}, {}
        for prec in ("fp32", "bf16"):
            net = fplx.UNet2D5_dsbn(dict(p, precision=prec))
            net.load_state_dict(init)
            net.cuda()
            ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-3, weight_decay=1e-5)
            losses = []
            for it in range(a.iters):
                bs = []
                for d in (0, 1):
                    x, y = to_batch([train[d][i] for i in order[d][it]])
                    bs.append({"image": x.cuda(), "label_prob": y.cuda()})
                outs = ts.step_all(bs)
                losses.append(outs)
            traj[prec] = [0.5 * (float(o[0][0].item()) + float(o[1][0].item())) for o in losses]
            net.eval()
            dice = []
            with torch.no_grad():
                for d in (0, 1):
                    for img, lab in test[d]:
                        lg = net(torch.from_numpy(img[None, None]).cuda(), domain_label=d * torch.ones(1, dtype=torch.long))
                        seg = fplx.filter.hard_label(lg)[0]
                        dice.append(evaluation.binary_dice(seg, torch.from_numpy(lab).cuda()))
            results[prec] = np.asarray(dice, np.float64)
        per_seed.append((100 * results["fp32"].mean(), 100 * results["bf16"].mean()))
    if a.oracle_iters > 0:
        from oracle import torch_ref as R
        torch.set_num_threads(max(1, min(os.cpu_count() or 1, 32)))
        sd, prm = R.split_state({k: v.numpy().copy() for k, v in init.items()})
        opt = R.AdamRef(prm, 1e-3, 1e-5)
        loss_fn = R.loss_from_config({"loss_type": "DiceLoss"})
        p0 = dict(p, dropout=[0, 0, 0, 0, 0])
        tr = []
        for it in range(a.oracle_iters):
            bs = []
            for d in (0, 1):
                x, y = to_batch([train[d][i] for i in order[d][it]])
                bs.append({"image": x, "label_prob": y})
            tr.append(float(R.training_all_step(sd, prm, opt, p0, bs, loss_fn)[0]))
        traj["oracle"] = tr
    print("config: %d-base, conv_dims %s, crops %s, %d training_all iterations (2 crops per domain), 8 held-out volumes per domain"
          % (b, a.dims, shape, a.iters))
    for k in ("fp32", "bf16"):
        t = traj[k]
        print("%-6s loss: first %s ... last 5 mean %.4f | Dice %% mean %.2f  (domain 0 %.2f, domain 1 %.2f), min %.2f"
              % (k, ["%.4f" % v for v in t[:3]], float(np.mean(t[-5:])), 100 * results[k].mean(), 100 * results[k][:8].mean(),
                 100 * results[k][8:].mean(), 100 * results[k].min()))
    if "oracle" in traj:
        print("oracle (CPU restatement of the reference, dropout off) loss of the first iterations: %s"
              % ["%.4f" % v for v in traj["oracle"]])
    diff = 100 * (results["bf16"].mean() - results["fp32"].mean())
    print("bf16 - fp32 mean Dice (last batch order): %+.2f points (north_star tolerance +-0.5); per-volume |difference| max %.2f points"
          % (diff, 100 * np.abs(results["bf16"] - results["fp32"]).max()))
    if a.seeds > 1:
        ps = np.asarray(per_seed)
        print("over %d batch orders: fp32 Dice %s (mean %.2f, std %.2f) | bf16 Dice %s (mean %.2f, std %.2f) | mean difference %+.2f points"
              % (a.seeds, ["%.2f" % v for v in ps[:, 0]], ps[:, 0].mean(), ps[:, 0].std(), ["%.2f" % v for v in ps[:, 1]],
                 ps[:, 1].mean(), ps[:, 1].std(), (ps[:, 1] - ps[:, 0]).mean()))


if __name__ == "__main__":
    main()

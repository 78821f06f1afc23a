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

Round 3: a PAIRED design that can decide something - every batch order trains all arms from the same initial
weights on the same batches with the same (counter-based) dropout masks; reported are the per-order differences of the mean
Dice with their standard error, the per-volume distribution, and a CONTROL arm (fp32 from weights perturbed by 1e-6): the
spread two trainings show when they differ by far less than bf16 rounding - the scale a bf16-vs-fp32 difference has to be
read on.  The real-data arm on the 13 volumes the reference ships is tools/dice_real.py.

    python tools/dice_proxy.py [--iters 300] [--base 16] [--oracle-iters 3] [--dims 33333] [--seeds 24] [--control 1e-6]
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


def run_arm(p, prec, init, train, test, order, iters, perturb=0.0):
    """train one arm from `init` on the given batch order, return (loss trajectory, per-volume Dice of the held-out volumes).
    perturb > 0: the CONTROL arm - every initial parameter multiplied by (1 + perturb * N(0, 1)) (fixed generator): how far
    apart two fp32 trainings end up when they differ by far less than one bf16 rounding."""
    net = fplx.UNet2D5_dsbn(dict(p, precision=prec))
    sd = {k: v.clone() for k, v in init.items()}
    if perturb > 0:
        g = torch.Generator().manual_seed(99)
        for k, v in sd.items():
            if v.dtype.is_floating_point and "running_" not in k:
                sd[k] = v * (1.0 + perturb * torch.randn(v.shape, generator=g))
    net.load_state_dict(sd)
    net.cuda()
    net.dropout_seed = 4321                              # counter-based Philox masks: identical in every arm
    ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-3, weight_decay=1e-5)
    traj = []
    for it in range(iters):
        bs = []
        for d in (0, 1):
            x, y = to_batch([train[d][i] for i in order[d][it]])
            bs.append({"image": x.cuda(), "label_prob": y.cuda()})
        outs = ts.step_all(bs)
        traj.append(outs)
    traj = [0.5 * (float(o[0][0].item()) + float(o[1][0].item())) for o in traj]
    net.eval()
    dice = []
    with torch.no_grad():
        for d in (0, 1):
            for img, lab in test[d]:
                lg = net(torch.from_numpy(img[None, None]).cuda(), domain_label=d * torch.ones(1, dtype=torch.long))
                seg = fplx.filter.hard_label(lg)[0]
                dice.append(evaluation.binary_dice(seg, torch.from_numpy(lab).cuda()))
    return traj, np.asarray(dice, np.float64)


def paired_report(name, a, b):
    """a, b: [orders, volumes] Dice in percent points of two arms trained on the same batches -> text lines"""
    d = b.mean(1) - a.mean(1)                            # per-order difference of the mean Dice
    n = len(d)
    se = d.std(ddof=1) / np.sqrt(n) if n > 1 else float("nan")
    pv = (b - a).reshape(-1)
    q = np.percentile(pv, [5, 25, 50, 75, 95])
    lines = ["%s over %d batch orders (paired: same init, same batches, same dropout masks):" % (name, n),
             "  per-order difference of the mean Dice: %s" % " ".join("%+.2f" % v for v in d),
             "  mean %+.3f points, std %.3f, standard error %.3f -> |mean| + 2 SE = %.3f (north_star tolerance 0.5)"
             % (d.mean(), d.std(ddof=1) if n > 1 else float("nan"), se, abs(d.mean()) + 2 * se),
             "  per-volume differences (%d): 5/25/50/75/95 %% = %+.2f %+.2f %+.2f %+.2f %+.2f, max |.| %.2f"
             % (pv.size, q[0], q[1], q[2], q[3], q[4], np.abs(pv).max())]
    return lines, dict(per_order=d.tolist(), mean=float(d.mean()), se=float(se), bound=float(abs(d.mean()) + 2 * se))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--base", type=int, default=16)
    ap.add_argument("--oracle-iters", type=int, default=3)
    ap.add_argument("--dims", default="33333")
    ap.add_argument("--shape", default="32,64,64")
    ap.add_argument("--seeds", type=int, default=1, help="batch orders; every order trains all arms from the same init on the same batches")
    ap.add_argument("--held-out", type=int, default=8, help="held-out volumes per domain")
    ap.add_argument("--control", type=float, default=0.0, help="> 0: third arm = fp32 from an init perturbed by this relative noise (e.g. 1e-6)")
    ap.add_argument("--dropout", default="0,0,0,0,0", help="e.g. 0,0,0.3,0.4,0.5 (then the CPU oracle trajectory is not comparable)")
    ap.add_argument("--reference-arm", default=None,
                    help="json written by tools/dice_proxy_refarm.py (the reference's own training_all on the CPU, same init / "
                         "batches / held-out volumes): adds the paired reports fplx - reference")
    ap.add_argument("--arms", default=None, help="comma list out of fp32,bf16,fp32c (default: fp32,bf16 and fp32c with --control)")
    ap.add_argument("--first-seed", type=int, default=0, help="first batch order (with --seeds: orders first .. first + seeds - 1)")
    ap.add_argument("--save-json", default=None, help="per-order, per-volume Dice of every arm (for merging runs)")
    ap.add_argument("--out", default=None, help="write the report (text) here as well")
    a = ap.parse_args()
    shape = tuple(int(t) for t in a.shape.split(","))
    b = a.base
    p = dict(in_chns=1, feature_chns=[b, 2 * b, 4 * b, 8 * b, 16 * b], dropout=[float(t) for t in a.dropout.split(",")],
             conv_dims=[int(c) for c in a.dims], class_num=2, bilinear=False, num_domains=2, net_type="UNet2D5_dsbn")
    rs = np.random.RandomState(7)
    train = [[make_case(rs, shape, d) for _ in range(13)] for d in (0, 1)]          # 13 cases per domain, like the sample data
    test = [[make_case(rs, shape, d) for _ in range(a.held_out)] for d in (0, 1)]
    arms = ["fp32", "bf16"] + (["fp32c"] if a.control > 0 else [])
    if a.arms:
        arms = [k for k in a.arms.split(",") if k]
        assert all(k in ("fp32", "bf16", "fp32c") for k in arms) and ("fp32c" not in arms or a.control > 0)
    res = {k: [] for k in arms}
    traj, order0, init0 = {}, None, None
    for seed in range(a.first_seed, a.first_seed + a.seeds):
        ors = np.random.RandomState(100 + seed)
        order = [[ors.permutation(13)[:2] for _ in range(a.iters)] for _ in (0, 1)]      # the same batches for every arm
        torch.manual_seed(1 + seed)
        init = fplx.UNet2D5_dsbn(dict(p)).state_dict()
        if seed == a.first_seed:
            order0, init0 = order, init
        for k in arms:
            t, dice = run_arm(p, "bf16" if k == "bf16" else "fp32", init, train, test, order, a.iters, a.control if k == "fp32c" else 0.0)
            res[k].append(100 * dice)
            if seed == a.first_seed:
                traj[k] = t
        print("order %d: %s" % (seed, "  ".join("%s %.2f" % (k, res[k][-1].mean()) for k in arms)), flush=True)
    res = {k: np.asarray(v) for k, v in res.items()}
    if a.save_json:
        import json
        os.makedirs(os.path.dirname(os.path.abspath(a.save_json)), exist_ok=True)
        json.dump({"config": {"base": b, "dims": a.dims, "shape": list(shape), "iters": a.iters, "held_out": a.held_out,
                              "dropout": a.dropout, "control": a.control, "first_seed": a.first_seed},
                   "dice_percent": {k: v.tolist() for k, v in res.items()}}, open(a.save_json, "w"))
    if a.oracle_iters > 0:
        from oracle import torch_ref as R
        torch.set_num_threads(max(1, min(os.cpu_count() or 1, 32)))
        sd, prm = R.split_state({k: v.numpy().copy() for k, v in init0.items()})
        opt = R.AdamRef(prm, 1e-3, 1e-5)
        loss_fn = R.loss_from_config({"loss_type": "DiceLoss"})
        p0 = dict(p, dropout=[0, 0, 0, 0, 0])
        tr = []
        for it in range(a.oracle_iters):
            bs = []
            for d in (0, 1):
                x, y = to_batch([train[d][i] for i in order0[d][it]])
                bs.append({"image": x, "label_prob": y})
            tr.append(float(R.training_all_step(sd, prm, opt, p0, bs, loss_fn)[0]))
        traj["oracle"] = tr
    out = ["config: %d-base, conv_dims %s, crops %s, dropout %s, %d training_all iterations (2 crops per domain), %d held-out volumes per domain"
           % (b, a.dims, shape, a.dropout, a.iters, a.held_out)]
    for k in arms:
        t = traj[k]
        out.append("%-6s (order 0) loss: first %s ... last 5 mean %.4f | Dice %% over orders: %s (mean %.2f, std %.2f)"
                   % (k, ["%.4f" % v for v in t[:3]], float(np.mean(t[-5:])), " ".join("%.2f" % v for v in res[k].mean(1)),
                      res[k].mean(), res[k].mean(1).std()))
    if "oracle" in traj:
        out.append("oracle (CPU restatement of the reference, dropout off) loss of the first iterations: %s"
                   % ["%.4f" % v for v in traj["oracle"]])
    st = None
    if "fp32" in res and "bf16" in res:
        lines, st = paired_report("bf16 - fp32", res["fp32"], res["bf16"])
        out += lines
    if a.control > 0 and "fp32c" in res and st is not None:
        lines, stc = paired_report("CONTROL fp32(init x (1 + %.0e N(0,1))) - fp32" % a.control, res["fp32"], res["fp32c"])
        out += lines
        out.append("reading: the bf16 arm differs from fp32 by bf16 rounding (2^-9 relative) at every stored activation; the control "
                   "differs by %.0e in the initial weights only.  Spread of the per-order differences: bf16 %.2f, control %.2f points."
                   % (a.control, np.std(st["per_order"], ddof=1) if a.seeds > 1 else float("nan"),
                      np.std(stc["per_order"], ddof=1) if a.seeds > 1 else float("nan")))
    if st is not None:
        verdict = "MET" if st["bound"] < 0.5 else "NOT RESOLVED at this number of orders"
        out.append("north_star +-0.5 Dice points on this proxy: |mean| + 2 SE = %.3f -> %s" % (st["bound"], verdict))
    if a.reference_arm:
        import json
        ra = json.load(open(a.reference_arm))
        rc = ra["config"]
        same = (rc["base"] == b and rc["dims"] == a.dims and tuple(rc["shape"]) == shape and rc["iters"] == a.iters and
                rc["held_out"] == a.held_out and [float(t) for t in a.dropout.split(",")] == [float(t) for t in rc["dropout"]])
        seeds = [s_ for s_ in range(a.seeds) if str(a.first_seed + s_) in ra["orders"]]
        if not same or not seeds:
            out.append("reference arm %s: configuration differs or no common batch order - not compared" % a.reference_arm)
        else:
            ref = np.asarray([ra["orders"][str(a.first_seed + s_)]["dice_percent"] for s_ in seeds])
            out.append("REFERENCE arm (%s): Dice %% over orders: %s (mean %.2f, std %.2f)"
                       % (ra["what"], " ".join("%.2f" % v for v in ref.mean(1)), ref.mean(), ref.mean(1).std()))
            for k in [k_ for k_ in ("fp32", "bf16") if k_ in res]:
                lines, sr = paired_report("fplx %s - REFERENCE" % k, ref, res[k][seeds])
                out += lines
                out.append("north_star +-0.5 Dice points, fplx %s against the reference's own training on this proxy: |mean| + 2 SE = "
                           "%.3f -> %s" % (k, sr["bound"], "MET" if sr["bound"] < 0.5 else "NOT RESOLVED at this number of orders"))
    text = "\n".join(out)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()

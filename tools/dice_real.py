"""Real-data arm of the Dice comparison (north_star: "Dice on VS hrT2 within +-0.5 of the reference"):
the 13 NIfTI volumes the reference ships (`tools/copy_refdata.sh` -> tests/golden/_refdata/) through fplx's own data path
(fplx.nifti reader, NiftyDataset, the GPU transforms of config_dual/data_vs/vs_t1s_g.cfg: NormalizeWithMeanStd, Pad,
RandomCrop[28,128,128] with foreground focus, RandomFlip, LabelToProbability), UNet2D5_dsbn with conv_dims [2, 2, 3, 3, 3]
and the shipped dropout, `training_all` iterations (one batch of all cases per domain, one Adam step).

Two protocols (--protocol):
  loo   (default) leave-one-out over the three REAL hrT2 cases that come with labels (hrT2_valid 95, hrT2_test 9,
        hrT2_train 98): domain 1 trains on the other two real hrT2 cases (their own labels) + the two translated hrT2-like
        volumes (label of case 99), domain 0 on ceT1 + its two translated copies; the held-out hrT2 case is segmented
        (sliding window 28 x 128 x 128 + 4-flip TTA, eval-mode BatchNorm) - "Dice on VS hrT2" with target supervision,
        as far as three labelled cases allow.
  uda   the stage-"g" setting of vs_t1s_g.cfg: NO hrT2 label is used (domain 1 = translated volumes only), all three
        real hrT2 cases are test cases.  With ONE labelled source case the target Dice is chaotic (0 - 80 %): kept as the
        record of round 3's first run (profiles/r03_dice_real_uda.txt).
Arms: fp32 parity mode (the stand-in for the reference: logits match its CPU path to 1e-3) and bf16 (the benchmarked
kernels), PAIRED per run: same initial weights, same crops and flips (Python `random` / torch generator re-seeded), same
dropout masks.

    python tools/dice_real.py [--protocol loo] [--seeds 4] [--iters 300] [--base 16] [--out gpurun_out/dice_real.txt]
"""
import argparse
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fpl-plus_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fplx  # noqa: E402
from fplx import evaluation  # noqa: E402
from fplx.dataset import NiftyDataset, BatchLoader  # noqa: E402
from fplx.transform import build_transforms, Compose  # noqa: E402
from dice_proxy import paired_report  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "_refdata")
LAB99 = "ceT1_train/lab/vs_gk_99_t1.nii.gz"
TRAIN = [["ceT1_train/img/vs_gk_99_t1.nii.gz", "fake_data/ceT1-hrT2-ceT1_cc/vs_gk_99_t1.nii.gz",
          "fake_data/ceT1-hrT2-ceT1_ac/vs_gk_99_t1.nii.gz"],
         ["fake_data/ceT1-hrT2_cyc/vs_gk_99_t1.nii.gz", "fake_data/ceT1-hrT2_auxcyc/vs_gk_99_t1.nii.gz"]]
EVAL = [("hrT2_valid/vs_gk_95_t2.nii.gz", "hrT2_valid/vs_gk_95_t2_seg.nii.gz"),
        ("hrT2_test/vs_gk_9_t2.nii.gz", "hrT2_test/vs_gk_9_t2_seg.nii.gz"),
        ("hrT2_train/img/vs_gk_98_t2.nii.gz", "hrT2_train/lab/vs_gk_98_t2.nii.gz")]
TF = {"task": "segmentation", "normalizewithmeanstd_channels": [0], "normalizewithmeanstd_mean": None,
      "normalizewithmeanstd_std": None, "normalizewithmeanstd_mask": False, "normalizewithmeanstd_random_fill": False,
      "normalizewithmeanstd_inverse": False, "pad_output_size": [28, 128, 128], "pad_ceil_mode": False, "pad_inverse": True,
      "randomcrop_output_size": [28, 128, 128], "randomcrop_foreground_focus": True, "randomcrop_foreground_ratio": 0.5,
      "randomcrop_mask_label": [1, 2], "randomcrop_inverse": False, "randomflip_flip_depth": False,
      "randomflip_flip_height": True, "randomflip_flip_width": True, "randomflip_inverse": False,
      "labeltoprobability_class_num": 2, "labeltoprobability_inverse": False}


def loaders(tmp, seed, train_rows):
    """train_rows: per domain a list of (image, label) paths relative to DATA"""
    gen = torch.Generator().manual_seed(seed)
    out = []
    for d in (0, 1):
        csv = os.path.join(tmp, "train_%d_%d.csv" % (d, seed))
        open(csv, "w").write("image,label\n" + "".join("%s,%s\n" % r for r in train_rows[d]))
        tr = Compose(build_transforms(["NormalizeWithMeanStd", "Pad", "RandomCrop", "RandomFlip", "LabelToProbability"], dict(TF)))
        out.append(BatchLoader(NiftyDataset(DATA, csv, 1, True, tr, "cuda:0", cache=True), 4, True, gen))
    return out


def batches(loader):
    """the reference's loader over a 3- (2-) case csv with train_batch_size 4: ONE batch of all cases per epoch (drop_last is
    False, agent_abstract.py:269-281), every case cropped / flipped anew each epoch"""
    while True:
        for b in loader:
            yield b


def run_arm(p, prec, init, seed, iters, tmp, eval_cases, train_rows):
    random.seed(seed)
    torch.manual_seed(seed)
    net = fplx.UNet2D5_dsbn(dict(p, precision=prec))
    net.load_state_dict(init)
    net.cuda()
    net.dropout_seed = 4321
    ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-3, weight_decay=1e-5)
    l0, l1 = loaders(tmp, seed, train_rows)
    g0, g1 = batches(l0), batches(l1)
    losses = []
    for it in range(iters):
        bs = [next(g0), next(g1)]
        outs = ts.step_all([{"image": b["image"], "label_prob": b["label_prob"]} for b in bs])
        losses.append(outs)
    losses = [0.5 * (float(o[0][0].item()) + float(o[1][0].item())) for o in losses]
    net.eval()
    inf = fplx.Inferer(dict(sliding_window_enable=True, sliding_window_size=[28, 128, 128], sliding_window_stride=[28, 128, 128],
                            tta_mode=1, class_num=2))
    dice = []
    with torch.no_grad():
        for img, lab in eval_cases:
            lg = inf.run(net, img, torch.ones(1, dtype=torch.long))
            dice.append(evaluation.binary_dice(fplx.filter.hard_label(lg)[0], lab))
    return losses, 100 * np.asarray(dice, np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--protocol", default="loo", choices=["loo", "uda"])
    ap.add_argument("--seeds", type=int, default=4)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--base", type=int, default=16)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if not os.path.isdir(DATA):
        sys.exit("tools/dice_real.py: %s is missing - run tools/copy_refdata.sh where /root/reference exists" % DATA)
    b = a.base
    p = dict(in_chns=1, feature_chns=[b, 2 * b, 4 * b, 8 * b, 16 * b], dropout=[0, 0, 0.3, 0.4, 0.5], conv_dims=[2, 2, 3, 3, 3],
             class_num=2, bilinear=False, num_domains=2, net_type="UNet2D5_dsbn")
    from fplx.nifti import load_image_as_nd_array
    cases = []
    for f, l in EVAL:                      # test chain of the cfg: NormalizeWithMeanStd, Pad (a no-op on 30/40 x 160 x 272)
        img = torch.from_numpy(np.asarray(load_image_as_nd_array(os.path.join(DATA, f))["data_array"], np.float32)).cuda()
        img = fplx.ops.normalize_mean_std(img.contiguous())
        lab = torch.from_numpy(np.asarray(load_image_as_nd_array(os.path.join(DATA, l))["data_array"][0], np.uint8)).cuda()
        cases.append((img[None], lab))
    src = [[(f, LAB99) for f in TRAIN[0]], [(f, LAB99) for f in TRAIN[1]]]
    if a.protocol == "uda":
        folds = [("all three hrT2 cases", src, cases)]
    else:
        folds = []
        for k in range(len(EVAL)):
            rows1 = src[1] + [EVAL[m] for m in range(len(EVAL)) if m != k]
            folds.append(("held out " + EVAL[k][0], [src[0], rows1], [cases[k]]))
    res = {"fp32": [], "bf16": []}
    lines = ["real data (13 shipped volumes), protocol %s: %d-base UNet2D5_dsbn conv_dims [2,2,3,3,3], dropout [0,0,.3,.4,.5], %d training_all "
             "iterations (crops 28x128x128), sliding window + 4-flip TTA on the test case(s)" % (a.protocol, b, a.iters)]
    with tempfile.TemporaryDirectory() as tmp:
        for fi, (fname, rows, evals) in enumerate(folds):
            for seed in range(a.seeds):
                torch.manual_seed(1000 + 10 * fi + seed)
                init = fplx.UNet2D5_dsbn(dict(p)).state_dict()
                row = []
                for prec in ("fp32", "bf16"):
                    losses, dice = run_arm(p, prec, init, 50 + 10 * fi + seed, a.iters, tmp, evals, rows)
                    res[prec].append(dice)
                    row.append("%s: loss %.4f -> %.4f, Dice %s" % (prec, losses[0], float(np.mean(losses[-5:])), " ".join("%.2f" % v for v in dice)))
                lines.append("%s, seed %d | %s" % (fname, seed, " | ".join(row)))
                print(lines[-1], flush=True)
    r32, r16 = np.asarray(res["fp32"]), np.asarray(res["bf16"])
    lines.append("mean Dice: fp32 %.2f (std over runs %.2f), bf16 %.2f (std %.2f)" % (r32.mean(), r32.mean(1).std(), r16.mean(), r16.mean(1).std()))
    rep, st = paired_report("bf16 - fp32 on real hrT2 (%s)" % a.protocol, r32, r16)
    lines += rep
    text = "\n".join(lines)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()

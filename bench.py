#!/usr/bin/env python3
"""bench.py - FPL+ hot path on MI355X: 3D volumes/sec of the DSBN 3D U-Net train step.

Workload (BASELINE.json configs[1] / SURVEY.md section 8d cfg2): UNet2D5_dsbn all-3D, feature_chns
[32,64,128,256,512], 2 domains, bf16 activations / fp32 master weights, per-GPU batch
2 x 1 x 80 x 160 x 160 synthetic VS-like crops (N(0,1) intensities, ball label), DiceLoss, Adam(1e-4, wd 1e-5).
One step = zero-grad -> forward -> loss -> backward -> (RCCL all-reduce) -> Adam for ONE batch of one
domain; the domain alternates 0/1 per step.  Inputs are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W [--repeats R]
N > 1: one rank per GPU over RCCL, weak scaling (per-GPU batch fixed).  Either the driver launches the ranks
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`) or - when RANK is not in the
environment - this process starts them itself as a child `torch.distributed.run` BEFORE it touches the GPU, relays rank
0's JSON line and exits with the child's code.  Every rank checks WORLD_SIZE == --gpus and fails loudly otherwise.
The K-step timed region (barrier + synchronize on both sides, max over ranks) is run R times (default 5) in the one
command; `ms_per_step` / `value` are those of the MEDIAN region, min / max are reported beside it.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # see fplx/_lib.py: keeps the two backward streams on queues of their own
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory (fplx/_lib.py)

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "fpl-plus_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_BF16_PEAK_TF = 2500.0    # dense bf16 MFMA ~2.5 PFLOP/s
VALU_F32_PEAK_TF = 157.3

NET = dict(in_chns=1, feature_chns=[32, 64, 128, 256, 512], dropout=[0.0, 0.0, 0.3, 0.4, 0.5],
           conv_dims=[3, 3, 3, 3, 3], class_num=2, bilinear=False, num_domains=2, precision="bf16",
           net_type="UNet2D5_dsbn")
SHAPE = (2, 1, 80, 160, 160)


def synth_batch(shape, seed, device):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(shape, generator=g)
    n, _, D, H, W = shape
    zz, yy, xx = torch.meshgrid(torch.arange(D), torch.arange(H), torch.arange(W), indexing="ij")
    lab = torch.zeros((n, 2, D, H, W))
    for i in range(n):
        off = torch.randint(-8, 9, (3,), generator=g)
        m = ((zz - D / 2 - off[0]) ** 2 + (yy - H / 2 - off[1]) ** 2 + (xx - W / 2 - off[2]) ** 2) <= 100
        lab[i, 1][m] = 1.0
        lab[i, 0][~m] = 1.0
    return x.to(device), lab.to(device)


class KernelTimer(object):
    """HIP-event timing of individual launches on the stream they are issued on (torch's current
    stream = the stream fplx launches on).  Wraps selected fplx.ops entry points."""

    def __init__(self, ops):
        self.ops, self.records, self.on = ops, {}, False
        self._orig = {}
        for name in ("conv3d_fwd", "conv3d_wgrad", "conv3d_fwd_cat2", "conv3d_dgrad_split2", "conv3d_wgrad_cat2",
                     "bn_act_fwd", "bn_act_bwd", "deconv2_fwd", "deconv2_dgrad", "deconv2_wgrad", "maxpool2_fwd",
                     "maxpool2_bwd", "seg_loss_fwd", "seg_loss_bwd", "adam_step", "adam_pack_step", "pack_weights_multi",
                     "pack_conv_weights_batched", "outconv_fwd_bn", "outconv_dgrad_bn_bwd", "stem_wgrad_bn", "outconv_wgrad_bn"):
            self._wrap(name)

    def _key(self, name, a, kw={}):
        if name in ("conv3d_fwd",):
            dims, cin, cout, k = a[8], a[9], a[10], a[11]
            return (name, dims, cin, cout, k)
        if name == "conv3d_wgrad":
            dims, cin, cout, k = a[8], a[9], a[10], a[11]
            return (name, dims, cin, cout, k)
        # the split-concat entry points are the same convolutions (same kernels) with two-tensor operands
        if name == "conv3d_fwd_cat2":           # (x0, x1, wp, bias, y, dims, cin, cout, stats)
            return ("conv3d_fwd", a[5], a[6], a[7], (3, 3, 3))
        if name == "conv3d_dgrad_split2":       # (dy, wb, dx0, dx1, dims, cin, cout): conv of cout -> cin channels
            return ("conv3d_fwd", a[4], a[6], a[5], (3, 3, 3))
        if name == "conv3d_wgrad_cat2":         # (x0, x1, dy, dw, dims, cin, cout, ws)
            return ("conv3d_wgrad", a[4], a[5], a[6], (3, 3, 3))
        if name == "bn_act_bwd":                # (y, dout, dy, bnbuf, slope, p, seed, sid, c, ...): voxels x channels of the site
            red = bool(kw.get("reduced", a[15] if len(a) > 15 else False))      # the DownBlock tails: the reduction was fused elsewhere
            app = bool(kw.get("apply", a[16] if len(a) > 16 else True))         # the stem site: the weight gradient forms dy itself
            return (name + ("(apply only)" if red else "") + ("" if app else "(reduce only)"), int(a[0].shape[0]), int(a[8]))
        if name == "bn_act_fwd":                # (y, out, bnbuf, slope, p, seed, sid, c)
            return (name, int(a[0].shape[0]), int(a[7]))
        # the remaining ops carry what their byte model needs (row_model below)
        if name == "deconv2_fwd":               # (x, pack, bias, up, dims_in, cin, cout, pd)
            return (name, tuple(a[4]), int(a[5]), int(a[6]), int(a[7]) if len(a) > 7 else 2)
        if name == "deconv2_dgrad":             # (d_up, packb, d_x, dims_in, cin, cout, pd)
            return (name, tuple(a[3]), int(a[4]), int(a[5]), int(a[6]) if len(a) > 6 else 2)
        if name == "deconv2_wgrad":             # (x, d_up, gw, gb, dims_in, cin, cout, ws, pd)
            return (name, tuple(a[4]), int(a[5]), int(a[6]), int(a[8]) if len(a) > 8 else 2)
        if name == "outconv_fwd_bn":            # (y, bnbuf, slope, a, wf, bias, logits, dims, c0, ncls)
            return (name + ("(logits only)" if a[3] is None else ""), tuple(a[7]), int(a[8]), int(a[9]))
        if name == "outconv_wgrad_bn":          # (y, bnbuf, slope, dlogits, dw, db, dims, c0, ncls, ws)
            return (name, tuple(a[6]), int(a[7]), int(a[8]))
        if name == "outconv_dgrad_bn_bwd":      # (dlogits, wb, y, bnbuf, slope, train, dgamma, dbeta, dslope, part, coef, dy, dims, c0, ncls)
            return (name, tuple(a[12]), int(a[13]), int(a[14]))
        if name == "stem_wgrad_bn":             # (x, y, dout, bnbuf, slope, coef, dw, dims, cin, cout, ws)
            return (name, tuple(a[7]), int(a[8]), int(a[9]))
        if name in ("maxpool2_fwd",):           # (x, pooled, dims, c, pd)
            return (name, tuple(a[2]), int(a[3]), int(a[4]) if len(a) > 4 else 2)
        if name in ("maxpool2_bwd",):           # (skip, d_pool, d_skip, d_x, dims, c, pd)
            return (name, tuple(a[4]), int(a[5]), int(a[6]) if len(a) > 6 else 2)
        if name in ("seg_loss_fwd", "seg_loss_bwd"):          # (logits, label, pw, ...)
            return (name, tuple(a[0].shape), a[2] is not None)
        if name == "adam_step":                 # (p, g, m, v, ...)
            return (name, int(a[0].numel()))
        if name == "adam_pack_step":            # (p, g, m, v, lr, step, wd, gscale, betas, eps, layers): segment + packed elements
            return (name, int(a[0].numel()), int(sum(l[1] * l[2] * 27 * (2 if l[4] is not None else 1) for l in a[10])))
        if name == "pack_weights_multi":        # (jobs): [(kind, w, wf, wb, a, b, taps)]
            return (name, int(sum(j[1].numel() for j in a[0])),
                    int(sum(j[1].numel() * sum(t.element_size() for t in (j[2], j[3]) if t is not None) for j in a[0])))
        if name == "pack_conv_weights_batched":     # (ws, dtype, want_wb, into, stamps, verify)
            verify = bool(kw.get("verify", a[5] if len(a) > 5 else False))
            nel = int(sum(w.numel() for w in a[0]))
            nb = int(sum(w.numel() * (2 if wb else 1) for w, wb in zip(a[0], a[2])))
            return (name + ("(verify)" if verify else ""), nel, nb)
        return (name,)

    def _wrap(self, name):
        orig = getattr(self.ops, name)
        self._orig[name] = orig

        def f(*a, **kw):
            if not self.on:
                return orig(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **kw)
            e1.record()
            self.records.setdefault(self._key(name, a, kw), []).append((e0, e1))
            return r
        setattr(self.ops, name, f)

    def summary(self):
        out = {}
        for k, evs in self.records.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[k] = (len(ms), float(np.mean(ms)), float(np.sum(ms)))
        return out


def roofline_of(key, n_avg_ms):
    """algorithmic bytes / flops per launch of a convolution key (bf16 activations, SURVEY 8d rule:
    every input read once, every output written once)."""
    name, dims, cin, cout, k = key
    n, d, h, w = dims
    vox = n * d * h * w
    taps = k[0] * k[1] * k[2]
    flops = 2.0 * vox * cin * cout * taps
    if name == "conv3d_fwd":
        nbytes = vox * (cin + cout) * 2.0
    else:  # wgrad reads x and dy, writes the tiny dW
        nbytes = vox * (cin + cout) * 2.0 + taps * cin * cout * 4.0
    sec = n_avg_ms * 1e-3
    return flops, nbytes, flops / sec / 1e12, nbytes / sec / 1e9


def row_model(key):
    """-> (bound, algorithmic bytes, algorithmic flops) of one row of the per-kernel table, or None.  Byte models: DESIGN
    section 4's per-unit figures (SURVEY 8d rule: every input read once, every output written once; bf16 activations 2 B,
    fp32 network input / logits 4 B); a row is judged against the MFMA peak when its arithmetic intensity exceeds the ridge
    (peak FLOP/s over peak bytes/s), else against HBM."""
    name = key[0]
    flops = 0.0
    if name in ("conv3d_fwd", "conv3d_wgrad"):
        _, dims, cin, cout, k = key
        n, d, h, w = dims
        vox = n * d * h * w
        taps = k[0] * k[1] * k[2]
        flops = 2.0 * vox * cin * cout * taps
        if tuple(k) == (1, 3, 3):                # out_conv: one side is the fp32 planar logits / their gradient
            narrow, wide = min(cin, cout), max(cin, cout)
            nbytes = vox * (wide * 2.0 + narrow * 4.0)
        elif cin <= 4 and name == "conv3d_fwd":  # stem forward: fp32 network input
            nbytes = vox * (cin * 4.0 + cout * 2.0)
        elif cin <= 4:                           # stem weight gradient: fp32 input + bf16 dy
            nbytes = vox * (cin * 4.0 + cout * 2.0)
        else:
            nbytes = vox * (cin + cout) * 2.0
        if name == "conv3d_wgrad":
            nbytes += taps * cin * cout * 4.0
    elif name in ("bn_act_bwd", "bn_act_bwd(apply only)", "bn_act_bwd(reduce only)", "bn_act_fwd"):
        tensors = {"bn_act_bwd": 5.0, "bn_act_bwd(apply only)": 3.0, "bn_act_bwd(reduce only)": 2.0, "bn_act_fwd": 2.0}[name]
        nbytes = tensors * key[1] * key[2] * 2.0
    elif name == "stem_wgrad_bn":                # fp32 input + y + d(a) read, dw written: the apply pass's dy never exists
        _, dims, cin, cout = key
        vox = dims[0] * dims[1] * dims[2] * dims[3]
        flops = 2.0 * vox * cin * cout * 27
        nbytes = vox * (cin * 4.0 + 2 * cout * 2.0) + 27 * cin * cout * 4.0
    elif name in ("deconv2_fwd", "deconv2_dgrad", "deconv2_wgrad"):
        _, dims, cin, cout, pd = key
        vox = dims[0] * dims[1] * dims[2] * dims[3]
        kids = 4 * pd                            # children per input voxel (2 x 2 x 2, or 1 x 2 x 2 per depth slice)
        flops = 2.0 * vox * kids * cin * cout
        nbytes = vox * (cin + kids * cout) * 2.0 + (kids * cin * cout * 4.0 if name == "deconv2_wgrad" else 0.0)
    elif name == "outconv_fwd_bn":
        _, dims, c0, ncls = key
        vox = dims[0] * dims[1] * dims[2] * dims[3]
        flops, nbytes = 2.0 * vox * 9 * c0 * ncls, vox * (4.0 * c0 + 4.0 * ncls)
    elif name == "outconv_fwd_bn(logits only)":  # y read, logits written: the activation is not stored
        _, dims, c0, ncls = key
        vox = dims[0] * dims[1] * dims[2] * dims[3]
        flops, nbytes = 2.0 * vox * 9 * c0 * ncls, vox * (2.0 * c0 + 4.0 * ncls)
    elif name == "outconv_wgrad_bn":             # y + dlogits read
        _, dims, c0, ncls = key
        vox = dims[0] * dims[1] * dims[2] * dims[3]
        flops, nbytes = 2.0 * vox * 9 * c0 * ncls, vox * (2.0 * c0 + 4.0 * ncls) + 9 * c0 * ncls * 4.0
    elif name == "outconv_dgrad_bn_bwd":
        _, dims, c0, ncls = key
        vox = dims[0] * dims[1] * dims[2] * dims[3]
        flops, nbytes = 2.0 * 2.0 * vox * 9 * c0 * ncls, vox * (6.0 * c0 + 8.0 * ncls)       # the data gradient is formed twice
    elif name == "maxpool2_fwd":
        _, dims, c, pd = key
        vox = dims[0] * dims[1] * dims[2] * dims[3]
        nbytes = vox * c * 2.0 * (1.0 + 1.0 / (4 * pd))
    elif name == "maxpool2_bwd":
        _, dims, c, pd = key
        vox = dims[0] * dims[1] * dims[2] * dims[3]
        nbytes = vox * c * 2.0 * (3.0 + 1.0 / (4 * pd))                 # skip tensor, skip gradient in, result out + pooled gradient
    elif name == "seg_loss_fwd":
        shp, has_pw = key[1], key[2]
        vox = shp[0] * shp[2] * shp[3] * shp[4]
        nbytes = vox * (8.0 * shp[1] + (4.0 if has_pw else 0.0))        # logits + one-hot label (+ pixel weight), fp32
    elif name == "seg_loss_bwd":
        shp, has_pw = key[1], key[2]
        vox = shp[0] * shp[2] * shp[3] * shp[4]
        nbytes = vox * (12.0 * shp[1] + (4.0 if has_pw else 0.0))
    elif name == "adam_step":
        nbytes = 28.0 * key[1]
    elif name == "adam_pack_step":               # p, g, m, v read + p, m, v written (28 B / element) + the bf16 packs written
        nbytes = 28.0 * key[1] + 2.0 * key[2]
    elif name == "pack_weights_multi":           # fp32 masters read, packs written (key[2] already in bytes)
        nbytes = 4.0 * key[1] + 1.0 * key[2]
    elif name == "pack_conv_weights_batched":    # fp32 masters read, bf16 packs written
        nbytes = 4.0 * key[1] + 2.0 * key[2]
    elif name == "pack_conv_weights_batched(verify)":     # 32 of every tile's 13 824 masters + as many stamped floats
        nbytes = 8.0 * key[1] * 32.0 / 13824.0
    else:
        return None
    ridge = MFMA_BF16_PEAK_TF * 1e3 / HBM_PEAK_GBS
    return ("mfma" if flops / nbytes >= ridge else "hbm"), nbytes, flops


PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_hbm_traffic.json")


def pmc_traffic(key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (profiles/r06_pmc_hbm_traffic.json,
    made by tools/pmc_summary.py from separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of THIS command with the gfx950
    x2 FETCH correction).  The summary records the sha256 of the kernel sources it was measured on: if conv_march.hip or
    common.h have changed since, the figure is stale and None is reported.  None also if there is no matching entry."""
    import hashlib
    path = PMC_FILE
    if not os.path.exists(path):
        return None
    meta = json.load(open(path)).get("_meta", {}).get("sources_sha256", {})
    for f in ("conv_march.hip", "common.h"):
        src = os.path.join(ROOT, "fpl-plus_amd", "csrc", f)
        if not os.path.exists(src) or hashlib.sha256(open(src, "rb").read()).hexdigest() != meta.get(f):
            return None
    name, dims, cin, cout, k = key
    n, d, h, w = dims
    if not (name == "conv3d_fwd" and k == (3, 3, 3) and cin in (32, 64) and cout % 32 == 0 and h >= 16 and w >= 64):
        return None
    # launch geometry of the depth-march kernels (fpl-plus_amd/csrc/conv_march.hip: march_cfg): Cin = 32 interior shapes run
    # conv_fwd_march32v2 / v3 (4 waves), ragged ones conv_fwd_march32 (8 waves); Cin = 64 conv_fwd_march64 (4 waves)
    interior = cin == 32 and h % 16 == 0 and w % 32 == 0
    fh, threads = (16, 256 if interior else 512) if cin == 32 else (8, 256)
    if h < fh:
        return None
    tiles_h, tiles_w = (h + fh - 1) // fh, (w + 31) // 32
    tiles = n * tiles_h * tiles_w * (cout // 32)
    best, segs_best = None, 1
    for ds in range(1, d + 1):
        dl = (d + ds - 1) // ds
        if dl < 4 and ds > 1:
            break
        segs = (d + dl - 1) // dl
        if d - (segs - 1) * dl < 2 and d >= 2:
            continue
        cost = ((tiles * segs + 255) // 256) * (dl + 2 + 1.5)
        if best is None or cost < best - 1e-9:
            best, segs_best = cost, segs
    dlen = (d + segs_best - 1) // segs_best
    dsegs = (d + dlen - 1) // dlen
    grid = n * tiles_h * tiles_w * dsegs * (cout // 32) * threads
    tab = json.load(open(path))
    hits = [e for k, e in tab.items() if k != "_meta" and k.startswith("conv_fwd_march%d" % cin) and k.endswith("|grid=%d" % grid)]
    if not hits:
        return None
    # forward (statistics) and data-gradient launches of one shape are different instantiations: launch-weighted mean
    return sum(e["hbm_bytes_per_launch"] * e["launches"] for e in hits) / sum(e["launches"] for e in hits)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline():
    """The oracle (PyTorch-CPU restatement of the reference modules, oracle/torch_ref.py; `kind: port`) timed on the host
    cores on a bounded sample of the SAME workload, fp32 as the reference runs it: a few train steps (forward, Dice loss,
    backward, Adam) on a 1 x 1 x 32 x 80 x 80 crop (1/10 volume) to size the run, then ONE full step at the benchmark
    shape 2 x 1 x 80 x 160 x 160 (about 20 s on the GPU box's 64 host cores; skipped, and the crop figure extrapolated,
    when the crop says it would take more than 90 s)."""
    from oracle import torch_ref as R
    import detdata
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))      # oversubscribing a cgroup-limited box makes ATen crawl
    torch.set_num_threads(cores)
    p = dict(NET)
    p["dropout"] = [0, 0, 0, 0, 0]
    loss_fn = R.loss_from_config({"loss_type": "DiceLoss"})

    def steps(shape, count, budget):
        sd, prm = R.split_state(detdata.state_dict_3d(p))
        opt = R.AdamRef(prm, 1e-4, 1e-5)
        x = torch.randn(shape)
        lab = torch.from_numpy(detdata.ball_label(shape[2:], 6.0 if shape[2] < 80 else 10.0, n=shape[0]))
        times, t_start = [], time.time()
        for it in range(count):
            t0 = time.time()
            R.training_all_step(sd, prm, opt, p, [{"image": x, "label_prob": lab}], loss_fn)
            times.append(time.time() - t0)
            if time.time() - t_start > budget:
                break
        return times

    crop = steps((1, 1, 32, 80, 80), 4, 8.0)
    t_crop = float(np.median(crop[1:])) if len(crop) > 1 else crop[0]
    est_full = t_crop * 10.0 * SHAPE[0]
    note = "crop 1x1x32x80x80 (1/10 volume): %.2f s/step" % t_crop
    if est_full <= 45.0:
        # two full steps: the first one pays oneDNN primitive creation, first-touch page faults and allocator growth for the
        # full-size tensors; the SECOND is the baseline (an un-warmed single step understated the CPU)
        full = steps(SHAPE, 2, 1e9)
        t_full = min(full[1:]) if len(full) > 1 else full[0]
        value = SHAPE[0] / t_full
        sample = ("full fp32 train steps at the benchmark shape %dx1x80x160x160: %s s, the fastest after the first is the "
                  "baseline; %s" % (SHAPE[0], ", ".join("%.1f" % t for t in full), note))
    elif est_full <= 90.0:
        t_full = steps(SHAPE, 1, 0.0)[0]
        value = SHAPE[0] / t_full
        sample = ("ONE full fp32 train step at the benchmark shape %dx1x80x160x160 (no warm-up: a lower bound of the CPU "
                  "rate), %.1f s; %s" % (SHAPE[0], t_full, note))
    else:
        value = (1.0 / 10.0) / t_crop
        sample = "extrapolated from the " + note + " (a full step would take about %.0f s here)" % est_full
    return {"value": value, "unit": "volumes/s", "cores": cores, "kind": "port", "cpu": _cpu_model(),
            "sample": "32-base UNet-DSBN, DiceLoss + Adam, oracle/torch_ref.py, %d threads: %s" % (cores, sample)}


def rccl_report(ts, dev, rank, world):
    """The `rccl` object of a data-parallel bench line (BASELINE.md section 3 asks for the measured bus
    bandwidth as the denominator of the scaling figure), measured BEFORE the timed region on the step's own buffers and bucket
    ranges, HIP events on the stream the collectives are enqueued from:
      world / backend / version      what torch.distributed reports (backend "nccl" = RCCL on ROCm)
      devices                        per rank: device name, PCI bus id, uuid - gathered to rank 0, so the record shows that
                                     RCCL really saw N different GPUs
      allreduce_flat_ms              ONE all-reduce(sum) of the whole flat fp32 gradient buffer (90 MB for the 32-base network)
      allreduce_buckets_ms           the step's own sequence: every bucket of TrainStep's reducer + one BatchNorm segment,
                                     launched asynchronously back to back as backward does, then waited for
      busbw_GBps                     ring bus bandwidth of the flat all-reduce: 2 (N - 1) / N x bytes / time (0 at N = 1:
                                     nothing crosses a link; algbw_GBps = bytes / time is reported beside it)"""
    g = ts.gflat
    nbytes = g.numel() * 4

    def timed(fn, warm=2, reps=5):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) / reps], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def buckets():
        works = [dist.all_reduce(g[s:e], op=dist.ReduceOp.SUM, async_op=True) for s, e in ts.reducer.buckets]
        s, e = ts.reducer.domain_ranges[0]
        works.append(dist.all_reduce(g[s:e], op=dist.ReduceOp.SUM, async_op=True))
        for w in works:
            w.wait()

    g.zero_()                                            # (sums of zeros: the measurement leaves the buffer as the step expects it)
    t_flat = timed(lambda: dist.all_reduce(g, op=dist.ReduceOp.SUM))
    t_b = timed(buckets)
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "device": props.name,
            "pci_bus_id": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                                              getattr(props, "pci_device_id", 0)),
            "uuid": str(getattr(props, "uuid", "")), "cuda_index": dev.index}
    devs = [None] * world
    dist.all_gather_object(devs, mine)
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:                               # noqa: BLE001
        ver = "unknown (%s)" % type(e).__name__
    return {"world": world, "backend": dist.get_backend(), "version": ver, "devices": devs,
            "flat_gradient_MB": round(nbytes / 1e6, 2), "buckets": len(ts.reducer.buckets) + 1,
            "bucket_MB": [round((e - s) * 4 / 1e6, 2) for s, e in ts.reducer.buckets],
            "allreduce_flat_ms": round(t_flat, 4), "allreduce_90MB_ms": round(t_flat * 90e6 / nbytes, 4),
            "allreduce_buckets_ms": round(t_b, 4),
            "algbw_GBps": round(nbytes / (t_flat * 1e-3) / 1e9, 2),
            "busbw_GBps": round(2.0 * (world - 1) / world * nbytes / (t_flat * 1e-3) / 1e9, 2),
            "distinct_devices": len({d["pci_bus_id"] + d["uuid"] for d in devs})}


SECONDARY_PARTS = ("plugin_path", "config4", "config5_one_gpu")


def _timed(fn, warm, reps):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def secondary_part(name, dev):
    """One of the measurements reported beside the headline (`secondary`, OUTSIDE the timed region; SURVEY 8d asks for configs 4
    and 5, VERDICT r03 for the plugin path), run in a process of its own - see secondary().  ms per unit on this GPU.
      plugin_path     : SegmentationAgent.training_all (agent_seg.py:415-508) at the headline shape - one iteration = both
                        domains + one Adam step, iter_valid = 8 iterations per call as a training round has many - as the agent
                        runs it (routed through fplx.TrainStep), in its autograd route (net(x), loss module, loss.backward(),
                        FusedAdam.step()), and TrainStep.step_all on the same batches
      config4         : pseudo-label selection on one 1 x 1 x 48 x 160 x 272 volume (eval-mode BatchNorm, test-time dropout):
                        T = 4 Monte-Carlo forwards + filter, the reference-literal 6 passes x 4-flip TTA (24 forwards), and the
                        filter kernel's rate on the T = 6 stack
      config5_one_gpu : one training_all iteration of the 4-channel 2 x 4 x 128^3-per-domain network, 0.5 Dice + 0.5 CE with
                        pixel weights, ONE GPU (the 8-GPU half needs a node)"""
    import fplx
    from fplx import ops
    from fplx.infer import Inferer
    if name == "plugin_path":
        batches = [synth_batch(SHAPE, 7 + d, dev) for d in (0, 1)]
        iv = 8
        tcfg = {"dis": False, "train_fpl_uda": True, "loss_type": "DiceLoss", "optimizer": "Adam", "learning_rate": 1e-4,
                "momentum": 0.9, "weight_decay": 1e-5, "lr_scheduler": "MultiStepLR", "lr_gamma": 0.5,
                "lr_milestones": [10000, 20000], "iter_valid": iv, "gpus": [dev.index or 0]}
        agent = fplx.SegmentationAgent({"dataset": {"tensor_type": "float"}, "network": dict(NET), "training": tcfg, "testing": {}},
                                       "train")
        torch.manual_seed(1)
        agent.create_network()
        agent.create_optimizer()
        agent.create_loss_calculator()
        agent.set_loaders([{"image": batches[0][0], "label_prob": batches[0][1]}], [{"image": batches[1][0], "label_prob": batches[1][1]}])
        t_agent = _timed(agent.training_all, 1, 2) / iv
        agent.engine_mode = False
        t_autograd = _timed(agent.training_all, 1, 2) / iv
        ts = fplx.TrainStep(agent.net, (1.0, 0.0, 0.0, 0.0), True, optimizer=agent.optimizer)
        bl = [{"image": b[0], "label_prob": b[1]} for b in batches]
        t_engine = _timed(lambda: ts.step_all(bl), 4, 2 * iv)
        return {"shape": list(SHAPE), "unit": "ms per training_all iteration (two domains, one Adam step)",
                "SegmentationAgent.training_all": round(t_agent, 3),
                "SegmentationAgent.training_all, autograd route (engine_mode = False)": round(t_autograd, 3),
                "TrainStep.step_all": round(t_engine, 3), "agent_over_engine": round(t_agent / t_engine, 4)}
    if name == "config4":
        net = fplx.UNet2D5_dsbn(dict(NET)).to(dev)
        net.eval()
        for m in net.modules():
            if type(m) == torch.nn.Dropout:
                m.train()
        x = torch.randn(1, 1, 48, 160, 272, device=dev)
        dl = torch.ones(1, dtype=torch.long)
        c4 = {"volume": [1, 1, 48, 160, 272]}
        stack = None
        for T, tta, key in ((4, 0, "T4_ms_per_volume"), (6, 1, "T6x4flip_24_forwards_ms_per_volume")):
            inf = Inferer(dict(class_num=2, tta_mode=tta, infer_batch_voxels=1 << 23))

            def mc():
                with torch.no_grad():
                    st = inf.run_mc(net, x, dl, T)[:, 0]
                return st, ops.mc_filter(st, 0.01)
            c4[key] = round(_timed(mc, 2, 4), 3)
            stack = mc()[0]
        ms = _timed(lambda: ops.mc_filter(stack, 0.01), 3, 20)
        c4["mc_filter_ms"] = round(ms, 4)
        c4["mc_filter_GBps_of_logits"] = round(stack.numel() * 4 / ms / 1e6, 1)
        return c4
    if name == "config5_one_gpu":
        torch.manual_seed(1)
        net = fplx.UNet2D5_dsbn(dict(NET, in_chns=4)).to(dev)
        loss = fplx.make_loss({"loss_type": ["DiceLoss", "CrossEntropyLoss"], "loss_weight": [0.5, 0.5]})
        ts = fplx.TrainStep(net, loss.terms, True, lr=1e-4, weight_decay=1e-5)
        g = torch.Generator().manual_seed(0)
        b5 = []
        for d in range(2):
            xx = torch.randn(2, 4, 128, 128, 128, generator=g)
            lab = torch.zeros(2, 2, 128, 128, 128)
            lab[:, 0] = 1.0
            lab[:, 0, 40:90, 30:100, 50:110] = 0.0
            lab[:, 1, 40:90, 30:100, 50:110] = 1.0
            w_img = torch.rand(2, generator=g) + 0.01
            pw = (torch.rand(2, 1, 128, 128, 128, generator=g) > 0.1).float() * w_img.view(2, 1, 1, 1, 1)
            b5.append({"image": xx.to(dev), "label_prob": lab.to(dev), "pixel_weight": pw.to(dev), "image_weight": w_img.to(dev)})
        t5 = _timed(lambda: ts.step_all(b5), 4, 10)
        return {"batch": "2 x 4 x 128^3 per domain, two domains", "ms_per_training_all_iteration": round(t5, 3),
                "crops_per_s": round(4 / t5 * 1e3, 2)}
    raise ValueError(name)


def secondary():
    """The `secondary` object of the bench line: every part in a CHILD process of its own, after the headline (measured in THIS
    process's fresh address space: a network built after others were freed ran up to 35 % slower here whatever its kernels -
    tools/step_ab.py saw the same with a fourth live network - so the parts do not share a process either).  A failing part
    leaves an `error` entry; the headline is never affected."""
    import subprocess
    out = {}
    for name in SECONDARY_PARTS:
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--secondary-part", name], stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, text=True, timeout=180)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            out[name] = json.loads(lines[-1]) if r.returncode == 0 and lines else {"error": "rc %d: %s" % (r.returncode, r.stderr[-300:])}
        except Exception as e:
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def launch_ranks(n, timeout_s=1500.0, attempts=3):
    """`bench.py --gpus N` without a launcher: start N ranks as a child `python -m torch.distributed.run` (this process has
    not initialised the GPU and never will), relay rank 0's JSON line, return the child's exit code.  The child runs in a
    process group of its own: when it exceeds `timeout_s` (a rank hung in RCCL initialisation, say) the whole group is
    killed and the exit code is non-zero.  A rendezvous port that was taken between the probe and the bind gets a FRESH
    child on a fresh port (never a re-exec of a process that touched the GPU)."""
    import signal
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rc, out = 1, ""
    for attempt in range(attempts):
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, err = proc.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            out, err = proc.communicate()
            sys.stderr.write("bench.py: the %d-rank child exceeded %.0f s and was killed (process group %d)\n%s\n"
                             % (n, timeout_s, proc.pid, (err or "")[-2000:]))
            return 124
        rc = proc.returncode
        sys.stderr.write(err or "")
        if rc != 0 and ("EADDRINUSE" in (err or "") or "address already in use" in (err or "").lower()):
            continue                                     # the port was taken after the probe: a fresh child, a fresh port
        break
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    if rc == 0 and lines:
        print(lines[-1])                                 # rank 0 prints the line once; the last one if a rank echoed it
        return 0
    sys.stderr.write("bench.py: the %d-rank child exited with code %d and printed %d result line(s)\n%s\n"
                     % (n, rc, len(lines), out[-2000:]))
    return rc or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--launch", action="store_true", help="start the ranks as a child torch.distributed.run even for --gpus 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` measurements (plugin path, configs 4 and 5)")
    ap.add_argument("--secondary-part", choices=SECONDARY_PARTS, help="(internal) run ONE secondary measurement and print its JSON")
    args = ap.parse_args()
    if args.secondary_part:
        dev0 = torch.device("cuda", 0)
        torch.cuda.set_device(dev0)
        print(json.dumps(secondary_part(args.secondary_part, dev0)))
        return
    if args.gpus < 1 or args.steps < 1 or args.repeats < 1 or args.warmup < 0:
        ap.error("--gpus, --steps, --repeats must be >= 1 and --warmup >= 0")

    if (args.gpus > 1 or args.launch) and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))            # nothing has touched the GPU yet: the ranks are CHILD processes

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d - refusing to print a line for the wrong world size\n"
                         % (args.gpus, world))
        sys.exit(2)
    use_dist = world > 1 or (os.environ.get("FPLX_DDP_FORCE", "0") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import fplx
    from fplx import ops
    torch.manual_seed(1)
    net = fplx.UNet2D5_dsbn(dict(NET)).to(dev)
    net._ensure_flat()
    fplx.ddp.broadcast_params_from_rank0(net)
    ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-4, weight_decay=1e-5,
                        milestones=[10000, 20000, 30000, 40000], gamma=0.5)
    batches = [synth_batch(SHAPE, 1000 * rank + d, dev) for d in (0, 1)]
    timer = None if args.no_kernel_timing else KernelTimer(ops)

    def run(k0, k):
        out = None
        for i in range(k0, k0 + k):
            d = i % 2
            out = ts.step(batches[d][0], batches[d][1], d)
        return out

    run(0, args.warmup)
    rccl = rccl_report(ts, dev, rank, world) if use_dist else None       # before (and outside) the timed region

    def timed_region(k0_):
        """EXACTLY --steps steps between barrier + synchronize pairs; -> (seconds on this rank, last step's output)"""
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        o_ = run(k0_, args.steps)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0_, o_

    regions, k0, out = [], args.warmup, None
    for _ in range(args.repeats):                    # each region: EXACTLY --steps steps between barrier + synchronize pairs
        t_reg, out = timed_region(k0)
        regions.append(t_reg)
        k0 += args.steps
    if use_dist:                                     # a region's time = the slowest rank's
        tt = torch.tensor(regions, dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        regions = [float(v) for v in tt.tolist()]
    dt = float(np.median(regions))
    loss = float(out[0].item())
    # exposed communication: the same regions with every collective of the step switched off - no gradient
    # buckets, no all-reduce of the loss sums (each rank then differentiates its LOCAL loss: a timing probe, its results are
    # discarded) - AFTER the headline regions; exposed_comm_ms = headline step - this step
    exposed = None
    if use_dist:
        red_on, loss_on = ts.reducer.enabled, ts.dist_loss
        ts.reducer.enabled, ts.dist_loss = False, False
        run(k0, 2)
        quiet = []
        for _ in range(min(args.repeats, 3)):
            t_reg, _o = timed_region(k0)
            quiet.append(t_reg)
        ts.reducer.enabled, ts.dist_loss = red_on, loss_on
        tq = torch.tensor(quiet, dtype=torch.float64, device=dev)
        dist.all_reduce(tq, op=dist.ReduceOp.MAX)
        dq = float(np.median(tq.tolist()))
        exposed = {"ms_per_step_without_collectives": round(dq / args.steps * 1e3, 3),
                   "exposed_comm_ms": round((dt - dq) / args.steps * 1e3, 3),
                   "note": "same process, after the headline regions: gradient buckets and the loss sums' all-reduce off"}

    # per-kernel timing pass (separate, short, so the event records do not perturb the headline number)
    roof = None
    roof_hbm = None
    roof_min = None
    ktable = []
    if timer is not None:
        timer.on = rank == 0            # every rank runs the pass (it contains the all-reduce), rank 0 records
        side = net.engine.use_side_stream
        net.engine.use_side_stream = False      # one kernel at a time: clean per-launch durations
        run(k0, 2)
        torch.cuda.synchronize()
        net.engine.use_side_stream = side
        timer.on = False
    if timer is not None and rank == 0:
        summ = timer.summary()
        tot = sum(v[2] for v in summ.values())
        for k, (cnt, avg, s) in sorted(summ.items(), key=lambda kv: -kv[1][2]):
            row = {"kernel": str(k), "launches": cnt, "avg_ms": round(avg, 4), "share": round(s / tot, 4)}
            m = row_model(k)
            if m is not None:            # every row against the roofline that bounds it
                bound, nb, fl = m
                ach = fl / (avg * 1e-3) / 1e12 if bound == "mfma" else nb / (avg * 1e-3) / 1e9
                row.update({"bound": bound, "frac": round(ach / (MFMA_BF16_PEAK_TF if bound == "mfma" else HBM_PEAK_GBS), 4)})
            ktable.append(row)
        fr = [r_ for r_ in ktable if "frac" in r_ and r_["share"] >= 0.01]
        if fr:
            roof_min = dict(min(fr, key=lambda r_: r_["frac"]))
            roof_min["note"] = "the lowest-fraction row with a time share of at least 1 % (kernels alone, second stream off)"
        # the largest HBM-bound consumer beside the dominant convolution (the BatchNorm backward passes are the
        # biggest group of the table): bn_act_bwd = reduce (reads d(out), y) + finalize + apply (reads both, writes dy) =
        # 5 tensors of voxels x C bf16; bn_act_fwd = 2
        bns = {k: v for k, v in summ.items() if k[0] in ("bn_act_bwd", "bn_act_fwd")}       # (not the apply-only / reduce-only forms)
        if bns:
            kb = max(bns, key=lambda kk: bns[kk][2])
            nb = (5.0 if kb[0] == "bn_act_bwd" else 2.0) * kb[1] * kb[2] * 2.0
            gb = nb / (bns[kb][1] * 1e-3) / 1e9
            roof_hbm = {"bound": "hbm", "achieved": round(gb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gb / HBM_PEAK_GBS, 5), "kernel": str(kb), "avg_ms": round(bns[kb][1], 4),
                        "launches_timed": bns[kb][0], "algorithmic_bytes": nb, "time_share": round(bns[kb][2] / tot, 4),
                        "note": "three launches (reduce, finalize, apply) timed as one op, second stream off"}
        convs = {k: v for k, v in summ.items() if k[0] in ("conv3d_fwd", "conv3d_wgrad")}
        if convs:
            k = max(convs, key=lambda kk: convs[kk][2])
            flops, nbytes, tf, gbs = roofline_of(k, convs[k][1])
            ai = flops / nbytes
            ridge = MFMA_BF16_PEAK_TF * 1e3 / HBM_PEAK_GBS
            if ai >= ridge:
                roof = {"bound": "mfma", "achieved": round(tf, 3), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                        "frac": round(tf / MFMA_BF16_PEAK_TF, 5), "traffic": None}
            else:
                roof = {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": None}
            roof["traffic"] = pmc_traffic(k)
            roof.update({"kernel": str(k), "avg_ms": round(convs[k][1], 4), "launches_timed": convs[k][0],
                         "algorithmic_bytes": nbytes, "algorithmic_flops": flops,
                         "hbm_GBps": round(gbs, 2), "TFLOPps": round(tf, 3), "time_share": round(convs[k][2] / tot, 4)})

    if rank == 0:
        vols = world * SHAPE[0] * args.steps
        res = {
            "metric": "3D volumes/sec (train step, 80x160x160)", "value": round(vols / dt, 4), "unit": "volumes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "repeats": args.repeats, "ms_per_step_min": round(min(regions) / args.steps * 1e3, 3),
            "ms_per_step_max": round(max(regions) / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "UNet3D-DSBN 32-base-ch bf16 (fp32 master weights), per-GPU batch 2x1x80x160x160 "
                                   "synthetic VS crops, DiceLoss + Adam, domain alternating per step",
                       "global_batch": world * SHAPE[0], "parallelism": "dp%d" % world,
                       "step": "zero_grad+forward+loss+backward+allreduce+adam"},
            "final_loss": round(loss, 6),
            # whole-step roofline: SURVEY 8d analytic work per volume (train): 2.793 TFLOP, 10.1 GB
            "step_roofline": {"TFLOPps": round(2.793 * vols / dt / world, 2), "hbm_GBps": round(10.1 * vols / dt / world, 1),
                              "frac_mfma": round(2.793 * vols / dt / world / MFMA_BF16_PEAK_TF, 5),
                              "frac_hbm": round(10.1 * vols / dt / world / HBM_PEAK_GBS, 5)},
        }
        if rccl is not None:
            res["rccl"] = rccl
            res["exposed_comm_ms"] = exposed["exposed_comm_ms"]
            res["comm"] = exposed
        if roof is not None:
            res["roofline"] = roof
        if roof_hbm is not None:
            res["roofline_hbm"] = roof_hbm
        if roof_min is not None:
            res["roofline_min"] = roof_min
        if ktable:
            res["kernels"] = ktable[:40]
        if world == 1 and not args.no_secondary:
            res["secondary"] = secondary()
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

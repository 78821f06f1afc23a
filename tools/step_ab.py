"""A/B of the benchmark's train step under tuning-knob settings, alternating in ONE process on one box (the +-3 % between boxes
is larger than most kernel wins): ONE network + TrainStep, the knobs (and engine attributes) are flipped between the
interleaved rounds (workspaces grow on first use: two untimed steps follow every flip), the median of the rounds is printed
per setting with min / max.

    python tools/step_ab.py "wg_roll=0" "wg_roll=1" ["wg_roll=1,wg_roll_geo=1" ...] [--steps 20] [--rounds 5]
    engine attributes are set with a leading '@':  "@use_side_stream=0";  TrainStep attributes with '%':  "%adam_overlap=0"
    "!fplx_name=1" SKIPS every call of that C entry point (an upper bound for what removing a launch could return: the step's
    results are then garbage - timing probes only)
"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "fpl-plus_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory (fplx/_lib.py)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fplx  # noqa: E402
from fplx import _lib  # noqa: E402
import bench  # noqa: E402


def parse(arg):
    knobs, attrs = {}, {}
    for item in arg.split(","):
        item = item.strip()
        if not item or item == "default":
            continue
        k, v = item.split("=")
        if k.startswith("@") or k.startswith("%") or k.startswith("!"):
            attrs[k] = int(v)
        else:
            knobs[k] = int(v)
    return knobs, attrs


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    steps, rounds = int(opts.get("steps", 20)), int(opts.get("rounds", 5))
    hi = None
    if int(opts.get("main-prio", 0)):           # --main-prio=1: the step runs on a high-priority stream (the second stream stays normal)
        hi = torch.cuda.Stream(priority=-1)
    settings = [(a, ) + parse(a) for a in args] or [("default", {}, {})]
    keys = sorted({k for _, kn, _ in settings for k in kn})
    defaults = {k: _lib.get_tuning(k) for k in keys}
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    batches = [bench.synth_batch(bench.SHAPE, d, dev) for d in (0, 1)]

    def apply(kn):
        for k in keys:
            _lib.set_tuning(k, kn.get(k, defaults[k]))

    torch.manual_seed(1)
    net = fplx.UNet2D5_dsbn(dict(bench.NET)).to(dev)
    net._ensure_flat()
    ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-4, weight_decay=1e-5, milestones=[10000, 20000], gamma=0.5)
    skipped = set()
    from fplx import ops
    real_call = ops.call
    ops.call = lambda name, *a: None if name in skipped else real_call(name, *a)
    skip_keys = sorted({k for _, _, at in settings for k in at if k[0] == "!"})
    attr_keys = sorted({k for _, _, at in settings for k in at if k[0] != "!"})
    owner = lambda k: net.engine if k[0] == "@" else ts
    attr_def = {k: getattr(owner(k), k[1:]) for k in attr_keys}
    runs = [(name, kn, at, []) for name, kn, at in settings]
    import contextlib
    ctx = torch.cuda.stream(hi) if hi is not None else contextlib.nullcontext()
    with ctx:
      for r in range(rounds):
          for name, kn, at, res in runs:
              apply(kn)
              net.engine.invalidate()                     # packs / plans cached under the previous knobs
              for k in attr_keys:
                  setattr(owner(k), k[1:], type(attr_def[k])(at[k]) if k in at else attr_def[k])
              skipped.clear()
              skipped.update(k[1:] for k in skip_keys if at.get(k, 0))
              for i in range(3 if r else 6):
                  ts.step(batches[i % 2][0], batches[i % 2][1], i % 2)
              torch.cuda.synchronize()
              t0 = time.perf_counter()
              for i in range(steps):
                  ts.step(batches[i % 2][0], batches[i % 2][1], i % 2)
              torch.cuda.synchronize()
              res.append((time.perf_counter() - t0) / steps * 1e3)
    apply({})
    for name, kn, at, res in runs:
        print("%-48s %7.3f ms/step (min %.3f, max %.3f over %d rounds of %d steps)  %.1f vol/s" %
              (name, float(np.median(res)), min(res), max(res), rounds, steps, bench.SHAPE[0] / float(np.median(res)) * 1e3), flush=True)


if __name__ == "__main__":
    main()

"""How far ahead of the device does the host run?  Times the Python side of N train steps (enqueue only) against the device time of
the same steps; prints both per step and the host time of forward / loss / backward / optimiser separately (cProfile-free: wall
clock around the calls, device not synchronised in between)."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "fpl-plus_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402

import fplx  # noqa: E402
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    batches = [bench.synth_batch(bench.SHAPE, d, dev) for d in (0, 1)]
    torch.manual_seed(1)
    net = fplx.UNet2D5_dsbn(dict(bench.NET)).to(dev)
    net._ensure_flat()
    ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-4, weight_decay=1e-5, milestones=[10000, 20000], gamma=0.5)
    for i in range(6):
        ts.step(batches[i % 2][0], batches[i % 2][1], i % 2)
    torch.cuda.synchronize()
    n = 20
    for rep in range(3):
        per = []
        t0 = time.perf_counter()
        for i in range(n):
            a = time.perf_counter()
            ts.step(batches[i % 2][0], batches[i % 2][1], i % 2)
            per.append(time.perf_counter() - a)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        per_ms = sorted(p * 1e3 for p in per)
        print("rep %d: host enqueue %.3f ms/step (median call %.3f, max %.3f), device done after %.3f ms/step; host finished %.2f ms before the device"
              % (rep, (t1 - t0) / n * 1e3, per_ms[n // 2], per_ms[-1], (t2 - t0) / n * 1e3, (t2 - t1) * 1e3), flush=True)
    # the first steps after a synchronise: the host has no head start
    torch.cuda.synchronize()
    per = []
    for i in range(6):
        a = time.perf_counter()
        ts.step(batches[i % 2][0], batches[i % 2][1], i % 2)
        per.append((time.perf_counter() - a) * 1e3)
    torch.cuda.synchronize()
    print("host time of the first calls after a synchronise (ms):", " ".join("%.2f" % p for p in per))


if __name__ == "__main__":
    main()

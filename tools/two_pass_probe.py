"""Would running the two domain passes of a training_all iteration CONCURRENTLY (each on its own pair of streams) pay?
Probe with two independent networks: N iterations of [A.step, B.step] on one stream against the same with B on a second stream
(the host enqueues A's step, then B's; the device overlaps whatever the queues allow).  python tools/two_pass_probe.py"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "fpl-plus_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory (fplx/_lib.py)
import torch  # noqa: E402

import fplx  # noqa: E402
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    nets, steps = [], []
    for i in range(2):
        torch.manual_seed(1 + i)
        net = fplx.UNet2D5_dsbn(dict(bench.NET)).to(dev)
        net._ensure_flat()
        nets.append(net)
        steps.append(fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-4, weight_decay=1e-5))
    batches = [bench.synth_batch(bench.SHAPE, d, dev) for d in (0, 1)]
    s1 = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()

    def iteration(concurrent):
        steps[0].step(batches[0][0], batches[0][1], 0)
        if concurrent:
            ev = torch.cuda.Event()
            ev.record(main_s)
            s1.wait_event(ev)                      # (nothing to wait for here - the shape of the real dependency)
            with torch.cuda.stream(s1):
                steps[1].step(batches[1][0], batches[1][1], 1)
            ev2 = torch.cuda.Event()
            ev2.record(s1)
            return ev2
        steps[1].step(batches[1][0], batches[1][1], 1)
        return None

    for mode in (False, True, False, True):
        for _ in range(3):
            ev = iteration(mode)
        if ev is not None:
            main_s.wait_event(ev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        last = None
        for _ in range(n):
            if last is not None:
                main_s.wait_event(last)            # the next iteration follows both passes (as the optimiser step would)
            last = iteration(mode)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        print("%-32s %.3f ms per pair of steps" % ("two streams (concurrent passes)" if mode else "one stream (one after the other)", dt))


if __name__ == "__main__":
    main()

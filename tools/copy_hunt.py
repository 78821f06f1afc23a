"""Where do the small device copies of a train step come from?  (56 `__amd_rocclr_copyBuffer` launches per
step that DESIGN.md never explained.)  Runs a few bench-shape train steps under torch.profiler with Python stacks and lists
every memcpy / copy kernel of the LAST step with the innermost fplx / bench frame that issued it.
usage: python tools/copy_hunt.py"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fpl-plus_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

import bench  # noqa: E402
import fplx  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    net = fplx.UNet2D5_dsbn(dict(bench.NET)).to(dev)
    ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-4, weight_decay=1e-5)
    batches = [bench.synth_batch(bench.SHAPE, d, dev) for d in (0, 1)]
    for i in range(3):
        ts.step(batches[i % 2][0], batches[i % 2][1], i % 2)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for i in range(2):
            ts.step(batches[i % 2][0], batches[i % 2][1], i % 2)
        torch.cuda.synchronize()
    by_site = collections.Counter()
    names = collections.Counter()
    for ev in prof.events():
        n = ev.name
        if not any(t in n.lower() for t in ("memcpy", "copybuffer", "copy_", "memset", "fillbuffer")):
            continue
        names[n] += 1
        site = "?"
        for fr in (ev.stack or []):
            if "fplx" in fr or "bench.py" in fr or "copy_hunt" in fr:
                site = fr
                break
        by_site[(n, site)] += 1
    print("copy-like events in 2 steps, by name:")
    for n, c in names.most_common():
        print("  %5d  %s" % (c, n))
    print("by (name, innermost fplx frame):")
    for (n, s), c in by_site.most_common(60):
        print("  %5d  %-40s %s" % (c, n[:40], s))
    kern = collections.Counter(ev.name for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA)
    print("device-side events by name (top 40):")
    for n, c in kern.most_common(40):
        print("  %5d  %s" % (c, n[:120]))


if __name__ == "__main__":
    main()

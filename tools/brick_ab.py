"""A/B of the brick kernels under tuning-knob settings on the benchmark's brick layers (levels 1-4, forward with statistics and
data gradient without), in ONE process, interleaved rounds; every setting's output is compared bit for bit with the first
setting's (the settings select kernels that add in the same order).

    python tools/brick_ab.py "brick_fill=256" "brick_fill=128" [--rounds=5] [--iters=20]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fpl-plus_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from fplx import ops, _lib  # noqa: E402

LAYERS = [  # (name, cin, cout, dims, stats)
    ("L1 64->64 fwd", 64, 64, (2, 40, 80, 80), 1), ("L1 64->64 dgrad", 64, 64, (2, 40, 80, 80), 0),
    ("L1 128->64 fwd", 128, 64, (2, 40, 80, 80), 1), ("L1 64->128 dgrad", 64, 128, (2, 40, 80, 80), 0),
    ("L2 128->128 fwd", 128, 128, (2, 20, 40, 40), 1), ("L2 128->128 dgrad", 128, 128, (2, 20, 40, 40), 0),
    ("L2 256->128 fwd", 256, 128, (2, 20, 40, 40), 1), ("L2 128->256 dgrad", 128, 256, (2, 20, 40, 40), 0),
    ("L2 64->128 fwd", 64, 128, (2, 20, 40, 40), 1), ("L2 128->64 dgrad", 128, 64, (2, 20, 40, 40), 0),
    ("L3 256->256 fwd", 256, 256, (2, 10, 20, 20), 1), ("L3 256->256 dgrad", 256, 256, (2, 10, 20, 20), 0),
    ("L3 512->256 fwd", 512, 256, (2, 10, 20, 20), 1), ("L3 256->512 dgrad", 256, 512, (2, 10, 20, 20), 0),
    ("L3 128->256 fwd", 128, 256, (2, 10, 20, 20), 1), ("L3 256->128 dgrad", 256, 128, (2, 10, 20, 20), 0),
    ("L4 512->512 fwd", 512, 512, (2, 5, 10, 10), 1), ("L4 512->512 dgrad", 512, 512, (2, 5, 10, 10), 0),
    ("L4 256->512 fwd", 256, 512, (2, 5, 10, 10), 1), ("L4 512->256 dgrad", 512, 256, (2, 5, 10, 10), 0),
]


# layers of the depth-march kernels for Cin = 64 / 128 (conv_fwd_march64 / _lw): the benchmark's level-0 decoder forward, and the
# shipped 2.5D configuration's Conv2d levels (mid = 1: the 9 taps in the middle depth plane), batch 4 x 28 x 128 x 128
MARCH64 = [
    ("L0 64->32 fwd", 64, 32, (2, 80, 160, 160), 1, 0), ("L0 64->32 nostats", 64, 32, (2, 80, 160, 160), 0, 0),
    ("2.5D L0 64->32 fwd", 64, 32, (4, 28, 128, 128), 1, 1), ("2.5D L1 64->64 fwd", 64, 64, (4, 28, 64, 64), 1, 1),
    ("2.5D L1 128->64 fwd", 128, 64, (4, 28, 64, 64), 1, 1), ("2.5D L1 64->128 dgrad", 64, 128, (4, 28, 64, 64), 0, 1),
    ("cfg4 L0 64->32 fwd", 64, 32, (4, 28, 128, 128), 1, 0),
]


def parse(arg):
    kn = {}
    for item in arg.split(","):
        item = item.strip()
        if item and item != "default":
            k, v = item.split("=")
            kn[k] = int(v)
    return kn


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    rounds, iters = int(opts.get("rounds", 5)), int(opts.get("iters", 20))
    only = opts.get("only")
    settings = [(a, parse(a)) for a in args] or [("default", {})]
    keys = sorted({k for _, kn in settings for k in kn})
    defaults = {k: _lib.get_tuning(k) for k in keys}
    dev = torch.device("cuda:0")
    dt = ops._DT[torch.bfloat16]
    print("%-20s" % "layer" + "".join("%24s" % s[0] for s in settings) + "   (us per launch, median of %d rounds x %d; '=' bit-identical to the first)" % (rounds, iters))
    tot = np.zeros(len(settings))
    layers = [l + (0,) for l in LAYERS] if opts.get("set", "brick") == "brick" else MARCH64
    for name, cin, cout, dims, want_stats, mid in layers:
        mid = bool(mid)
        if only and only not in name:
            continue
        n, d, h, w = dims
        v = n * d * h * w
        g = torch.Generator(device=dev).manual_seed(0)
        x = torch.randn(v, cin, device=dev, generator=g).bfloat16()
        if mid:
            wt = torch.randn(cout, cin, 3, 3, device=dev, generator=g) * 0.05
            wf, _ = ops.pack_conv2d_weight(wt, torch.bfloat16, False)
        else:
            wt = torch.randn(cout, cin, 3, 3, 3, device=dev, generator=g) * 0.05
            wf, _ = ops.pack_conv_weight(wt, torch.bfloat16)
        b = torch.randn(cout, device=dev, generator=g)
        res, times = [], [[] for _ in settings]
        outs = []
        for si, (_, kn) in enumerate(settings):
            for k in keys:
                _lib.set_tuning(k, kn.get(k, defaults[k]))
            rows = ops.conv3d_stats_rows(dims, cin, cout, (3, 3, 3), dt, dt, mid)
            stats = torch.zeros((rows, 2, cout), dtype=torch.float32, device=dev) if want_stats else None
            y = torch.full((v, cout), 7.0, device=dev, dtype=torch.bfloat16)
            ops.conv3d_fwd(x, ops.cl_strides(d, h, w, cin), dt, wf, b, y, ops.cl_strides(d, h, w, cout), dt, dims, cin, cout, (3, 3, 3), stats,
                           mid=mid)
            torch.cuda.synchronize()
            outs.append((y.clone(), None if stats is None else stats.sum(0).clone()))
        for rd in range(rounds):
            for si, (_, kn) in enumerate(settings):
                for k in keys:
                    _lib.set_tuning(k, kn.get(k, defaults[k]))
                rows = ops.conv3d_stats_rows(dims, cin, cout, (3, 3, 3), dt, dt, mid)
                stats = torch.zeros((rows, 2, cout), dtype=torch.float32, device=dev) if want_stats else None
                y = torch.empty(v, cout, device=dev, dtype=torch.bfloat16)
                run = lambda: ops.conv3d_fwd(x, ops.cl_strides(d, h, w, cin), dt, wf, b, y, ops.cl_strides(d, h, w, cout), dt, dims, cin,
                                             cout, (3, 3, 3), stats, mid=mid)
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                times[si].append(e0.elapsed_time(e1) / iters * 1e3)
        line = "%-20s" % name
        for si in range(len(settings)):
            same = torch.equal(outs[si][0], outs[0][0]) and (outs[0][1] is None or bool(torch.allclose(outs[si][1], outs[0][1], rtol=1e-5, atol=1e-3)))
            med = float(np.median(times[si]))
            tot[si] += med
            line += "%17.1f (%4.2f)%s" % (med, med / float(np.median(times[0])), "=" if same else "!")
        print(line, flush=True)
    for k in keys:
        _lib.set_tuning(k, defaults[k])
    print("%-20s" % "sum" + "".join("%17.1f (%4.2f) " % (t, t / tot[0]) for t in tot))


if __name__ == "__main__":
    main()

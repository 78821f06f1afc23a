"""Weight-gradient kernels, isolated (HIP events, interleaved rounds in one process): the benchmark's 3x3x3 layers under a list of
tuning-knob settings, e.g. the footprint march (wg_roll = 0) against the rolling-window kernel (wg_roll = 1) and its footprints.
Every setting's result is compared with the first one's (relative to the largest gradient entry).

    python tools/wgrad_bench.py [level ...]            levels 0-4 of the benchmark shape (default: 0 1 2)
    FPLX_WGB_SETTINGS="wg_roll=0;wg_roll=1;wg_roll=1,wg_roll_geo=1" python tools/wgrad_bench.py 0"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fpl-plus_amd"))
import torch  # noqa: E402

from fplx import ops, _lib  # noqa: E402

LEVELS = {0: [((2, 80, 160, 160), 32, 32), ((2, 80, 160, 160), 64, 32)],
          1: [((2, 40, 80, 80), 32, 64), ((2, 40, 80, 80), 64, 64), ((2, 40, 80, 80), 128, 64)],
          2: [((2, 20, 40, 40), 64, 128), ((2, 20, 40, 40), 128, 128), ((2, 20, 40, 40), 256, 128)],
          3: [((2, 10, 20, 20), 128, 256), ((2, 10, 20, 20), 256, 256), ((2, 10, 20, 20), 512, 256)],
          4: [((2, 5, 10, 10), 256, 512), ((2, 5, 10, 10), 512, 512)],
          25: [((4, 28, 32, 32), 64, 128), ((4, 28, 32, 32), 128, 128)]}     # level 2 of the shipped 2.5D config


def parse_settings():
    txt = os.environ.get("FPLX_WGB_SETTINGS", "wg_roll=0;wg_roll=1")
    out = []
    for s in txt.split(";"):
        kv = {}
        for item in s.split(","):
            k, v = item.split("=")
            kv[k.strip()] = int(v)
        out.append((s.strip(), kv))
    return out


def main():
    levels = [int(a) for a in sys.argv[1:]] or [0, 1, 2]
    settings = parse_settings()
    keys = sorted({k for _, kv in settings for k in kv})
    defaults = {k: _lib.get_tuning(k) for k in keys}
    bf, dt = torch.bfloat16, ops.BF16
    print("us per launch incl. the reduction, median of 5 rounds x 10 launches; fraction of 2.5 PFLOP/s; max |diff| vs the first "
          "setting / max |dw|")
    for lv in levels:
        for dims, cin, cout in LEVELS[lv]:
            n, d, h, w = dims
            v = n * d * h * w
            x = torch.randn(v, cin, device="cuda").to(bf)
            dy = (torch.randn(v, cout, device="cuda") * 0.01).to(bf)
            res = {name: [] for name, _ in settings}
            outs = {}
            for rnd in range(5):
                for name, kv in settings:
                    for k in keys:
                        _lib.set_tuning(k, kv.get(k, defaults[k]))
                    ws = torch.empty(ops.conv3d_wgrad_ws_bytes(dims, cin, cout, (3, 3, 3)), dtype=torch.uint8, device="cuda")
                    dw = torch.empty((cout, cin, 3, 3, 3), dtype=torch.float32, device="cuda")
                    f = lambda: ops.conv3d_wgrad(x, ops.cl_strides(d, h, w, cin), dt, dy, ops.cl_strides(d, h, w, cout), dt, dw,
                                                 None, dims, cin, cout, (3, 3, 3), ws)
                    for _ in range(2):
                        f()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        f()
                    e1.record()
                    torch.cuda.synchronize()
                    res[name].append(e0.elapsed_time(e1) * 100.0)
                    if rnd == 0:
                        outs[name] = dw.clone()
            for k in keys:
                _lib.set_tuning(k, defaults[k])
            fl = 2.0 * v * 27 * cin * cout
            ref = outs[settings[0][0]]
            for name, _ in settings:
                t = sorted(res[name])[2]
                err = float((outs[name] - ref).abs().max() / ref.abs().max())
                print("%-26s %-34s %8.1f us  %.3f   diff %.2e" % ("%s %d->%d" % (dims, cin, cout), name, t, fl / (t * 1e-6) / 2.5e15, err),
                      flush=True)


if __name__ == "__main__":
    main()

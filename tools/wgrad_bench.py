"""Weight-gradient kernels of the deep levels, isolated: the footprint march (wg_vox = 0) against the voxel GEMM (wg_vox = 2) on
the benchmark's level 2-4 layers (2 x 20 x 40 x 40, 2 x 10 x 20 x 20, 2 x 5 x 10 x 10), interleaved rounds in one process.
usage: python tools/wgrad_bench.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fpl-plus_amd"))
import torch  # noqa: E402

from fplx import ops, _lib  # noqa: E402

LAYERS = [((2, 20, 40, 40), 64, 128), ((2, 20, 40, 40), 128, 128), ((2, 20, 40, 40), 256, 128),
          ((2, 10, 20, 20), 128, 256), ((2, 10, 20, 20), 256, 256), ((2, 10, 20, 20), 512, 256),
          ((2, 5, 10, 10), 256, 512), ((2, 5, 10, 10), 512, 512),
          ((4, 28, 32, 32), 64, 128), ((4, 28, 32, 32), 128, 128)]       # + level 2 of the shipped 2.5D config


def main():
    bf, dt = torch.bfloat16, ops.BF16
    print("%-28s %10s %10s %8s   (us per launch incl. the reduction, median of 5 rounds x 20 launches)" % ("layer", "march", "vox", "ratio"))
    tot = [0.0, 0.0]
    for dims, cin, cout in LAYERS:
        n, d, h, w = dims
        v = n * d * h * w
        x = torch.randn(v, cin, device="cuda").to(bf)
        dy = (torch.randn(v, cout, device="cuda") * 0.01).to(bf)
        dw = torch.empty((cout, cin, 3, 3, 3), dtype=torch.float32, device="cuda")
        res = {0: [], 2: []}
        for rnd in range(5):
            for mode in (0, 2):
                _lib.set_tuning("wg_vox", mode)
                ws = torch.empty(ops.conv3d_wgrad_ws_bytes(dims, cin, cout, (3, 3, 3)), dtype=torch.uint8, device="cuda")
                f = lambda: ops.conv3d_wgrad(x, ops.cl_strides(d, h, w, cin), dt, dy, ops.cl_strides(d, h, w, cout), dt, dw, None,
                                             dims, cin, cout, (3, 3, 3), ws)
                for _ in range(3):
                    f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    f()
                e1.record()
                torch.cuda.synchronize()
                res[mode].append(e0.elapsed_time(e1) * 50.0)
        _lib.set_tuning("wg_vox", 1)
        a, b = sorted(res[0])[2], sorted(res[2])[2]
        fl = 2.0 * v * 27 * cin * cout
        print("%-28s %10.1f %10.1f %8.2f   vox: %.3f of 2.5 PFLOP/s" % ("%s %d->%d" % (dims, cin, cout), a, b, a / b, fl / (b * 1e-6) / 2.5e15))
        if dims[0] == 2:
            tot[0] += a
            tot[1] += b
    print("benchmark layers (once each): march %.0f us, vox %.0f us" % (tot[0], tot[1]))


if __name__ == "__main__":
    main()

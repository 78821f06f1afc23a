"""Times the transposed-convolution kernels (ConvTranspose3d k2 s2: forward, data gradient) at the benchmark's four decoder
levels, alone on the device; prints us and GB/s against the algorithmic bytes (Cin + 8 Cout) x 2 B per input voxel."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fpl-plus_amd"))
import torch  # noqa: E402

from fplx import ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    bf = torch.bfloat16
    for name, dims, cin, cout in (("L1->L0", (2, 40, 80, 80), 64, 32), ("L2->L1", (2, 20, 40, 40), 128, 64),
                                  ("L3->L2", (2, 10, 20, 20), 256, 128), ("L4->L3", (2, 5, 10, 10), 512, 256)):
        n, d, h, w = dims
        v = n * d * h * w
        x = torch.randn(v, cin, device=dev, generator=g).to(bf)
        wt = torch.randn(cin, cout, 2, 2, 2, device=dev, generator=g) * 0.1
        wf, wb = ops.pack_deconv_weight(wt, bf)
        b = torch.zeros(cout, device=dev)
        y = torch.empty(8 * v, cout, device=dev, dtype=bf)
        dy = torch.randn(8 * v, cout, device=dev, generator=g).to(bf)
        dx = torch.empty(v, cin, device=dev, dtype=bf)
        nb = v * (cin + 8 * cout) * 2
        t = timeit(lambda: ops.deconv2_fwd(x, wf, b, y, dims, cin, cout))
        print("%-7s deconv fwd   %3d -> %3d  %7.1f us  %6.0f GB/s" % (name, cin, cout, t, nb / t / 1e3))
        t = timeit(lambda: ops.deconv2_dgrad(dy, wb, dx, dims, cin, cout))
        print("%-7s deconv dgrad %3d <- %3d  %7.1f us  %6.0f GB/s" % (name, cin, cout, t, nb / t / 1e3))


if __name__ == "__main__":
    main()

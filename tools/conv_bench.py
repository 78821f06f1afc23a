"""Times fplx conv3d_fwd / conv3d_wgrad (bf16 NDHWC) for one shape:
python tools/conv_bench.py CIN COUT [N D H W] [stats]      (env CONV_BENCH_OP=wgrad for the weight gradient)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fpl-plus_amd"))
import torch  # noqa: E402

from fplx import ops  # noqa: E402


def main():
    a = [int(t) for t in sys.argv[1:]]
    cin, cout = a[0], a[1]
    n, d, h, w = tuple(a[2:6]) if len(a) >= 6 else (2, 80, 160, 160)
    want_stats = (a[6] if len(a) >= 7 else 1) != 0
    dev = torch.device("cuda:0")
    v = n * d * h * w
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(v, cin, device=dev, generator=g).bfloat16()
    wt = torch.randn(cout, cin, 3, 3, 3, device=dev, generator=g) * 0.05
    wf, _ = ops.pack_conv_weight(wt, torch.bfloat16)
    b = torch.randn(cout, device=dev, generator=g)
    y = torch.empty(v, cout, device=dev, dtype=torch.bfloat16)
    dt = ops._DT[torch.bfloat16]
    dims = (n, d, h, w)
    rows = ops.conv3d_stats_rows(dims, cin, cout, (3, 3, 3), dt, dt)
    stats = torch.zeros((rows, 2, cout), dtype=torch.float32, device=dev) if want_stats else None

    dyt = torch.randn(v, cout, device=dev, generator=g).bfloat16()
    ws = torch.empty(max(ops.conv3d_wgrad_ws_bytes(dims, cin, cout, (3, 3, 3)), 16), dtype=torch.uint8, device=dev)
    dw = torch.empty((cout, cin, 3, 3, 3), dtype=torch.float32, device=dev)
    db = torch.empty(cout, dtype=torch.float32, device=dev)
    op = os.environ.get("CONV_BENCH_OP", "fwd")

    def run():
        if op == "wgrad":
            ops.conv3d_wgrad(x, ops.cl_strides(d, h, w, cin), dt, dyt, ops.cl_strides(d, h, w, cout), dt, dw,
                             db if os.environ.get('CONV_BENCH_DB') else None, dims,
                             cin, cout, (3, 3, 3), ws)
            return
        ops.conv3d_fwd(x, ops.cl_strides(d, h, w, cin), dt, wf, b, y, ops.cl_strides(d, h, w, cout), dt, dims, cin, cout,
                       (3, 3, 3), stats)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * v * cin * cout * 27
    env = {k: v_ for k, v_ in os.environ.items() if k.startswith("FPLX_")}
    print(f"{op} cin={cin} cout={cout} dims={dims} rows={rows} env={env}"
          f" {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s  {(v * (cin + cout) * 2) / ms / 1e6:7.0f} GB/s")


if __name__ == "__main__":
    main()

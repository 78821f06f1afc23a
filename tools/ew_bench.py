"""Times the HBM-bound passes (BN-apply/PReLU fwd + 3-stage bwd, max-pool fwd/bwd, device copy as yardstick)
at one level's shape and prints the achieved GB/s of each.  Usage: python tools/ew_bench.py [C] [N D H W] [ld]"""
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
    return e0.elapsed_time(e1) / iters


def main():
    a = [int(t) for t in sys.argv[1:]]
    c = a[0] if a else 32
    dims = tuple(a[1:5]) if len(a) >= 5 else (2, 80, 160, 160)
    ld = a[5] if len(a) >= 6 else c
    n, d, h, w = dims
    v = n * d * h * w
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)

    def act():
        buf = torch.randn(v, ld, device=dev, generator=g).to(dt)
        return buf[:, :c]

    y, dout, dy, out = act(), act(), act(), act()
    bnbuf = torch.randn(4, c, device=dev)
    bnbuf[1].abs_().add_(0.5)
    slope = torch.full((1,), 0.25, device=dev)
    rows = ops.num_partials(v)
    part = torch.empty(rows * (2 * c + 1), device=dev)
    coef = torch.empty(2 * c, device=dev)
    dgamma, dbeta, dslope = torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.zeros(1, device=dev)
    tb = v * c * 2  # bytes of one activation tensor
    res = []

    def rec(name, ms, nbytes):
        res.append((name, ms, nbytes / ms / 1e6))

    src, dst = torch.empty(v * c, device=dev, dtype=dt), torch.empty(v * c, device=dev, dtype=dt)
    rec("torch copy (r+w)", timeit(lambda: dst.copy_(src)), 2 * tb)
    rec("torch sum (r)", timeit(lambda: src.view(torch.int16).sum()), tb)
    for p in (0.0, 0.5):
        rec(f"bn_act_fwd p={p}", timeit(lambda: ops.bn_act_fwd(y, out, bnbuf, slope, p, 1234, 3, c)), 2 * tb)
        rec(f"bn_act_bwd p={p} (reduce+finalize+apply)",
            timeit(lambda: ops.bn_act_bwd(y, dout, dy, bnbuf, slope, p, 1234, 3, c, True, dgamma, dbeta, dslope, part,
                                          coef)), 5 * tb)
        lib_call = ops.call
        rec(f"  bwd_reduce p={p}", timeit(lambda: lib_call(
            "fplx_bn_act_bwd_reduce", ops.ptr(y), ops.ld_of(y), ops.ptr(dout), ops.ld_of(dout), ops.ptr(bnbuf[0]),
            ops.ptr(bnbuf[1]), ops.ptr(bnbuf[2]), ops.ptr(bnbuf[3]), ops.ptr(slope), float(p), 1234, 3, v, c,
            ops.dt_of(y), ops.ptr(part), ops.stream())), 2 * tb)
        rec(f"  bwd_apply p={p}", timeit(lambda: lib_call(
            "fplx_bn_act_bwd_apply", ops.ptr(y), ops.ld_of(y), ops.ptr(dout), ops.ld_of(dout), ops.ptr(dy),
            ops.ld_of(dy), ops.ptr(bnbuf[0]), ops.ptr(bnbuf[1]), ops.ptr(bnbuf[2]), ops.ptr(bnbuf[3]), ops.ptr(slope),
            ops.ptr(coef), float(p), 1234, 3, v, c, ops.dt_of(y), ops.stream())), 3 * tb)
    if d % 2 == 0 and h % 2 == 0 and w % 2 == 0:
        vo = v // 8
        po = torch.empty(vo, c, device=dev, dtype=dt)
        dpo = torch.randn(vo, c, device=dev, generator=g).to(dt)
        dskip = act()
        dx = torch.empty(v, c, device=dev, dtype=dt)
        rec("maxpool2_fwd", timeit(lambda: ops.maxpool2_fwd(y, po, dims, c)), tb + tb // 8)
        rec("maxpool2_bwd (+skip)", timeit(lambda: ops.maxpool2_bwd(y, dpo, dskip, dx, dims, c)), 3 * tb + tb // 8)
    stats = torch.empty(rows * 2 * c, device=dev)
    rec("channel_stats", timeit(lambda: ops.call("fplx_channel_stats", ops.ptr(y), ops.ld_of(y), v, c, ops.dt_of(y),
                                                 ops.ptr(stats), ops.stream())), tb)
    print(f"C={c} dims={dims} ld={ld} tensor={tb / 1e6:.1f} MB")
    for name, ms, gbs in res:
        print(f"{name:45s} {ms * 1e3:9.1f} us {gbs:9.0f} GB/s")


if __name__ == "__main__":
    main()

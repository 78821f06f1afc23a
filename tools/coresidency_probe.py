"""What a memory-bound pass costs beside a weight-gradient kernel on the second stream (the backward's normal situation): the
level-0 bn_act_bwd apply pass, alone and while a conv3d_wgrad of a chosen layer runs on another stream, for the pass's
register footprints (ew_inflight 4 / 2 / 1 = 131 / 99 / 72 registers; a weight-gradient wave holds 292-504 of the SIMD's 512).
Prints the pass's average time per launch in each situation and the weight gradient's own time with and without the pass.

    python tools/coresidency_probe.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fpl-plus_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory (fplx/_lib.py)
import torch  # noqa: E402

from fplx import ops, _lib  # noqa: E402


def main():
    dev, bf = torch.device("cuda:0"), torch.bfloat16
    n, d, h, w = 2, 80, 160, 160
    dims, v = (n, d, h, w), n * d * h * w
    c = 32
    y, dout, dy = (torch.randn(v, c, device=dev).to(bf) for _ in range(3))
    bnbuf = torch.randn(4, c, device=dev)
    bnbuf[1].abs_().add_(0.5)
    slope = torch.full((1,), 0.25, device=dev)
    coef = torch.randn(2, c, device=dev) * 0.01
    side = torch.cuda.Stream()

    def apply_pass():
        _lib.call("fplx_bn_act_bwd_apply", ops.ptr(y), c, ops.ptr(dout), c, ops.ptr(dy), c, ops.ptr(bnbuf[0]), ops.ptr(bnbuf[1]),
                  ops.ptr(bnbuf[2]), ops.ptr(bnbuf[3]), ops.ptr(slope), ops.ptr(coef), 0.0, 0, 0, v, c, ops.BF16, ops.stream())

    layers = {"32->32 (292 registers)": (32, 32), "64->32 two ci tiles (504)": (64, 32), "32->64 two co tiles (440)": (32, 64)}
    for lname, (cin, cout) in layers.items():
        x = torch.randn(v, cin, device=dev).to(bf)
        g = torch.randn(v, cout, device=dev).to(bf)
        dw = torch.empty(cout, cin, 3, 3, 3, device=dev)
        ws = torch.empty(ops.conv3d_wgrad_ws_bytes(dims, cin, cout, (3, 3, 3)), dtype=torch.uint8, device=dev)

        def wgrad():
            ops.conv3d_wgrad(x, ops.cl_strides(d, h, w, cin), ops.BF16, g, ops.cl_strides(d, h, w, cout), ops.BF16, dw, None, dims, cin,
                             cout, (3, 3, 3), ws)

        def timed(fn, reps, other=None, other_reps=0):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if other is not None:
                with torch.cuda.stream(side):
                    for _ in range(other_reps):
                        other()
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps

        wg_alone = timed(wgrad, 10)
        print("weight gradient %s alone: %.1f us" % (lname, wg_alone), flush=True)
        for infl in (4, 2, 1):
            _lib.set_tuning("ew_inflight", infl)
            alone = timed(apply_pass, 20)
            # enough weight-gradient launches on the side stream to cover the 20 passes
            k = int(20 * alone * 3 / wg_alone) + 4
            beside = timed(apply_pass, 20, wgrad, k)
            # and the other way round: the weight gradient's time while passes keep the main stream busy
            wg_beside = timed(wgrad, 10, apply_pass, int(10 * wg_alone * 3 / alone) + 4)
            print("  apply pass, ew_inflight %d: alone %.1f us, beside the weight gradient %.1f us (x %.2f); weight gradient beside "
                  "the passes %.1f us (x %.2f)" % (infl, alone, beside, beside / alone, wg_beside, wg_beside / wg_alone), flush=True)
        _lib.set_tuning("ew_inflight", 2)


if __name__ == "__main__":
    main()

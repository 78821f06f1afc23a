"""What does a cross-stream hand-off cost the stream that records the event?  Engine.backward records one event per weight
gradient on the main stream (23 per step); the two-stream timeline shows 6-7 us between the kernel in front of such a record
and the kernel behind it, 0.1 us between kernels without one.  This probe times a chain of N small kernels on one stream
with and without a record between them (torch events; HIP events created with other flags through ctypes).
python tools/event_cost_probe.py"""
import ctypes
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fpl-plus_amd"))
from fplx import ops  # noqa: E402


def main():
    hip = ctypes.CDLL("libamdhip64.so")
    dev = torch.device("cuda", 0)
    v, c = 2048000, 64          # a 100-us kernel: the host stays far ahead of the device, the chain time is device time
    y = torch.randn(v, c, device=dev).bfloat16()
    a = torch.empty_like(y)
    buf = torch.ones(4, c, device=dev)
    slope = torch.full((1,), 0.25, device=dev)
    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    N = 60

    def kern():
        ops.bn_act_fwd(y, a, buf, slope, 0.0, 1, 1, c)

    def timed(between, reps=5):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for i in range(N):
                kern()
                between()
            t1.record()
            torch.cuda.synchronize()
            best = min(best, t0.elapsed_time(t1) * 1e3 / N)
        return best

    base = timed(lambda: None)
    print("kernel chain, no events:                         %.2f us per kernel" % base)

    def torch_rec():
        ev = torch.cuda.Event()
        ev.record(main_s)
    print("torch.cuda.Event().record() after each:         +%.2f us" % (timed(torch_rec) - base))

    def torch_rec_wait():
        ev = torch.cuda.Event()
        ev.record(main_s)
        side.wait_event(ev)
    print("... + side.wait_event (nothing queued on side): +%.2f us" % (timed(torch_rec_wait) - base))

    def torch_rec_wait_work():
        ev = torch.cuda.Event()
        ev.record(main_s)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            ops.bn_act_fwd(y, a2, buf, slope, 0.0, 1, 1, c)
    a2 = torch.empty_like(y)
    print("... + a kernel on the side stream behind it:     +%.2f us" % (timed(torch_rec_wait_work) - base))
    torch.cuda.synchronize()

    hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
    for name, flags in (("hipEventDisableTiming", 0x2), ("DisableTiming | ReleaseToDevice", 0x2 | 0x40000000),
                        ("DisableTiming | DisableSystemFence", 0x2 | 0x20000000), ("DisableTiming | ReleaseToSystem", 0x2 | 0x80000000),
                        ("default (timing)", 0x0)):
        evs = []
        for _ in range(N):
            e = ctypes.c_void_p()
            assert hip.hipEventCreateWithFlags(ctypes.byref(e), flags) == 0
            evs.append(e)
        it = [0]
        ms, ss = ctypes.c_void_p(main_s.cuda_stream), ctypes.c_void_p(side.cuda_stream)

        def rec():
            e = evs[it[0] % N]
            it[0] += 1
            assert hip.hipEventRecord(e, ms) == 0

        def rec_wait():
            e = evs[it[0] % N]
            it[0] += 1
            assert hip.hipEventRecord(e, ms) == 0
            assert hip.hipStreamWaitEvent(ss, e, 0) == 0
            with torch.cuda.stream(side):
                ops.bn_act_fwd(y, a2, buf, slope, 0.0, 1, 1, c)
        print("HIP event, %-36s record: +%.2f us   record + wait + side kernel: +%.2f us"
              % (name + ":", timed(rec) - base, timed(rec_wait) - base))
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()

"""Times fplx_pack_conv_weight (fp32 master -> bf16 wf[tap][co][ci] + wb[26-tap][ci][co]) per layer shape and checks it
against a torch permute.  FPLX_PACK_TILED=0 selects the element-wise kernel."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fpl-plus_amd"))
import torch  # noqa: E402

from fplx import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    tot = 0.0
    for co, ci in ((32, 32), (32, 64), (64, 32), (64, 64), (64, 128), (128, 64), (128, 128), (128, 256), (256, 128),
                   (256, 256), (256, 512), (512, 256), (512, 512)):
        w = torch.randn(co, ci, 3, 3, 3, device=dev)
        wf, wb = ops.pack_conv_weight(w, torch.bfloat16)
        ref_f = w.reshape(co, ci, 27).permute(2, 0, 1).bfloat16()
        ref_b = w.reshape(co, ci, 27).flip(2).permute(2, 1, 0).bfloat16()
        assert torch.equal(wf, ref_f) and torch.equal(wb, ref_b), (co, ci)
        for _ in range(3):
            ops.pack_conv_weight(w, torch.bfloat16)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.call("fplx_pack_conv_weight", ops.ptr(w), ops.ptr(wf), ops.ptr(wb), co, ci, 3, 3, 3, ops.BF16, ops.stream())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        tot += us
        print("pack %3dx%3d  %7.1f us  %6.0f GB/s" % (co, ci, us, co * ci * 27 * 8 / us / 1e3))
    print("sum %.1f us  (FPLX_PACK_TILED=%s)" % (tot, os.environ.get("FPLX_PACK_TILED", "1")))


if __name__ == "__main__":
    main()

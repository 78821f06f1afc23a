"""Secondary measurement (not a bench.py line): the FPL+ pseudo-label selection path on one
hrT2-sized volume - T Monte-Carlo forwards (eval-mode BatchNorm, dropout active), then the variance / entropy filter.
Prints volumes/s for T = 4 (and the reference-literal T = 6 x 4-flip TTA = 24 forwards) and the filter kernel's GB/s.
usage: python tools/fpl_infer_bench.py [D H W]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fpl-plus_amd"))
import torch  # noqa: E402

import fplx  # noqa: E402
from fplx import ops  # noqa: E402
from fplx.infer import Inferer  # noqa: E402


def main():
    a = [int(t) for t in sys.argv[1:]]
    d, h, w = a if len(a) == 3 else (48, 160, 272)
    dev = torch.device("cuda:0")
    net = fplx.UNet2D5_dsbn(dict(in_chns=1, feature_chns=[32, 64, 128, 256, 512], dropout=[0, 0, 0.3, 0.4, 0.5],
                                 conv_dims=[3, 3, 3, 3, 3], class_num=2, bilinear=False, num_domains=2,
                                 precision="bf16")).to(dev)
    net.eval()
    for m in net.modules():                                   # test-time dropout (agent_seg.py:845-852)
        if type(m) == torch.nn.Dropout:
            m.train()
    x = torch.randn(1, 1, d, h, w, device=dev)
    dl = torch.ones(1, dtype=torch.long)

    def mc(T, tta):
        # all T passes x flips gathered into one batch, forwards in chunks of `infer_batch_voxels`, ordered merge (fplx/infer.py)
        inf = Inferer(dict(class_num=2, tta_mode=tta, infer_batch_voxels=int(os.environ.get("INFER_BATCH_VOXELS", 1 << 23))))
        with torch.no_grad():
            stack = inf.run_mc(net, x, dl, T)[:, 0]
        return stack, ops.mc_filter(stack, 0.01)

    for T, tta in ((4, 0), (6, 1)):
        mc(T, tta)
        torch.cuda.synchronize()
        t0 = time.time()
        reps = 3
        for _ in range(reps):
            stack, r = mc(T, tta)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / reps
        print("T=%d tta=%d (%d forwards) on 1x1x%dx%dx%d: %.1f ms per volume = %.2f volumes/s" %
              (T, tta, T * (4 if tta else 1), d, h, w, dt * 1e3, 1.0 / dt))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.mc_filter(stack, 0.01)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("mc_filter alone (T=%d): %.3f ms, %.0f GB/s of logits read" % (stack.shape[0], ms, stack.numel() * 4 / ms / 1e6))


if __name__ == "__main__":
    main()

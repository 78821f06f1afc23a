"""Train-step rate at the SHIPPED configuration (config_dual/data_vs/vs_t1s_g.cfg): UNet2D5_dsbn 32-base,
conv_dims = [2, 2, 3, 3, 3], crops 28 x 128 x 128, batch 4 per domain, training_all (both domains forward, one Adam
step), bf16 activations.  Prints ms per iteration and crops/s; not a bench.py line (BASELINE's metric is the all-3D
80 x 160 x 160 step)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fpl-plus_amd"))
import torch  # noqa: E402

import fplx  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    p = dict(in_chns=1, feature_chns=[32, 64, 128, 256, 512], dropout=[0.0, 0.0, 0.3, 0.4, 0.5], conv_dims=[2, 2, 3, 3, 3],
             class_num=2, bilinear=False, num_domains=2, precision=prec)
    torch.manual_seed(1)
    net = fplx.UNet2D5_dsbn(p).cuda()
    ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-4, weight_decay=1e-5)
    g = torch.Generator().manual_seed(0)
    batches = []
    for d in range(2):
        x = torch.randn(bs, 1, 28, 128, 128, generator=g).cuda()
        lab = torch.zeros(bs, 2, 28, 128, 128)
        lab[:, 0] = 1.0
        lab[:, 0, 8:20, 40:90, 30:100] = 0.0
        lab[:, 1, 8:20, 40:90, 30:100] = 1.0
        batches.append({"image": x, "label_prob": lab.cuda()})
    for _ in range(3):
        ts.step_all(batches)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        out = ts.step_all(batches)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("shipped cfg (2.5D, %s, batch %d x 2 domains, 28x128x128): %.2f ms per training_all iteration, %.0f crops/s, loss %.4f"
          % (prec, bs, dt * 1e3, 2 * bs / dt, float(out[0][0].item())))


if __name__ == "__main__":
    main()

"""Train-step rate at BASELINE config 5 on ONE GPU: dual-domain UNet3D-DSBN, in_chns 4, 32-base bf16, N x 4 x 128^3 crops
per domain (N = 2 default), CombinedLoss 0.5 Dice + 0.5 CE with pixel_weight = {0, w_img}; one `training_all` iteration =
both domains forward + backward, one Adam step.  Prints ms per iteration and crops/s; not a bench.py line."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fpl-plus_amd"))
import torch  # noqa: E402

import fplx  # noqa: E402


def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    p = dict(in_chns=4, feature_chns=[32, 64, 128, 256, 512], dropout=[0.0, 0.0, 0.3, 0.4, 0.5], conv_dims=[3] * 5,
             class_num=2, bilinear=False, num_domains=2, precision="bf16")
    torch.manual_seed(1)
    net = fplx.UNet2D5_dsbn(p).cuda()
    loss = fplx.make_loss({"loss_type": ["DiceLoss", "CrossEntropyLoss"], "loss_weight": [0.5, 0.5]})
    ts = fplx.TrainStep(net, loss.terms, True, lr=1e-4, weight_decay=1e-5)
    g = torch.Generator().manual_seed(0)
    batches = []
    for d in range(2):
        x = torch.randn(bs, 4, 128, 128, 128, generator=g)
        lab = torch.zeros(bs, 2, 128, 128, 128)
        lab[:, 0] = 1.0
        lab[:, 0, 40:90, 30:100, 50:110] = 0.0
        lab[:, 1, 40:90, 30:100, 50:110] = 1.0
        w_img = torch.rand(bs, generator=g) + 0.01
        pw = (torch.rand(bs, 1, 128, 128, 128, generator=g) > 0.1).float() * w_img.view(bs, 1, 1, 1, 1)
        batches.append({"image": x.cuda(), "label_prob": lab.cuda(), "pixel_weight": pw.cuda(), "image_weight": w_img.cuda()})
    for _ in range(3):
        ts.step_all(batches)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        out = ts.step_all(batches)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("config 5 (4-ch 128^3, bf16, batch %d x 2 domains, Dice+CE weighted): %.2f ms per training_all iteration, "
          "%.1f crops/s, loss %.4f, peak HBM %.1f GB" % (bs, dt * 1e3, 2 * bs / dt, float(out[0][0].item()),
                                                       torch.cuda.max_memory_allocated() / 2 ** 30))


if __name__ == "__main__":
    main()

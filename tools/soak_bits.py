"""Bitwise soak of the train step: the same few iterations from the same initial state, again and again - every repetition must
give the same parameter bits (all reductions run in a fixed order; a difference means a data race or a hardware hazard, like
the store-data hazard round 2 found in the march kernels' inline-asm stores).

    python tools/soak_bits.py [reps] [cfg]      cfg: bench (2 x 80 x 160 x 160, all 3D) | ship (4 x 28 x 128 x 128, 2.5D) | c5"""
import os
import sys

sys.path.insert(0, os.path.join(os.getcwd(), "fpl-plus_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory (fplx/_lib.py)
import torch  # noqa: E402
import fplx  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = sys.argv[2] if len(sys.argv) > 2 else "bench"
dims, shape, cin = {"bench": ([3] * 5, (2, 1, 80, 160, 160), 1), "ship": ([2, 2, 3, 3, 3], (4, 1, 28, 128, 128), 1),
                    "c5": ([3] * 5, (2, 4, 128, 128, 128), 4)}[cfg]
NET = dict(in_chns=cin, feature_chns=[32, 64, 128, 256, 512], dropout=[0.0, 0.0, 0.3, 0.4, 0.5], conv_dims=dims, class_num=2,
           bilinear=False, num_domains=2, precision="bf16")
g = torch.Generator().manual_seed(0)
x = torch.randn(*shape, generator=g).cuda()
n, _, D, H, W = shape
lab = torch.zeros(n, 2, D, H, W)
lab[:, 0] = 1.0
lab[:, 0, D // 4:D // 2, H // 4:H // 2, W // 3:2 * W // 3] = 0.0
lab[:, 1, D // 4:D // 2, H // 4:H // 2, W // 3:2 * W // 3] = 1.0
lab = lab.cuda()


def run():
    torch.manual_seed(1)
    net = fplx.UNet2D5_dsbn(dict(NET)).cuda()
    ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-3, weight_decay=1e-5)
    for i in range(6):
        out = ts.step(x, lab, i % 2)
    torch.cuda.synchronize()
    return net.flat_params.clone(), float(out[0])


ref, loss = run()
bad = 0
for r in range(reps):
    p, l = run()
    if not torch.equal(p, ref):
        bad += 1
        print("repetition %d differs: %d of %d parameters, max |diff| %.3e" % (r, int((p != ref).sum()), p.numel(),
                                                                              float((p - ref).abs().max())), flush=True)
print("%s: %d repetitions of 6 steps, %d differ from the first; loss %.6f, finite %s" % (cfg, reps, bad, loss, bool(torch.isfinite(ref).all())))

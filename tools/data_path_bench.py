"""Times the GPU data path at the shipped VS config (config_dual/data_vs/vs_t1s_g.cfg): hrT2-sized volumes
1x40x160x272 (float32 image, uint8 label, float32 pixel weight) resident in HBM ->
NormalizeWithMeanStd -> Pad[28,128,128] -> RandomCrop[28,128,128] (foreground focus 0.5) -> RandomFlip(h, w)
-> LabelToProbability(2) -> collate to a batch of 2.  Prints samples/s; the train step consumes ~160 volumes/s."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fpl-plus_amd"))
import torch  # noqa: E402

from fplx import transform as T  # noqa: E402
from fplx.dataset import collate  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    cases = []
    for i in range(8):
        img = torch.randn(1, 40, 160, 272, device=dev, generator=g) * 40 + 200
        lab = torch.zeros(1, 40, 160, 272, dtype=torch.uint8, device=dev)
        lab[0, 15:25, 70:95, 120:150] = 1
        pw = (torch.rand(1, 40, 160, 272, device=dev, generator=g) > 0.1).float()
        cases.append({"image": img, "label": lab, "pixel_weight": pw, "image_weight": 0.8})
    p = {"task": "segmentation", "normalizewithmeanstd_channels": [0], "pad_output_size": [28, 128, 128],
         "randomcrop_output_size": [28, 128, 128], "randomcrop_foreground_focus": True, "randomcrop_foreground_ratio": 0.5,
         "randomcrop_mask_label": [1, 2], "randomflip_flip_depth": False, "randomflip_flip_height": True,
         "randomflip_flip_width": True, "labeltoprobability_class_num": 2}
    chain = T.Compose(T.build_transforms(["NormalizeWithMeanStd", "Pad", "RandomCrop", "RandomFlip", "LabelToProbability"], p))
    random.seed(0)

    def sample(i):
        s = dict(cases[i % 8])
        s["image"] = s["image"].clone()
        return chain(s)

    for i in range(10):
        collate([sample(i), sample(i + 1)])
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for i in range(n):
        b = collate([sample(2 * i), sample(2 * i + 1)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("data path: %d batches of 2 in %.3f s -> %.0f samples/s (%.2f ms per batch), batch image %s label_prob %s"
          % (n, dt, 2 * n / dt, dt / n * 1e3, tuple(b["image"].shape), tuple(b["label_prob"].shape)))


if __name__ == "__main__":
    main()

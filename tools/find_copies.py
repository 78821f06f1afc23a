"""where do the small device-to-device copies of a train step come from? (torch profiler, with python stacks)"""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "fpl-plus_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, fplx
from torch.profiler import profile, ProfilerActivity
NET = dict(in_chns=1, feature_chns=[32, 64, 128, 256, 512], dropout=[0.0, 0.0, 0.3, 0.4, 0.5], conv_dims=[3] * 5, class_num=2,
           bilinear=False, num_domains=2, precision="bf16")
torch.manual_seed(1)
net = fplx.UNet2D5_dsbn(NET).cuda()
ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True)
x = torch.randn(2, 1, 80, 160, 160).cuda()
lab = torch.zeros(2, 2, 80, 160, 160).cuda(); lab[:, 0] = 1
for i in range(3):
    ts.step(x, lab, i % 2)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ts.step(x, lab, 1)
    torch.cuda.synchronize()
from collections import Counter
c = Counter()
for ev in prof.events():
    n = ev.name
    if any(k in n for k in ("copy_", "fill_", "zero_", "aten::to", "_to_copy", "clone", "contiguous", "Memcpy", "Memset", "add_")):
        st = [s for s in ev.stack if "fplx" in s or "train.py" in s or "engine.py" in s or "ops.py" in s][:2]
        c[(n, tuple(st))] += 1
for (n, st), k in c.most_common(25):
    print(k, n, st)

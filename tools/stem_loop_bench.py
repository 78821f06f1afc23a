"""stem forward timed N times in one process (fresh output tensor each time or the same one)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fpl-plus_amd"))
import torch
from fplx import ops
n, d, h, w, c0 = 2, 80, 160, 160, 32
dims, v = (n, d, h, w), n * d * h * w
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
img = torch.randn(n, 1, d, h, w, device=dev, generator=g)
ws_ = torch.randn(c0, 1, 3, 3, 3, device=dev, generator=g) * 0.1
wsf, _ = ops.pack_conv_weight(ws_, torch.bfloat16, False)
bs = torch.zeros(c0, device=dev)
rows = ops.conv3d_stats_rows(dims, 1, c0, (3, 3, 3), ops.F32, ops.BF16)
stats = torch.zeros((rows, 2, c0), device=dev)
feats = [torch.empty(v, c0, device=dev, dtype=torch.bfloat16) for _ in range(3)]
res = []
for rep in range(8):
    feat = feats[rep % 3]
    fn = lambda: ops.conv3d_fwd(img, ops.planar_strides(1, d, h, w), ops.F32, wsf, bs, feat, ops.cl_strides(d, h, w, c0), ops.BF16, dims, 1, c0, (3, 3, 3), stats)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 20 * 1e3)
print(" ".join("%.1f" % t for t in res), " ptrs", [hex(f.data_ptr()) for f in feats])

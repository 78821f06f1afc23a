import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "fpl-plus_amd"))
import torch
from fplx import ops
def run(cin, cout, dims, mid):
    n, d, h, w = dims
    v = n*d*h*w
    dev = torch.device("cuda:0")
    x = torch.randn(v, cin, device=dev).bfloat16()
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    wf, _ = ops.pack_conv2d_weight(wt, torch.bfloat16)
    b = torch.zeros(cout, device=dev)
    y = torch.empty(v, cout, device=dev, dtype=torch.bfloat16)
    dt = ops._DT[torch.bfloat16]
    rows = ops.conv3d_stats_rows(dims, cin, cout, (3,3,3), dt, dt, mid)
    stats = torch.zeros((rows, 2, cout), device=dev)
    f = lambda: ops.conv3d_fwd(x, ops.cl_strides(d,h,w,cin), dt, wf, b, y, ops.cl_strides(d,h,w,cout), dt, dims, cin, cout, (3,3,3), stats, mid=mid)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)/20*1e3
    print("cin=%d cout=%d dims=%s mid=%s rows=%d %8.1f us  (9-tap work: %.0f TF/s)" % (cin, cout, dims, mid, rows, us, 2.0*v*cin*cout*9/us/1e6))
for cin, cout, dims in ((64,64,(4,28,64,64)), (32,64,(4,28,64,64)), (128,64,(4,28,64,64)), (64,128,(4,28,64,64))):
    for mid in (False, True):
        run(cin, cout, dims, mid)

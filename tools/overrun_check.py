"""Do the 2D-level weight-gradient launches write outside their dw / db tensors?  Each output sits in the middle of a
sentinel-filled buffer; prints the number of sentinel elements that changed."""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "fpl-plus_amd"))
import torch
from fplx import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
dt = ops._DT[bf]
PAD = 4096
def guarded(shape):
    n = 1
    for s in shape: n *= s
    buf = torch.full((PAD + n + PAD,), 7.0, dtype=torch.float32, device=dev)
    return buf, buf[PAD:PAD + n].view(shape)
WPAD = 1 << 22
def guarded_ws(nbytes):
    """workspace of exactly nbytes followed by 4 MiB of sentinel bytes (the library is told nbytes)"""
    nbytes = (int(nbytes) + 255) // 256 * 256
    buf = torch.full((nbytes + WPAD,), 0x5A, dtype=torch.uint8, device=dev)
    return buf, buf[:nbytes]
def bad_ws(buf):
    return int((buf[-WPAD:] != 0x5A).sum())
def bad(buf, n):
    return int((buf[:PAD] != 7.0).sum()) + int((buf[PAD + n:] != 7.0).sum())
for dims, cin, cout in (((4, 28, 128, 128), 32, 32), ((4, 28, 64, 64), 32, 64), ((4, 28, 64, 64), 64, 64), ((4, 28, 64, 64), 128, 64),
                        ((4, 28, 128, 128), 1, 32)):
    n, d, h, w = dims
    v = n * d * h * w
    if cin == 1:
        x = torch.randn(n, 1, d, h, w, device=dev); xs = ops.planar_strides(1, d, h, w); xdt = ops._DT[torch.float32]
    else:
        x = torch.randn(v, cin, device=dev).to(bf); xs = ops.cl_strides(d, h, w, cin); xdt = dt
    dy = torch.randn(v, cout, device=dev).to(bf)
    wb, ws = guarded_ws(ops.conv2d_wgrad_ws_bytes(dims, cin, cout))
    buf, dw = guarded((cout, cin, 3, 3))
    ops.conv2d_wgrad(x, xs, xdt, dy, ops.cl_strides(d, h, w, cout), dt, dw, None, dims, cin, cout, ws)
    torch.cuda.synchronize()
    print("conv2d_wgrad", dims, cin, cout, "sentinels changed:", bad(buf, dw.numel()), "past the workspace:", bad_ws(wb))
for dims, cin, cout in (((4, 28, 64, 64), 64, 32), ((4, 28, 32, 32), 128, 64)):
    n, d, h, w = dims
    v = n * d * h * w
    x = torch.randn(v, cin, device=dev).to(bf)
    dyo = torch.randn(v * 4, cout, device=dev).to(bf)
    wb, ws = guarded_ws(ops.deconv2_wgrad_ws_bytes(dims, cin, cout, 1))
    bw, dw = guarded((cin, cout, 2, 2))
    bb, db = guarded((cout,))
    ops.deconv2_wgrad(x, dyo, dw, db, dims, cin, cout, ws, 1)
    torch.cuda.synchronize()
    print("deconv122_wgrad", dims, cin, cout, "sentinels changed:", bad(bw, dw.numel()), bad(bb, db.numel()),
          "past the workspace:", bad_ws(wb))
n, d, h, w = dims = (4, 28, 128, 128)
v = n * d * h * w
x0, x1 = torch.randn(v, 32, device=dev).to(bf), torch.randn(v, 32, device=dev).to(bf)
dy = torch.randn(v, 32, device=dev).to(bf)
wb, ws = guarded_ws(max(ops.conv2d_wgrad_ws_bytes(dims, 64, 32), ops.conv3d_wgrad_ws_bytes(dims, 64, 32, (3, 3, 3))))
buf, dw = guarded((32, 64, 3, 3))
ops.conv3d_wgrad_cat2(x0, x1, dy, dw, dims, 64, 32, ws, True)
torch.cuda.synchronize()
print("conv2d_wgrad_cat2", "sentinels changed:", bad(buf, dw.numel()), "past the workspace:", bad_ws(wb))

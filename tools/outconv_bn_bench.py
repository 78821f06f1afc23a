import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "fpl-plus_amd"))
import torch
from fplx import ops
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
n, d, h, w, c0, ncls = 2, 80, 160, 160, 32, 2
dims, v, bf = (n, d, h, w), n*d*h*w, torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
y = torch.randn(v, c0, device="cuda", generator=g).to(bf)
a = torch.empty_like(y); dd = torch.empty_like(y); dy = torch.empty_like(y)
bnbuf = torch.randn(4, c0, device="cuda"); bnbuf[1].abs_().add_(0.5)
slope = torch.full((1,), 0.25, device="cuda")
wo = torch.randn(ncls, c0, 1, 3, 3, device="cuda") * 0.1
wof, _ = ops.pack_conv_weight(wo, torch.float32, False); _, wob = ops.pack_conv_weight(wo, bf, True)
bias = torch.zeros(ncls, device="cuda")
lg = torch.empty(n, ncls, d, h, w, device="cuda"); dl = torch.randn(n, ncls, d, h, w, device="cuda") * 0.05
cl, pl = ops.cl_strides, ops.planar_strides
part = torch.empty((2048, 2*1024+1), device="cuda"); coef = torch.empty((2, 1024), device="cuda")
gr = [torch.zeros(c0, device="cuda"), torch.zeros(c0, device="cuda"), torch.zeros(1, device="cuda")]
t1 = timeit(lambda: ops.bn_act_fwd(y, a, bnbuf, slope, 0.0, 0, 0, c0))
t2 = timeit(lambda: ops.conv3d_fwd(a, cl(d,h,w,c0), ops.BF16, wof, bias, lg, pl(ncls,d,h,w), ops.F32, dims, c0, ncls, (1,3,3), None))
t3 = timeit(lambda: ops.outconv_fwd_bn(y, bnbuf, slope, a, wof, bias, lg, dims, c0, ncls))
print("forward: bn_act_fwd %.1f + out_conv %.1f = %.1f us; fused %.1f us" % (t1, t2, t1+t2, t3))
t4 = timeit(lambda: ops.conv3d_fwd(dl, pl(ncls,d,h,w), ops.F32, wob, None, dd, cl(d,h,w,c0), ops.BF16, dims, ncls, c0, (1,3,3), None))
t5 = timeit(lambda: ops.bn_act_bwd(y, dd, dy, bnbuf, slope, 0.0, 0, 0, c0, True, gr[0], gr[1], gr[2], part, coef))
t6 = timeit(lambda: ops.outconv_dgrad_bn_bwd(dl, wob, y, bnbuf, slope, True, gr[0], gr[1], gr[2], part, coef, dy, dims, c0, ncls))
print("backward: dgrad %.1f + bn_act_bwd (reduce, finalize, apply) %.1f = %.1f us; fused %.1f us" % (t4, t5, t4+t5, t6))
from fplx import _lib
t7 = timeit(lambda: _lib.call("fplx_outconv_dgrad_bn_reduce", ops.ptr(dl), ops.ptr(wob), ops.ptr(y), c0, ops.ptr(bnbuf[0]), ops.ptr(bnbuf[1]), ops.ptr(bnbuf[2]), ops.ptr(bnbuf[3]), ops.ptr(slope), ops.ptr(part), n,d,h,w,c0,ncls, ops.stream()))
t8 = timeit(lambda: _lib.call("fplx_outconv_dgrad_bn_apply", ops.ptr(dl), ops.ptr(wob), ops.ptr(y), c0, ops.ptr(bnbuf[0]), ops.ptr(bnbuf[1]), ops.ptr(bnbuf[2]), ops.ptr(bnbuf[3]), ops.ptr(slope), ops.ptr(coef), ops.ptr(dy), c0, n,d,h,w,c0,ncls, ops.stream()))
print("fused reduce form %.1f us, fused apply form %.1f us" % (t7, t8))

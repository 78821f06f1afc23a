"""Locate the stream-order hazard of the 2.5D backward: one forward of the shipped-style network
(32-base, conv_dims = [2,2,3,3,3], bf16, 4 x 1 x 28 x 128 x 128), then the SAME backward again and again under different
stream schedules; every run's flat gradient is compared, parameter by parameter, with the single-stream result.

    python tools/race25.py [reps] [dims]        dims e.g. 22333 (default) or 33333
"""
import os
import sys

sys.path.insert(0, os.path.join(os.getcwd(), "fpl-plus_amd"))
import torch  # noqa: E402
import fplx  # noqa: E402
from fplx import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dims = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "22333")]
shape = (4, 1, 28, 128, 128) if 2 in dims else (2, 1, 80, 160, 160)
p = dict(in_chns=1, feature_chns=[32, 64, 128, 256, 512], dropout=[0.0, 0.0, 0.3, 0.4, 0.5], conv_dims=dims,
         class_num=2, bilinear=False, num_domains=2, precision="bf16")
torch.manual_seed(1)
net = fplx.UNet2D5_dsbn(p).cuda()
net._ensure_flat()
net.train()
g = torch.Generator().manual_seed(0)
x = torch.randn(*shape, generator=g).cuda()
n, _, D, H, W = shape
lab = torch.zeros(n, 2, D, H, W)
lab[:, 0] = 1.0
lab[:, 0, D // 4:D // 2, H // 4:H // 2, W // 3:2 * W // 3] = 0.0
lab[:, 1, D // 4:D // 2, H // 4:H // 2, W // 3:2 * W // 3] = 1.0
lab = lab.cuda()
eng = net.engine
logits, sv = eng.forward(x, 0, True, net.dropout_active(), 1, 0, keep=True)
c = logits.shape[1]
v = logits[0, 0].numel()
part = torch.empty((n, ops.loss_rows(v), ops.loss_k(c)), dtype=torch.float32, device="cuda")
coef = torch.empty(n * c * 2 + 2, dtype=torch.float32, device="cuda")
out = torch.empty(4 + c, dtype=torch.float32, device="cuda")
terms = (1.0, 0.0, 0.0, 0.0)
ops.seg_loss_fwd(logits, lab, None, None, terms, True, part, out, coef)
dlogits = torch.empty_like(logits)
one = torch.ones(1, device="cuda")
ops.seg_loss_bwd(logits, lab, None, coef, one, terms, True, dlogits)
torch.cuda.synchronize()


def bwd(side, joins, taps=None):
    eng.use_side_stream = side
    eng.block_joins = joins
    eng.debug_tap = None if taps is None else (lambda name, t: taps.append((name, t.clone())))
    gf = torch.empty_like(net.flat_params)
    eng.backward(sv, dlogits.clone(), gf)
    torch.cuda.synchronize()
    eng.debug_tap = None
    return gf


ref = bwd(False, False)
ref2 = bwd(False, False)
print("single stream reproducible:", bool(torch.equal(ref, ref2)))
for side, joins in ((True, True), (True, False)):
    bad = {}
    for r in range(reps):
        gf = bwd(side, joins)
        if torch.equal(gf, ref):
            continue
        for k, (o, cnt, shp) in net._layout.items():
            a, b = gf[o:o + cnt], ref[o:o + cnt]
            if not torch.equal(a, b):
                d = float((a - b).abs().max())
                e = bad.setdefault(k, [0, 0.0, float(b.abs().max())])
                e[0] += 1
                e[1] = max(e[1], d)
    print("side=%d joins=%d: %d parameter tensors differ in some of %d runs" % (side, joins, len(bad), reps))
    for k, e in sorted(bad.items(), key=lambda kv: -kv[1][0]):
        print("   %-40s runs %3d  max|diff| %.3e  (max|ref| %.3e)" % (k, e[0], e[1], e[2]))

# intermediate data gradients (main stream): first tensor that differs from the single-stream run
t_ref, t_run = [], []
bwd(False, False, t_ref)
for r in range(5):
    t_run = []
    bwd(True, False, t_run)
    diffs = [(a[0], int((a[1].float() != b[1].float()).sum()), float((a[1].float() - b[1].float()).abs().max()))
             for a, b in zip(t_ref, t_run) if not torch.equal(a[1], b[1])]
    print("run %d with taps: %d of %d intermediate tensors differ; first: %s" % (r, len(diffs), len(t_ref), diffs[:3]))

# discriminator: same schedule, same kernels, but nothing allocated during backward is freed before the final sync
_orig_empty, hold = torch.empty, []


def _empty_hold(*a, **k):
    t = _orig_empty(*a, **k)
    hold.append(t)
    return t


torch.empty = _empty_hold
nbad = 0
for r in range(10):
    hold.clear()
    nbad += int(not torch.equal(bwd(True, False), ref))
torch.empty = _orig_empty
print("side=1 joins=0 with every backward allocation held until the sync: %d of 10 runs differ" % nbad)

# bisect by position: ONE join of the weight-gradient stream, at tap i only.  If kernel S (enqueued on the side stream at
# position a) must not run beside main-stream kernel M (launched at position b), exactly the joins with a < i <= b help.
names = [n for n, _ in t_ref]
main_s = torch.cuda.current_stream()


def bwd_join_at(i):
    cnt = [0]

    def tap(name, t):
        if cnt[0] == i:
            main_s.wait_stream(eng._side)
        cnt[0] += 1
    eng.use_side_stream, eng.block_joins, eng.debug_tap = True, False, tap
    gf = torch.empty_like(net.flat_params)
    eng.backward(sv, dlogits.clone(), gf)
    torch.cuda.synchronize()
    eng.debug_tap = None
    return gf


for i, nm in enumerate(names):
    bad = sum(int(not torch.equal(bwd_join_at(i), ref)) for _ in range(3))
    print("join only at tap %2d (%-28s): %d of 3 runs differ" % (i, nm, bad))

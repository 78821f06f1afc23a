"""Find the float32 cut points of the FPL boundary test (reference agent_seg.py:922-924):

    uncertainty = -1.0 * (means * np.log(means + 1e-6));  boundary = (uncertainty > 0.01).sum()

`means` is float32, so the predicate is a function of the 2^30 float32 values in [0, 1] alone.  It is evaluated with
numpy (the reference's arithmetic) by bisection on the float32 bit pattern, then EXHAUSTIVELY over +-WINDOW bit
patterns around each crossing to prove the predicate is a clean step there; a coarse sweep over all of [0, 1] (every
STRIDE-th pattern) shows there are exactly two crossings.  Output: the inclusive bit range [LO, HI] with
predicate(m) == (LO <= bits(m) <= HI), pasted into csrc/loss_filter.hip (FPL_CUT_LO / FPL_CUT_HI)."""
import numpy as np

WINDOW, STRIDE = 1 << 16, 1 << 8


def pred(bits):
    m = np.asarray(bits, np.uint32).view(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        u = -1.0 * (m * np.log(m + 1e-6))
    assert u.dtype == np.float32
    return u > 0.01


one = int(np.float32(1.0).view(np.uint32))
coarse = pred(np.arange(0, one + 1, STRIDE, dtype=np.uint32))
flips = np.nonzero(coarse[1:] != coarse[:-1])[0]
assert len(flips) == 2 and not coarse[0] and not coarse[-1], flips
cuts = []
for f in flips:
    lo, hi = int(f) * STRIDE, (int(f) + 1) * STRIDE          # pred(lo) != pred(hi)
    plo = bool(pred([lo])[0])
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if bool(pred([mid])[0]) == plo:
            lo = mid
        else:
            hi = mid
    w = np.arange(max(0, lo - WINDOW), min(one, hi + WINDOW) + 1, dtype=np.uint32)
    pw = pred(w)
    nfl = int(np.count_nonzero(pw[1:] != pw[:-1]))
    assert nfl == 1, "predicate flickers around bit pattern %d (%d flips)" % (lo, nfl)
    cuts.append((lo, hi, plo))
(l0, h0, p0), (l1, h1, p1) = cuts
assert (not p0) and p1
LO, HI = h0, l1
print("predicate true for bits in [0x%08X, 0x%08X]  =  m in [%.9g, %.9g]" % (
    LO, HI, np.uint32(LO).view(np.float32), np.uint32(HI).view(np.float32)))

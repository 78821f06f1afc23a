#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV (both streams on) -> where the train step's wall time goes: per queue busy time, the
time no kernel runs at all, the time exactly one / two kernels run, and the largest idle gaps with the kernels around them.

usage: timeline_gaps.py <kernel_trace.csv> [first_fraction_to_skip]"""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), name[:44]))
rows.sort()
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
# steady state: the last nsteps train steps, delimited by the first kernel of a forward pass (the stem convolution: one per
# step).  (The optimiser launches no longer mark a step's end: since round 5 the shared segment's Adam runs in pieces during
# backward - train.py AdamBehindBackward.)
stem = [s for s, e, q, n in rows if n.startswith("stem_fwd")]
if len(stem) > nsteps:
    t0, t1 = stem[-nsteps - 1], stem[-1] - 1
else:
    adam = [e for s, e, q, n in rows if n.startswith("adam_k")]
    t0, t1 = adam[-2 * nsteps - 1], adam[-1]
rows = [r for r in rows if r[0] >= t0 and r[1] <= t1]
span = rows[-1][1] - rows[0][0]
per_q = defaultdict(int)
for s, e, q, n in rows:
    per_q[q] += e - s
ev = []
for s, e, q, n in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
depth, last, hist = 0, ev[0][0], defaultdict(int)
for t, d in ev:
    hist[min(depth, 3)] += t - last
    last = t
    depth += d
print("window %.3f ms = %d steps of %.3f ms, %d kernels" % (span / 1e6, nsteps, span / 1e6 / nsteps, len(rows)))
for q, b in sorted(per_q.items(), key=lambda kv: -kv[1]):
    print("  queue %s busy %.3f ms (%.1f %%)" % (q, b / 1e6, 100.0 * b / span))
for k in sorted(hist):
    print("  %d kernel(s) running: %.3f ms (%.1f %%)" % (k, hist[k] / 1e6, 100.0 * hist[k] / span))
# idle gaps of the whole device
gaps = []
end = rows[0][1]
prev = rows[0][3]
for s, e, q, n in rows[1:]:
    if s > end:
        gaps.append((s - end, prev, n))
    if e > end:
        end, prev = e, n
gaps.sort(reverse=True)
print("device idle in %d gaps; by (before -> after), total us:" % len(gaps))
agg = defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    agg[(a, b)][0] += g
    agg[(a, b)][1] += 1
for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print("  %8.1f us in %3d gaps (avg %.1f)  %s -> %s" % (g / 1e3, c, g / 1e3 / c, a, b))

# the critical path: kernels of the busiest queue (the main stream) by total time in the window
mainq = max(per_q.items(), key=lambda kv: kv[1])[0]
fam = defaultdict(lambda: [0, 0])
for s_, e_, q, n in rows:
    if q == mainq:
        fam[n][0] += e_ - s_
        fam[n][1] += 1
print("main queue %s, per step:" % mainq)
for n, (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:40]:
    print("  %8.1f us  %5.1f launches  avg %7.1f  %s" % (t / 1e3 / nsteps, c / nsteps, t / 1e3 / c, n))

# who is alone on the device: time with exactly one kernel running, by queue (the second stream's exposed tail shows up here)
ev2 = []
for s_, e_, q, n in rows:
    ev2.append((s_, 1, q, n))
    ev2.append((e_, -1, q, n))
ev2.sort(key=lambda t: (t[0], t[1]))
active, last_t = {}, ev2[0][0]
alone, alone_k = defaultdict(int), defaultdict(int)
for t, d, q, n in ev2:
    if len(active) == 1:
        (qq, nn), = active.items()
        alone[qq] += t - last_t
        alone_k[(qq, nn)] += t - last_t
    last_t = t
    if d > 0:
        active[q] = n
    else:
        active.pop(q, None)
print("alone on the device, per step:")
for q, v in sorted(alone.items(), key=lambda kv: -kv[1]):
    print("  queue %s: %.1f us" % (q, v / 1e3 / nsteps))
    for (qq, nn), vv in sorted(alone_k.items(), key=lambda kv: -kv[1]):
        if qq == q and qq != mainq and vv / 1e3 / nsteps > 5:
            print("      %8.1f us  %s" % (vv / 1e3 / nsteps, nn))

# the last step in full detail around its ends: offset from the step's first kernel, duration, queue
if len(stem) > 1:
    s0, s1 = stem[-2], stem[-1]
    last = [(s_, e_, q, n) for s_, e_, q, n in rows if s0 <= s_ < s1]
    if last:
        tend = max(e_ for s_, e_, q, n in last)
        print("last step, its final 1.3 ms (start offset us, duration us, queue, kernel):")
        for s_, e_, q, n in last:
            if e_ >= tend - 1300000:
                print("  %9.1f %8.1f  q%s  %s" % ((s_ - s0) / 1e3, (e_ - s_) / 1e3, "1" if q == mainq else "2", n))

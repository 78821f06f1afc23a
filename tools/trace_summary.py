#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per (kernel, grid size) launch count / average / min / max duration.
Separates the shapes one kernel template is launched with (the --stats summary averages them together).

usage: trace_summary.py <kernel_trace.csv> <out.csv>
"""
import csv
import sys
from collections import defaultdict


def main():
    acc = defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1])):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        acc[(name, str(grid))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = []
    for (name, grid), d in acc.items():
        rows.append((sum(d), name, grid, len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    with open(sys.argv[2], "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_size", "launches", "avg_us", "min_us", "max_us", "share"])
        for t, name, grid, n, avg, mn, mx in rows:
            w.writerow([name, grid, n, "%.2f" % avg, "%.2f" % mn, "%.2f" % mx, "%.4f" % (t / tot)])
    for t, name, grid, n, avg, mn, mx in rows[:12]:
        print("%-40s grid %9s  n %4d  avg %8.1f us  share %.3f" % (name[:40], grid, n, avg, t / tot))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs, CSV output) into
per-kernel HBM traffic per launch, with the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md
(section HBM): FETCH_SIZE/WRITE_SIZE are in KiB-units of 1024 B; FETCH_SIZE reports exactly 1/2 of the bytes
of a wide coalesced streaming read on gfx950 -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import csv
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hashes():
    """sha256 of every kernel source: bench.py reports `roofline.traffic` from this file only while they still match"""
    d = os.path.join(ROOT, "fpl-plus_amd", "csrc")
    return {f: hashlib.sha256(open(os.path.join(d, f), "rb").read()).hexdigest() for f in sorted(os.listdir(d))
            if f.endswith((".hip", ".h"))}


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        k = r["Kernel_Name"] + "|grid=" + r["Grid_Size"]
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def short(name):
    base, grid = name.rsplit("|", 1)
    base = base.replace("(anonymous namespace)::", "").replace("void ", "")
    return base.split("(")[0][:60] + "|" + grid


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        fetch = f.get(k, (0.0, 0))[0] * 1024.0 * 2.0       # KiB units, gfx950 x2 correction
        write = w.get(k, (0.0, 0))[0] * 1024.0
        out[short(k)] = {"fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                         "hbm_bytes_per_launch": fetch + write, "launches": max(f.get(k, (0, 0))[1], w.get(k, (0, 0))[1])}
    listing = sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])
    out["_meta"] = {"sources_sha256": source_hashes(), "command": os.environ.get("PMC_COMMAND", ""),
                    "total_hbm_bytes": sum(v["hbm_bytes_per_launch"] * v["launches"] for _, v in listing)}
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in listing[:12]:
        print("%-80s %4d launches  %8.1f MB fetch  %8.1f MB write" % (k[:80], v["launches"], v["fetch_bytes_per_launch"] / 1e6,
                                                                      v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()

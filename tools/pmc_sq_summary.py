#!/usr/bin/env python3
"""Summarise rocprofv3 SQ / GRBM counter passes (one or more `--pmc ...` runs of the same command, CSV output) per kernel:
average counter values per launch plus the derived figures north_star asks for (MFMA utilisation) and the ones the
kernels are tuned by.  Units (MI355X_MICROARCH.md, cycle-constants table): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs (= 32 per
v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE is summed over the 8 XCDs.

    mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)        busy share of the matrix pipes
    mfma_pf     = SQ_VALU_MFMA_BUSY_CYCLES / 32 * 32768 flop / duration                 (if timestamps are present)
    clock_ghz   = GRBM_GUI_ACTIVE / 8 / duration
    wait_share  = SQ_WAIT_ANY / SQ_WAVE_CYCLES, issue_stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, active = SQ_ACTIVE_INST_ANY / ..

usage: pmc_sq_summary.py out.json pass1_counter_collection.csv [pass2.csv ...]
"""
import csv
import json
import sys
from collections import defaultdict


def short(name, grid):
    base = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return base.split("(")[0][:70] + "|grid=" + grid


def main():
    out_path, paths = sys.argv[1], sys.argv[2:]
    tot = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    dur = defaultdict(list)
    for p in paths:
        seen = set()
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"], r["Grid_Size"])
            c = r["Counter_Name"]
            tot[k][c] += float(r["Counter_Value"])
            cnt[k][c] += 1
            did = (p, r.get("Dispatch_Id"))
            if "Start_Timestamp" in r and did not in seen:
                seen.add(did)
                try:
                    dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
                except (TypeError, ValueError):
                    pass
    out = {}
    for k in tot:
        e = {c: tot[k][c] / cnt[k][c] for c in tot[k]}
        e["launches"] = max(cnt[k].values())
        if dur[k]:
            e["duration_us_profiled"] = sum(dur[k]) / len(dur[k]) / 1e3
        g = e.get("GRBM_GUI_ACTIVE")
        mf = e.get("SQ_VALU_MFMA_BUSY_CYCLES")
        wc = e.get("SQ_WAVE_CYCLES")
        if g and mf is not None:
            e["mfma_util"] = mf / (1024.0 * g / 8.0)
        if g and "duration_us_profiled" in e:
            e["clock_ghz"] = g / 8.0 / (e["duration_us_profiled"] * 1e3)
        if mf is not None and "duration_us_profiled" in e:
            e["mfma_pflops_if_32x32x16"] = mf / 32.0 * 32768.0 / (e["duration_us_profiled"] * 1e-6) / 1e15
        if wc:
            for c, nm in (("SQ_WAIT_ANY", "wait_share"), ("SQ_WAIT_INST_ANY", "issue_stall_share"),
                          ("SQ_ACTIVE_INST_ANY", "active_share"), ("SQ_WAIT_INST_LDS", "lds_issue_stall_share")):
                if c in e:
                    e[nm] = e[c] / wc
        if e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_share"] = e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"]
        out[k] = e
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    key = lambda kv: -kv[1].get("duration_us_profiled", 0.0) * kv[1]["launches"]
    print("%-72s %5s %9s %6s %6s %6s %6s %6s" % ("kernel|grid", "n", "us", "mfma", "GHz", "wait", "stall", "ldscf"))
    for k, e in sorted(out.items(), key=key)[:40]:
        f = lambda n: ("%6.3f" % e[n]) if n in e else "     -"
        print("%-72s %5d %9.1f %s %s %s %s %s" % (k[:72], e["launches"], e.get("duration_us_profiled", 0.0), f("mfma_util"),
                                              f("clock_ghz"), f("wait_share"), f("issue_stall_share"), f("lds_conflict_share")))


if __name__ == "__main__":
    main()

"""Paired report from saved arms of the Dice proxy: an fplx arm (tools/dice_proxy.py --save-json, GPU box) against the reference
arm (tools/dice_proxy_refarm.py, the reference's own training on the CPU of the build container) over the batch orders both
hold.  Runs anywhere (no GPU, no reference).

    python tools/dice_proxy_merge.py <fplx arms json>[,<more json of later orders>] <reference arm json> [arm=bf16] [out.txt]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fpl-plus_amd"))
import numpy as np  # noqa: E402


def report(name, a, b):
    d = b.mean(1) - a.mean(1)
    n = len(d)
    se = d.std(ddof=1) / np.sqrt(n)
    pv = (b - a).reshape(-1)
    q = np.percentile(pv, [5, 25, 50, 75, 95])
    return ["%s over %d batch orders (paired: same initial weights, same batches in the same order, same held-out volumes):" % (name, n),
            "  per-order difference of the mean Dice: %s" % " ".join("%+.2f" % v for v in d),
            "  mean %+.3f points, std %.3f, standard error %.3f -> |mean| + 2 SE = %.3f (north_star tolerance 0.5)"
            % (d.mean(), d.std(ddof=1), se, abs(d.mean()) + 2 * se),
            "  per-volume differences (%d): 5/25/50/75/95 %% = %+.2f %+.2f %+.2f %+.2f %+.2f, max |.| %.2f"
            % (pv.size, q[0], q[1], q[2], q[3], q[4], np.abs(pv).max()),
            "  -> by the |mean| + 2 SE < 0.5 rule: %s" % ("MET" if abs(d.mean()) + 2 * se < 0.5 else "NOT RESOLVED at this number of orders"),
            "  -> equivalence test at +-0.5 points (two one-sided tests, 5 %% each): mean -+ 1.645 SE = [%+.3f, %+.3f] -> %s"
            % (d.mean() - 1.645 * se, d.mean() + 1.645 * se,
               "inside +-0.5: equivalent at the 5 % level" if (d.mean() - 1.645 * se > -0.5 and d.mean() + 1.645 * se < 0.5)
               else "not inside +-0.5")]


def main():
    rf = json.load(open(sys.argv[2]))
    arm = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    rc = rf["config"]
    by_seed = {}
    for path in sys.argv[1].split(","):
        fx = json.load(open(path))
        fc = fx["config"]
        assert (fc["base"], fc["dims"], list(fc["shape"]), fc["iters"], fc["held_out"]) == \
               (rc["base"], rc["dims"], list(rc["shape"]), rc["iters"], rc["held_out"]), (fc, rc)
        for i, row in enumerate(fx["dice_percent"][arm]):
            by_seed[fc.get("first_seed", 0) + i] = row
    seeds = sorted(s for s in by_seed if str(s) in rf["orders"])
    got = {s: by_seed[s] for s in seeds}
    ref = np.asarray([rf["orders"][str(s)]["dice_percent"] for s in seeds])
    got = np.asarray([got[s] for s in seeds])
    seeds = list(range(len(seeds)))
    out = ["config: %s" % json.dumps(rc),
           "REFERENCE arm: %s" % rf["what"],
           "  Dice %% over orders: %s (mean %.2f, std %.2f)" % (" ".join("%.2f" % v for v in ref.mean(1)), ref.mean(), ref.mean(1).std()),
           "fplx %s arm (tools/dice_proxy.py on the MI355X):" % arm,
           "  Dice %% over orders: %s (mean %.2f, std %.2f)" % (" ".join("%.2f" % v for v in got[seeds].mean(1)), got[seeds].mean(),
                                                             got[seeds].mean(1).std())]
    out += report("fplx %s - REFERENCE" % arm, ref, got[seeds])
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(text + "\n")


if __name__ == "__main__":
    main()

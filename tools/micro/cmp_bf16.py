"""compare two raw bf16 dumps of tools/micro/march_bench (MB_DUMP) and their statistics rows"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], np.uint16).astype(np.uint32) << 16
b = np.fromfile(sys.argv[2], np.uint16).astype(np.uint32) << 16
a, b = a.view(np.float32), b.view(np.float32)
d = np.abs(a - b)
print("outputs: %d values, max |a| %.3f, max |diff| %.4g, mean |diff| %.3g, values differing %d (bf16 ulp flips are expected: different summation order)"
      % (a.size, np.abs(a).max(), d.max(), d.mean(), int((d > 0).sum())))
sa, sb = np.fromfile(sys.argv[1] + ".stats", np.float32), np.fromfile(sys.argv[2] + ".stats", np.float32)
print("statistics rows: max rel diff %.3g" % (np.abs(sa - sb).max() / np.abs(sa).max()))
assert d.max() <= 0.02 * np.abs(a).max() and (d > 0.004 * np.abs(a).max()).mean() < 0.01

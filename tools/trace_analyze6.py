"""Lag between consecutive strips at three points of the sweep (library built with -DPK16_TRACE_ABS)."""
import sys, numpy as np
t = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 4)
S = int(sys.argv[2])
t = t[:S] / 100.0
for lo, hi in [(1, 100), (100, 1000), (1100, min(2000, S))]:
    if hi > lo + 1:
        for name, col in (("chunk 1", 2), ("chunk 1000", 3), ("end", 1)):
            d = np.diff(t[lo:hi, col])
            print("strips %4d..%4d  lag at %-10s mean %.1f us  (p10 %.1f p90 %.1f)" % (lo, hi, name, d.mean(), np.percentile(d, 10), np.percentile(d, 90)))

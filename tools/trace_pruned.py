"""Per-strip timeline of a pruned run from the kernel trace: python tools/trace_pruned.py trace.bin strips_total
(start, end per strip in 10 ns ticks; tracing switches the hot chunk loop off).  Prints how the strips' lifetimes add up."""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 4)
S = int((t[:, 0] != 0).sum())
us = t[:S, :2] / 100.0
us -= us[:, 0].min()
dur = (us[:, 1] - us[:, 0]) / 1e3                     # ms
span = us[:, 1].max() / 1e3
print("%d strips, kernel span %.0f ms, sum of strip lifetimes %.0f ms = %.0f wavefronts busy on average" % (S, span, dur.sum(), dur.sum() / span))
for lo, hi in ((0, S // 8), (S // 8, S // 2), (S // 2, 7 * S // 8), (7 * S // 8, S)):
    d = dur[lo:hi]
    lag = np.diff(us[lo:hi, 1]) if hi - lo > 2 else np.zeros(1)
    print("strips %6d..%6d: lifetime mean %.1f ms (p10 %.1f, p90 %.1f); end-to-end lag mean %.1f us" % (lo, hi, d.mean(), np.percentile(d, 10), np.percentile(d, 90), lag.mean()))

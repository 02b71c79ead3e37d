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
# chunk 1000 (column 64 000) lies left of the band for every strip below row ~100 000: where the general body handled it (and
# not a fast-forward run) its four phases are those of a skipped slab
ph = t[S // 4:S, 3]
ph = ph[ph != 0]
if len(ph):
    parts = [((ph >> sh) & 0xffff) / 100.0 for sh in (0, 16, 32, 48)]
    tot = sum(parts)
    print("chunk 1000 of %d strips below the first quarter, handled by the chunk body: input %.2f + stage %.2f + compute %.2f + tail %.2f = %.2f us (p10 %.2f, p90 %.2f)"
          % (len(ph), parts[0].mean(), parts[1].mean(), parts[2].mean(), parts[3].mean(), tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90)))

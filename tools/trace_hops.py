"""Start-to-start lag of consecutive strips and their sweep times from a kernel trace (MI355SW_TRACE=<file>: per strip
[start, end, phases of chunk 1, phases of chunk 1000], s_memrealtime ticks of 10 ns; tracing switches the hot chunk loop off).
python tools/trace_hops.py trace.bin [strips to look at, default 1200]"""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 4)
S = min(int(sys.argv[2]) if len(sys.argv) > 2 else 1200, (t[:, 0] != 0).sum())
us = t[:S, :2] / 100.0
us -= us[0, 0]
for lo, hi in ((1, 64), (64, 512), (512, 1023), (1024, S)):
    if hi > lo + 2:
        lag = np.diff(us[lo:hi, 0])
        dur = us[lo:hi, 1] - us[lo:hi, 0]
        print("strips %5d..%5d: start lag mean %.1f us (p10 %.1f, p50 %.1f, p90 %.1f), sweep %.1f ms, strip %d starts at %.2f ms" % (
            lo, hi, lag.mean(), np.percentile(lag, 10), np.percentile(lag, 50), np.percentile(lag, 90), dur.mean() / 1e3, hi - 1, us[hi - 1, 0] / 1e3))
ph = t[1:min(S, 512), 2]
parts = [((ph >> sh) & 0xffff) / 100.0 for sh in (0, 16, 32, 48)]
print("chunk 1 of strips 1..%d, phases (us): input wait %.2f, stage %.2f, compute %.2f, tail %.2f" % ((min(S, 512) - 1,) + tuple(float(p.mean()) for p in parts)))
ph = t[1:min(S, 512), 3]
if (ph != 0).any():
    parts = [((ph >> sh) & 0xffff) / 100.0 for sh in (0, 16, 32, 48)]
    print("chunk 1000, phases (us): input wait %.2f, stage %.2f, compute %.2f, tail %.2f" % tuple(float(p.mean()) for p in parts))

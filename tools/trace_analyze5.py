"""Strip timeline from MI355SW_TRACE: python tools/trace_analyze5.py trace.bin nstrips"""
import sys, numpy as np
t = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 4)
S = int(sys.argv[2])
st, en = t[:S, 0] / 100.0, t[:S, 1] / 100.0      # us (100 MHz realtime counter)
t0 = st.min()
st -= t0; en -= t0
dur = en - st
print("strips %d  total %.1f ms" % (S, en.max() / 1e3))
for lo, hi in [(0, 10), (10, 100), (100, 1000), (1000, 1100), (1100, 2000), (2000, 2100), (2100, S)]:
    if hi > lo and lo < S:
        hi = min(hi, S)
        print("strips %4d..%4d: start %.1f..%.1f ms  dur mean %.1f ms (min %.1f max %.1f)  hop %.1f us" % (
            lo, hi, st[lo] / 1e3, st[hi - 1] / 1e3, dur[lo:hi].mean() / 1e3, dur[lo:hi].min() / 1e3, dur[lo:hi].max() / 1e3,
            np.diff(st[lo:hi]).mean() if hi - lo > 1 else 0))
print("sum of strip durations / (1024 * total) = %.3f" % (dur.sum() / (1024 * en.max())))

import sys, numpy as np
t = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 4)
P = int(sys.argv[2])
mid, mid1k, end = t[:,3], t[:,0], t[:,1]
tstep = (mid1k - mid) / 100.0 / 64000.0   # us per step, measured over 1000 chunks at mid-strip
d = np.diff(mid) / 100.0                    # us between consecutive strips reaching the middle chunk
for lo, hi in [(100, 1000), (1000, 3000), (3000, min(P, len(t)) - 1)]:
    if hi > lo:
        print("strips %d..%d: t_step %.3f us  delay/hop %.1f us  => lag %.0f columns" % (lo, hi, tstep[lo:hi].mean(), d[lo:hi].mean(), d[lo:hi].mean() / tstep[lo:hi].mean()))

import sys, numpy as np
t = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 4)
start, end, c0, pub2 = t[:,0], t[:,1], t[:,2], t[:,3]
for lo, hi in [(10, 100), (100, 1000), (1000, 3000), (3000, 4090)]:
    sl = slice(lo, hi)
    print("strips %4d..%4d: chunk0 (poll ok -> publish) %.1f us | poll ok -> chunk2 published %.1f us | hop (start(s)-start(s-1)) %.1f us | start(s) - pub2(s-1) %.1f us" % (
        lo, hi, c0[sl].mean()/100.0, ((pub2-start)[sl]).mean()/100.0, np.diff(start)[lo-1:hi-1].mean()/100.0, (start[lo:hi]-pub2[lo-1:hi-1]).mean()/100.0))

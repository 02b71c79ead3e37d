import sys, numpy as np
t = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 4)
for name, col in (("chunk 1", 2), ("chunk 1000", 3)):
    v = t[:, col].astype(np.uint64)
    parts = [((v >> np.uint64(16 * k)) & np.uint64(0xffff)).astype(np.float64) / 100.0 for k in range(4)]
    for lo, hi in [(10, 100), (300, 1000), (1100, 2000), (2100, 2900)]:
        print("%s strips %4d..%4d: poll %.1f us | load+stage %.1f us | compute %.1f us | store+drain+flag %.1f us" % (
            name, lo, hi, parts[0][lo:hi].mean(), parts[1][lo:hi].mean(), parts[2][lo:hi].mean(), parts[3][lo:hi].mean()))

"""forced-int32 throughput (flags=2): python tools/int32_perf.py  -- unrelated and related 4 M x 3 M SW, and the generic-compare kernel"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
m, n = 4000000, 3000000
CASES = (("unrelated", 2, 0), ("related", 2, 0), ("related", 2, 4), ("related", 2, 8), ("related", 2, 16), ("unrelated", 3, 0))
if len(sys.argv) > 1:
    CASES = tuple((c.split(",")[0], int(c.split(",")[1]), int(c.split(",")[2])) for c in sys.argv[1:])
for kind, flags, R in CASES:
    s0, s1 = (pkg.seqgen.related_pair if kind == "related" else pkg.seqgen.unrelated_pair)(m, n, cfg=5)
    al = pkg.MI355Aligner(device=0, flags=flags, rows_per_lane=R)
    al.setSequences(s0, s1)
    for rep in range(2):
        al.streamBegin(pkg.Partition(0, 0, m, n))
        while not al.streamPoll()[1]:
            time.sleep(0.002)
        best, _ = al.streamEnd()
        st = al.getStatistics()
    print("%s flags=%d kernel=%d strip_rows=%d waves=%d kernel_ms=%.1f GCUPS=%.1f best=%s" % (
        kind, flags, st["profile_kernel"], st["strip_rows"], st["waves"], st["kernel_ms"], m * n / st["kernel_ms"] / 1e6, best), flush=True)
    al.close()

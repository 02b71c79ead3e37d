"""What does the host-streamed first column cost a stage-2-shaped partition?  The same tall NW partition, stopped after
`stop` rows, with the gap-initialised first column (a) streamed from pinned host memory as mi355sw_align_partition does
for every non-zero border, (b) resident on the device.  python tools/stage2_probe2.py m n stop [rows_per_lane]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
m, n, stop = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
R = int(sys.argv[4]) if len(sys.argv) > 4 else 0
s0, s1 = pkg.seqgen.related_pair(m, n, cfg=9)
INF = pkg.engine.INF
col = np.zeros((m + 1, 2), dtype=np.int32); col[:, 0] = -2 * np.arange(m + 1) - 3; col[0, 0] = 0; col[:, 1] = -INF
al = pkg.MI355Aligner(device=0, rows_per_lane=R)
al.setSequences(s0, s1)
part = pkg.Partition(0, 0, m, n)
for mode in ("streamed", "resident", "streamed", "resident"):
    kw = dict(recurrence_type=pkg.NEEDLEMAN_WUNSCH, track_best=False, first_row_init_type=pkg.INIT_WITH_GAPS, want_last_column=True,
              first_column_init_type=pkg.INIT_WITH_CUSTOM_DATA)
    t0 = time.time()
    if mode == "streamed":
        al.streamBegin(part, first_column=col[:1], stream_first_column=True, **kw)
        fed = 0
    else:
        al.streamBegin(part, first_column=col, **kw)
        fed = m
    t1 = time.time()
    while True:
        if fed < m:
            ln = min(65536, m - fed)
            al.streamFeedColumn(fed, col[1 + fed:1 + fed + ln]); fed += ln
        rows, fin = al.streamPoll()
        if rows >= stop or fin:
            break
    t2 = time.time()
    al.streamAbort()
    al.streamEnd()
    st = al.getStatistics()
    print("%-8s strip_rows=%d: begin %.1f ms, %d rows after %.1f ms, kernel %.1f ms" % (mode, st["strip_rows"], (t1 - t0) * 1e3, rows, (t2 - t1) * 1e3, st["kernel_ms"]), flush=True)
al.close()

"""NW (global, gap-initialised borders) and forced-int32 SW throughput: python tools/gpu_perf_nw.py m n"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
m, n = int(sys.argv[1]), int(sys.argv[2])
RR = int(sys.argv[3]) if len(sys.argv) > 3 else 0
only = sys.argv[4] if len(sys.argv) > 4 else ""
s0, s1 = pkg.seqgen.related_pair(m, n, cfg=5)
for name, kw, flags in (("NW ++ (int32)", dict(recurrence_type=pkg.NEEDLEMAN_WUNSCH, first_row_init_type=pkg.INIT_WITH_GAPS, first_column_init_type=pkg.INIT_WITH_GAPS, track_best=False, want_last_row=True), 0),
                        ("SW related (pk16, rebasing)", dict(), 0), ("SW related (int32 forced)", dict(), 2)):
    if only and only not in name: continue
    al = pkg.MI355Aligner(device=0, flags=flags, rows_per_lane=RR)
    al.setSequences(s0, s1)
    part = pkg.Partition(0, 0, m, n)
    for it in range(2):
        al.streamBegin(part, **kw)
        while True:
            rows, fin = al.streamPoll()
            if fin: break
            time.sleep(0.002)
        last = al.streamReadLastRow(n - 1, 1) if kw.get("want_last_row") else None
        best, _ = al.streamEnd()
        st = al.getStatistics()
    print("%-28s k=%d R=%d waves=%d kernel_ms=%.1f GCUPS=%.1f best=%s last=%s" % (name, st["profile_kernel"], st["strip_rows"] // 64, st["waves"], st["kernel_ms"], m * n / st["kernel_ms"] / 1e6, best, None if last is None else last.tolist()), flush=True)
    al.close()

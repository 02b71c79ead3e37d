"""A related M x N pair, local SW through the stream API: unpruned, pruned with the pruning window (the default), pruned without
it (MI355SW_F_NO_WINDOW) -- kernel time, skipped fraction, best cell.  python tools/window_probe.py M N [R] [nw]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
from masa_cudalign_amd.engine import F_NO_WINDOW, NEEDLEMAN_WUNSCH, SMITH_WATERMAN, INIT_WITH_GAPS, INIT_WITH_ZEROES

m, n = int(sys.argv[1]), int(sys.argv[2])
R = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nw = len(sys.argv) > 4 and sys.argv[4] == "nw"
s0, s1 = pkg.seqgen.related_pair(m, n, cfg=5)
part = pkg.Partition(0, 0, m, n)
runs = (("unpruned", 0, False), ("window", 0, True), ("no window", F_NO_WINDOW, True), ("window", 0, True), ("no window", F_NO_WINDOW, True))
if os.environ.get("PROBE_WINDOW_ONLY"):
    runs = (("window", 0, True), ("window", 0, True))
for name, flags, prune in runs:
    al = pkg.MI355Aligner(device=0, rows_per_lane=R, flags=flags)
    al.setSequences(s0, s1)
    t0 = time.time()
    if nw:
        al.streamBegin(part, recurrence_type=NEEDLEMAN_WUNSCH, track_best=False, first_row_init_type=INIT_WITH_GAPS, first_column_init_type=INIT_WITH_GAPS,
                       want_last_row=True, prune_blocks=prune)
    else:
        al.streamBegin(part, prune_blocks=prune)
    while True:
        rows, fin = al.streamPoll()
        if fin:
            break
        time.sleep(0.002)
    h = int(al.streamReadLastRow(col=n - 1, length=1)[0, 0]) if nw else None
    best, _ = al.streamEnd()
    st = al.getStatistics()
    print("%-10s R=%d kernel %.1f ms seed %.1f ms wall %.2f s  GCUPS(m*n, kernel+seed) %.0f  skipped %.1f%%  best %s %s  [%s]" % (
        name, st["strip_rows"] // 64, st["kernel_ms"], st["seed_ms"], time.time() - t0, m * n / (st["kernel_ms"] + st["seed_ms"]) / 1e6,
        100.0 * st["pruned_cells"] / st["cells"], best, h if nw else "", st["kernel"]), flush=True)
    al.close()

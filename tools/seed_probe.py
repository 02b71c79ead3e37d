"""The diagonal seed on its own (mi355sw_seed_bound) next to the answer: python tools/seed_probe.py M N sw|nw [cfg [nobase]]
MI355SW_VERBOSE=1 MI355SW_VERBOSE_TILES=1 prints where every tile's last row peaks."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402
pkg = g.load_package()
from masa_cudalign_amd.bands import BandRunner  # noqa: E402

m, n, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 5
s0, s1 = pkg.seqgen.related_pair(m, n, cfg=cfg)
al = pkg.MI355Aligner(device=0, rows_per_lane=int(os.environ.get("ROWS_PER_LANE", "0")))
al.setSequences(s0, s1)
part = pkg.Partition(0, 0, m, n)
t0 = time.time()
bound = al.seedBound(part, pkg.SMITH_WATERMAN if kind == "sw" else pkg.NEEDLEMAN_WUNSCH)
print("seed bound %s in %.2f s" % (bound, time.time() - t0), flush=True)
os.environ["MI355SW_NO_DIAGONAL_SEED"] = "1"          # the run below starts from the value above, or from nothing
for b in ((bound,) if "nobase" in sys.argv else (bound, None)):
    got = {}
    t0 = time.time()
    if kind == "sw":
        al.streamBegin(part, prune_blocks=True, initial_bound=b)
    else:
        al.streamBegin(part, recurrence_type=pkg.NEEDLEMAN_WUNSCH, track_best=False, first_row_init_type=pkg.INIT_WITH_GAPS,
                       first_column_init_type=pkg.INIT_WITH_GAPS, want_last_row=True, prune_blocks=True, initial_bound=b)
    while True:
        rows, fin = al.streamPoll()
        if fin:
            break
        time.sleep(0.005)
    if kind != "sw":
        got["h"] = int(al.streamReadLastRow(col=n - 1, length=1)[0, 0])
    best, _ = al.streamEnd()
    st = al.getStatistics()
    print("initial bound %s: %s, %.1f %% skipped, kernel %.0f ms" % (b, got.get("h", tuple(best)), 100.0 * st["pruned_cells"] / st["cells"], st["kernel_ms"]), flush=True)
al.close()

"""One GLOBAL alignment (NW, gap-initialised borders) of a related pair with block pruning, through the streaming form:
python tools/nw_big.py M N [out.json].  H[m][n], skipped fraction, kernel and seed time, m*n GCUPS including the seed."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
from masa_cudalign_amd.bands import BandRunner
m, n = int(sys.argv[1]), int(sys.argv[2])
s0, s1 = pkg.seqgen.related_pair(m, n, cfg=5)
al = pkg.MI355Aligner(device=0)
al.setSequences(s0, s1)
got = {}
t0 = time.time()
BandRunner(al, prune_blocks=True).run(m, 0, n, recurrence=pkg.NEEDLEMAN_WUNSCH, track_best=False, first_row_init_type=pkg.INIT_WITH_GAPS,
                                     first_col_init_type=pkg.INIT_WITH_GAPS, want_last_row=True,
                                     before_end=lambda eng: got.update(h=int(eng.streamReadLastRow(col=n - 1, length=1)[0, 0])))
dt = time.time() - t0
st = al.getStatistics()
al.close()
out = {"workload": "%dx%d related pair (seqgen cfg=5), global NW, gap-initialised borders, block pruning on" % (m, n), "h_last_cell": got["h"], "seconds": dt,
       "kernel_ms": st["kernel_ms"], "seed_ms": st["seed_ms"], "pruned_fraction": st["pruned_cells"] / float(m) / n, "kernel": st["kernel"],
       "gcups_m_n_kernel": float(m) * n / st["kernel_ms"] / 1e6, "gcups_m_n_wall": float(m) * n / dt / 1e9}
print(json.dumps(out))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)

"""One GLOBAL alignment (NW, gap-initialised borders) of a related pair with block pruning, through the streaming form:
python tools/nw_big.py M N [out.json].  First the seed on its own (mi355sw_seed_bound: a lower bound of H[m][n] from segments of
the alignment swept in band mode), then the pruned sweep of the whole matrix behind it (mi355sw_stream_end checks the result
against the bound: MI355SW_EBOUND).  H[m][n], skipped fraction, kernel and seed time, m*n GCUPS including the seed."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
m, n = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
s0, s1 = pkg.seqgen.related_pair(m, n, cfg=5)
gen_s = time.time() - t0
al = pkg.MI355Aligner(device=0, verbosity=1)
al.setSequences(s0, s1)
part = pkg.Partition(0, 0, m, n)
t0 = time.time()
seed = al.seedBound(part, pkg.NEEDLEMAN_WUNSCH)
seed_s = time.time() - t0
print("seed bound", seed, "in %.1f s" % seed_s, flush=True)
al.streamBegin(part, recurrence_type=pkg.NEEDLEMAN_WUNSCH, track_best=False, first_row_init_type=pkg.INIT_WITH_GAPS,
               first_column_init_type=pkg.INIT_WITH_GAPS, want_last_row=True, prune_blocks=True, initial_bound=seed)
last = time.time()
while True:
    rows, fin = al.streamPoll()
    if fin:
        break
    if time.time() - last > 60:
        last = time.time()
        print("  %d / %d rows after %.0f s" % (rows, m, time.time() - t0), flush=True)
    time.sleep(0.05)
h = int(al.streamReadLastRow(col=n - 1, length=1)[0, 0])
al.streamEnd()
dt = time.time() - t0
st = al.getStatistics()
al.close()
out = {"workload": "%dx%d related pair (seqgen cfg=5), global NW, gap-initialised borders, block pruning on behind the anchored seed" % (m, n),
       "generate_s": gen_s, "seed_bound": seed, "seed_seconds": seed_s, "h_last_cell": h, "h_equals_seed_bound": h == seed, "h_not_below_seed_bound": h >= seed,
       "seconds": dt, "kernel_ms": st["kernel_ms"], "pruned_fraction": st["pruned_cells"] / float(m) / n, "kernel": st["kernel"], "strip_rows": st["strip_rows"],
       "computed_cells_gcups": st["processed_cells"] / st["kernel_ms"] / 1e6,
       "gcups_m_n_kernel": float(m) * n / st["kernel_ms"] / 1e6, "gcups_m_n_with_seed": float(m) * n / dt / 1e9}
print(json.dumps(out))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)

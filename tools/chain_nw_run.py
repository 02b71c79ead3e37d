"""BASELINE config 5's shape on ONE MI355X: a GLOBAL alignment (NW, gap-initialised borders) of a related pair as a chain of
column bands whose kernels run SIDE BY SIDE (bands.InProcessChain: one process, 1024 / BANDS wavefronts per band, ports
attached inside the process), block pruning on in every band against the lower bound of H[m][n] that travels along the
chain -- next to the single partition with the same pruning.

    python tools/chain_nw_run.py M N BANDS [out.json]

What it shows: the bound reaches every band while it runs (pruned fraction per band), and the chain's H[m][n] is the single
partition's.  The time is that of eight kernels sharing one GPU, not of eight GPUs."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
from masa_cudalign_amd.bands import InProcessChain, BandRunner, band_limits  # noqa: E402


def main():
    m, n, bands = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    outfn = sys.argv[4] if len(sys.argv) > 4 else None
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=5)
    kw = dict(recurrence=pkg.NEEDLEMAN_WUNSCH, first_row_init_type=pkg.INIT_WITH_GAPS, first_col_init_type=pkg.INIT_WITH_GAPS)
    res = {"workload": "%dx%d related synthetic pair (seqgen cfg=5), global NW, gap-initialised borders, block pruning on" % (m, n), "bands": bands}
    als = [pkg.MI355Aligner(device=0, waves=1024 // bands) for _ in range(bands)]
    try:
        for a in als:
            a.setSequences(s0, s1)
        # CHAIN_SEED=0: without the diagonal seed of the whole matrix (round 4 before mi355sw_seed_bound)
        chain = InProcessChain(als, prune_blocks=True, seed_bound=os.environ.get("CHAIN_SEED", "1") != "0")
        t0 = time.time()
        best, stats = chain.run(m, band_limits(n, [1] * bands), **kw)
        dt = time.time() - t0
        res["chain"] = {"h_last_cell": best[2], "seconds": dt, "gcups_m_n": float(m) * n / dt / 1e9, "restarts": chain.restarts,
                        "initial_bound": chain.initial_bound, "seed_ms": stats[0].get("seed_ms"),
                        "pruned_fraction": sum(s["pruned_cells"] for s in stats) / (float(m) * n),
                        "band": [{"columns": s["cells"] // m, "kernel_ms": s["kernel_ms"], "pruned_fraction": s["pruned_cells"] / float(s["cells"]),
                                  "kernel": s["kernel"], "strip_rows": s["strip_rows"], "wait_ms": s["wait_ms"]} for s in stats]}
        print(json.dumps(res["chain"]), flush=True)
    finally:
        for a in als:
            a.close()
    al = pkg.MI355Aligner(device=0)
    try:
        al.setSequences(s0, s1)
        got = {}
        t0 = time.time()
        BandRunner(al, prune_blocks=True).run(m, 0, n, recurrence=pkg.NEEDLEMAN_WUNSCH, track_best=False, first_row_init_type=pkg.INIT_WITH_GAPS,
                                             first_col_init_type=pkg.INIT_WITH_GAPS, want_last_row=True,
                                             before_end=lambda eng: got.update(h=int(eng.streamReadLastRow(col=n - 1, length=1)[0, 0])))
        dt = time.time() - t0
        st = al.getStatistics()
        res["single_partition"] = {"h_last_cell": got["h"], "seconds": dt, "kernel_ms": st["kernel_ms"], "gcups_m_n": float(m) * n / st["kernel_ms"] / 1e6,
                                   "pruned_fraction": st["pruned_cells"] / float(st["cells"]), "kernel": st["kernel"], "strip_rows": st["strip_rows"]}
    finally:
        al.close()
    res["agree"] = res["single_partition"]["h_last_cell"] == res["chain"]["h_last_cell"]
    print(json.dumps(res), flush=True)
    if outfn:
        json.dump(res, open(outfn, "w"), indent=1)
    assert res["agree"], res


if __name__ == "__main__":
    main()

"""A RELATED pair at multi-GPU size on ONE MI355X: the matrix as a chain of column bands through column ports (the code
path that crosses xGMI between GPUs), block pruning ON in every band against the running best of the chain -- what the
reference gives up when it forks (M/libmasa/libmasa.cpp:1318-1321) -- next to the single partition with pruning.

    python tools/chain_related_run.py M N BANDS [out.json] [cfg]

The bands run one after the other here (one GPU), so a band starts with everything the bands before it found waiting in
its port; across GPUs they run side by side and the words travel while they run (tests/test_gpu_bands.py,
tools/chain_prune_probe.py).  Check: the chain's canonical best = the single partition's, cell and score."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
from masa_cudalign_amd.bands import band_limits, canonical_best  # noqa: E402


def wait(e):
    while not e.streamPoll()[1]:
        time.sleep(0.2)


def main():
    m, n, bands = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    outfn = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] != "-" else None
    cfg = int(sys.argv[5]) if len(sys.argv) > 5 else 4
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=cfg)
    res = {"workload": "%dx%d related synthetic pair (seqgen cfg=%d), local SW, block pruning on" % (m, n, cfg), "bands": bands, "band": []}
    lim = band_limits(n, [1] * bands)
    eng = [pkg.MI355Aligner(device=0), pkg.MI355Aligner(device=0)]
    for e in eng:
        e.setSequences(s0, s1)
    eng[0].portCreate(m); eng[1].portCreate(m)
    eng[0].portAttach(eng[1]); eng[1].portAttach(eng[0])
    corner = np.array([[0, -pkg.INF]], dtype=np.int32)
    cands, t_chain, kernel_ms, pruned = [], time.time(), 0.0, 0
    for k in range(bands):
        e, nxt = eng[k % 2], eng[(k + 1) % 2]
        kw = dict(track_best=True, last_column_port=k < bands - 1, prune_blocks=True, prune_rows=m, prune_cols=n - lim[k], share_best=True)
        if k > 0:
            kw.update(first_column_init_type=pkg.INIT_WITH_CUSTOM_DATA, first_column_port=True, first_column=corner)
        if k < bands - 1 and k > 0:
            nxt.portReset()                       # band k-1 is finished with it; band k writes it now
        t0 = time.time()
        e.streamBegin(pkg.Partition(0, lim[k], m, lim[k + 1]), **kw)
        wait(e)
        best, _ = e.streamEnd()
        st = e.getStatistics()
        cands.append(best)
        kernel_ms += st["kernel_ms"]
        pruned += st["pruned_cells"]
        rec = {"band": k, "columns": [lim[k], lim[k + 1]], "seconds": time.time() - t0, "kernel_ms": st["kernel_ms"],
               "gcups_m_n": float(m) * (lim[k + 1] - lim[k]) / st["kernel_ms"] / 1e6, "pruned_fraction": st["pruned_cells"] / float(st["cells"]),
               "kernel": st["profile_kernel"], "strip_rows": st["strip_rows"], "own_best": list(best), "running_best": list(canonical_best(cands))}
        res["band"].append(rec)
        print(json.dumps(rec), flush=True)
    chain_s = time.time() - t_chain
    res["chain"] = {"best": list(canonical_best(cands)), "seconds": chain_s, "kernel_seconds": kernel_ms / 1e3,
                    "gcups_m_n": float(m) * n / kernel_ms / 1e6, "pruned_fraction": pruned / (float(m) * n)}
    for e in eng:
        e.portClose()
    single = {}
    for prune in (True, False) if float(m) * n < 2e14 else (True,):
        t0 = time.time()
        eng[0].streamBegin(pkg.Partition(0, 0, m, n), track_best=True, prune_blocks=prune)
        wait(eng[0])
        best, _ = eng[0].streamEnd()
        st = eng[0].getStatistics()
        single["pruned" if prune else "unpruned"] = {"best": list(best), "seconds": time.time() - t0, "kernel_ms": st["kernel_ms"],
                                                     "gcups_m_n": float(m) * n / st["kernel_ms"] / 1e6, "pruned_fraction": st["pruned_cells"] / float(st["cells"])}
        print(json.dumps(single), flush=True)
    res["single_partition"] = single
    res["agree"] = all(v["best"] == res["chain"]["best"] for v in single.values())
    for e in eng:
        e.close()
    print(json.dumps({k: v for k, v in res.items() if k != "band"}), flush=True)
    if outfn:
        json.dump(res, open(outfn, "w"), indent=1)
    assert res["agree"], res


if __name__ == "__main__":
    main()

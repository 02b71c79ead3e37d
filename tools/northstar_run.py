"""The north-star matrix on ONE MI355X: 228,000,000 x 228,000,000 local SW, unrelated random ACGT (BASELINE.json
`north_star` target: >= 4000 GCUPS with a bit-exact best score), 5.2e16 cells.

    python tools/northstar_run.py [bands] [out.json] [m] [n]

The matrix is swept as a CHAIN of column bands on the one GPU, one kernel launch per band (default 14 bands of
16.3 M columns, about ten minutes each): band k stores its last column into the column port of band k+1
(mi355sw_port_attach, the same path that crosses xGMI between GPUs -- csrc/sw_kernel.h complete_strip_common), band k+1
reads it as its first column.  Two engine handles alternate; nothing but the 80-byte bookkeeping touches the host.
One JSON line per band goes to stdout / the log as it finishes, so a run that is cut short still leaves its bands.

Check: the chain's canonical best (max score, min i, min j over the bands, BestScoreList order) is recomputed by the
oracle on the 600 x 600 window that ends at the reported cell: H there equals the score and nothing in the window is
higher (an unrelated pair's best local alignment is a few dozen columns long)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
from masa_cudalign_amd.bands import band_limits, canonical_best  # noqa: E402


def main():
    bands = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    outfn = sys.argv[2] if len(sys.argv) > 2 else None
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 228000000
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 228000000
    t_all = time.time()
    s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=5)
    res = {"workload": "%dx%d unrelated random ACGT (seqgen cfg=5), local SW, best score + canonical position" % (m, n),
           "bands": bands, "generate_s": time.time() - t_all, "band": []}
    lim = band_limits(n, [1] * bands)
    eng = [pkg.MI355Aligner(device=0), pkg.MI355Aligner(device=0)]
    for e in eng:
        e.setSequences(s0, s1)
    if bands > 1:
        eng[1].portCreate(m); eng[0].portAttach(eng[1])
    if bands > 2:
        eng[0].portCreate(m); eng[1].portAttach(eng[0])
    corner = np.array([[0, -pkg.INF]], dtype=np.int32)
    cands, kernel_ms, t_chain = [], 0.0, time.time()
    for k in range(bands):
        e, nxt = eng[k % 2], eng[(k + 1) % 2]
        part = pkg.Partition(0, lim[k], m, lim[k + 1])
        kw = dict(track_best=True, last_column_port=(k < bands - 1))
        if k > 0:
            kw.update(first_column_init_type=pkg.INIT_WITH_CUSTOM_DATA, first_column_port=True, first_column=corner)
        if k < bands - 1 and k >= 1:
            nxt.portReset()                          # band k-1 is finished with it; band k writes it now
        t0 = time.time()
        e.streamBegin(part, **kw)
        while True:
            rows, fin = e.streamPoll()
            if fin:
                break
            time.sleep(0.5)
        best, _ = e.streamEnd()
        st = e.getStatistics()
        dt = time.time() - t0
        cands.append(best)
        kernel_ms += st["kernel_ms"]
        cells = float(m) * (lim[k + 1] - lim[k])
        rec = {"band": k, "columns": [lim[k], lim[k + 1]], "seconds": dt, "kernel_ms": st["kernel_ms"],
               "gcups": cells / st["kernel_ms"] / 1e6, "strip_rows": st["strip_rows"], "strips": st["strips"],
               "kernel_launches": st["kernel_launches"], "best": list(best), "running_best": list(canonical_best(cands))}
        res["band"].append(rec)
        print(json.dumps(rec), flush=True)
        if outfn:
            json.dump(res, open(outfn, "w"), indent=1)
    for e in eng:
        e.close()
    best = canonical_best(cands)
    chain_s = time.time() - t_chain
    res["best"] = {"i": best[0] + 1, "j": best[1] + 1, "score": best[2]}
    res["seconds"] = chain_s
    res["kernel_seconds"] = kernel_ms / 1e3
    res["gcups"] = float(m) * n / chain_s / 1e9
    res["gcups_kernel_time"] = float(m) * n / kernel_ms / 1e6
    oracle = g.load_oracle()
    i1, j1 = best[0] + 1, best[1] + 1
    i0, j0 = max(0, i1 - 600), max(0, j1 - 600)
    ref = oracle.stage1(s0[i0:i1], s1[j0:j1], want_last_row=True)
    res["check"] = {"oracle_window": "600x600 ending at the reported cell", "oracle_best_in_window": int(ref["best"][2]),
                    "oracle_H_at_cell": int(ref["last_row"][-1][0]),
                    "ok": bool(ref["best"][2] == best[2] and int(ref["last_row"][-1][0]) == best[2])}
    print(json.dumps({k: v for k, v in res.items() if k != "band"}), flush=True)
    if outfn:
        json.dump(res, open(outfn, "w"), indent=1)
    assert res["check"]["ok"], res["check"]


if __name__ == "__main__":
    main()

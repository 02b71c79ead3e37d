"""The north-star matrix on ONE MI355X: 228,000,000 x 228,000,000 local SW, unrelated random ACGT (BASELINE.json
`north_star` target: >= 4000 GCUPS with a bit-exact best score), 5.2e16 cells.

    python tools/northstar_run.py [bands] [out.json] [m] [n] [first_band] [last_band+1] [state_in] [state_out]

A gpurun call is capped at one hour and the matrix takes 2.3: the chain can be cut between two bands.  The call that
stops after band k-1 keeps that band's last column in pinned host memory, packs it (one byte per (H,E) pair through a
table of the pairs that occur -- an unrelated pair has a few dozen -- then LZMA in parallel chunks: ~1.8 bits per row,
48 MiB for 228 M rows) into `state_out` together with the bests so far; the next call unpacks `state_in` and hands the
column to band k as its first column.  Nothing is recomputed and nothing is approximated.

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


def _pack_chunk(b):
    import lzma
    return lzma.compress(b, preset=4)


def save_state(path, col, meta):
    """col: (m, 2) int32 (H,E) of rows 1..m; one byte per pair through a table, LZMA in parallel chunks"""
    import pickle
    from concurrent.futures import ProcessPoolExecutor
    t0 = time.time()
    H, E = col[:, 0], col[:, 1]
    if H.min() < 0 or H.max() > 255 or E.min() < -128 or E.max() > 127:
        raise RuntimeError("boundary column outside the compact range (H %d..%d, E %d..%d)" % (H.min(), H.max(), E.min(), E.max()))
    key = (H.astype(np.uint16) << 8) | (E + 128).astype(np.uint16)
    used = np.flatnonzero(np.bincount(key, minlength=65536))
    if len(used) > 256:
        raise RuntimeError("%d distinct (H,E) pairs: more than one byte per row" % len(used))
    lut = np.zeros(65536, dtype=np.uint8)
    lut[used] = np.arange(len(used), dtype=np.uint8)
    sym = lut[key]
    CH = 4 << 20
    parts = [sym[k:k + CH].tobytes() for k in range(0, len(sym), CH)]
    with ProcessPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        packed = list(ex.map(_pack_chunk, parts))
    with open(path, "wb") as f:
        pickle.dump({"meta": meta, "rows": int(len(sym)), "table": used.astype(np.uint16), "chunks": packed}, f, protocol=4)
    size = os.path.getsize(path)
    print(json.dumps({"state_out": path, "bytes": size, "bits_per_row": size * 8.0 / len(sym), "pairs": int(len(used)),
                      "pack_s": time.time() - t0}), flush=True)
    if size > 60 << 20:
        raise RuntimeError("state file of %d bytes does not fit gpurun_out's 64 MiB" % size)


def load_state(path):
    import lzma
    import pickle
    st = pickle.load(open(path, "rb"))
    sym = np.frombuffer(b"".join(lzma.decompress(c) for c in st["chunks"]), dtype=np.uint8)
    key = st["table"][sym]
    col = np.empty((st["rows"] + 1, 2), dtype=np.int32)
    col[0] = (0, -pkg.INF)                              # corner: zero first row
    col[1:, 0] = (key >> 8).astype(np.int32)
    col[1:, 1] = (key & 255).astype(np.int32) - 128
    return st["meta"], col


def main():
    bands = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    outfn = sys.argv[2] if len(sys.argv) > 2 else None
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 228000000
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 228000000
    k_first = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    k_end = int(sys.argv[6]) if len(sys.argv) > 6 else bands
    state_in = sys.argv[7] if len(sys.argv) > 7 and sys.argv[7] != "-" else None
    state_out = sys.argv[8] if len(sys.argv) > 8 and sys.argv[8] != "-" else None
    t_all = time.time()
    s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=5)
    res = {"workload": "%dx%d unrelated random ACGT (seqgen cfg=5), local SW, best score + canonical position" % (m, n),
           "bands": bands, "bands_of_this_call": [k_first, k_end], "generate_s": time.time() - t_all, "band": []}
    prior, first_col = {"cands": [], "kernel_ms": 0.0, "chain_s": 0.0, "band": []}, None
    if state_in:
        t0 = time.time()
        prior, first_col = load_state(state_in)
        assert prior["m"] == m and prior["n"] == n and prior["bands"] == bands and prior["next_band"] == k_first, prior
        res["band"] = prior["band"]
        res["unpack_s"] = time.time() - t0
    lim = band_limits(n, [1] * bands)
    eng = [pkg.MI355Aligner(device=0), pkg.MI355Aligner(device=0)]
    for e in eng:
        e.setSequences(s0, s1)
    if k_end - k_first > 1:
        eng[0].portCreate(m); eng[1].portCreate(m)
        eng[0].portAttach(eng[1]); eng[1].portAttach(eng[0])
    corner = np.array([[0, -pkg.INF]], dtype=np.int32)
    cands, kernel_ms, t_chain = [tuple(c) for c in prior["cands"]], prior["kernel_ms"], time.time()
    for k in range(k_first, k_end):
        e, nxt = eng[k % 2], eng[(k + 1) % 2]
        part = pkg.Partition(0, lim[k], m, lim[k + 1])
        to_port = k < k_end - 1                      # the last band of a call keeps its column for the state file
        kw = dict(track_best=True, last_column_port=to_port, want_last_column=(not to_port and k < bands - 1))
        if k > k_first:
            kw.update(first_column_init_type=pkg.INIT_WITH_CUSTOM_DATA, first_column_port=True, first_column=corner)
        elif first_col is not None:
            kw.update(first_column_init_type=pkg.INIT_WITH_CUSTOM_DATA, first_column=first_col)
        if to_port and k > k_first:
            nxt.portReset()                          # band k-1 is finished with it; band k writes it now
        t0 = time.time()
        e.streamBegin(part, **kw)
        while True:
            rows, fin = e.streamPoll()
            if fin:
                break
            time.sleep(0.5)
        last_col = e.streamReadColumn(0, m) if kw["want_last_column"] else None
        best, _ = e.streamEnd()
        first_col = None
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
    chain_s = prior["chain_s"] + time.time() - t_chain
    if k_end < bands:
        if state_out:
            save_state(state_out, last_col, {"m": m, "n": n, "bands": bands, "next_band": k_end, "cands": [list(c) for c in cands],
                                             "kernel_ms": kernel_ms, "chain_s": chain_s, "band": res["band"]})
        print(json.dumps({"stopped_before_band": k_end, "kernel_seconds_so_far": kernel_ms / 1e3}), flush=True)
        return
    best = canonical_best(cands)
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
    # ... and the cell round 2's kernels reported for the same pair (profiles/r02_northstar_228Mx228M.json), where that record is at hand
    try:
        old = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_northstar_228Mx228M.json")))
        if (m, n) == (228000000, 228000000) and getattr(pkg.seqgen, "GENERATOR_VERSION", 1) == 1:
            res["check"]["equals_the_cell_of_round_2s_run"] = res["best"] == old["best"]
            res["check"]["ok"] = res["check"]["ok"] and res["check"]["equals_the_cell_of_round_2s_run"]
    except (OSError, ValueError, KeyError):
        pass
    res["library_build_id"] = pkg.engine.library_build_id()
    print(json.dumps({k: v for k, v in res.items() if k != "band"}), flush=True)
    if outfn:
        json.dump(res, open(outfn, "w"), indent=1)
    assert res["check"]["ok"], res["check"]


if __name__ == "__main__":
    main()

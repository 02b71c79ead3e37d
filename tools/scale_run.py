"""Full-size runs of the BASELINE configurations that no oracle can sweep, pinned through size-independent
properties.  python tools/scale_run.py <case> [out.json]

  c3      48M x 46M related pair, local SW, special rows every ~2.4M rows: block pruning ON and OFF must report
          the same best cell; every special row of the pruned run is a lower bound of the unpruned one; the row
          maxima of rows above the best cell agree; sha256 of the unpruned rows recorded
  tall    228M x 1M unrelated SW (the north-star height; two-phase best tracking, no 2^27 limit): best cell
          confirmed by the oracle on the window that ends at it
  wide    1M x 228M unrelated SW (the north-star width: 1.8 GB bus row): same check
  c2x     3M x 3M with special rows + last row + last column requested (everything C2 can dispatch at once)
"""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

pkg = g.load_package()


def run(al, part, before_end=None, **kw):
    t0 = time.time()
    al.streamBegin(part, **kw)
    while True:
        rows, fin = al.streamPoll()
        if fin:
            break
        time.sleep(0.01)
    t_done = time.time()
    extra = before_end(al) if before_end else None      # special rows are readable while the stream is open
    t0 += time.time() - t_done
    best, nsp = al.streamEnd()
    st = al.getStatistics()
    m, n = part.getHeight(), part.getWidth()
    info = {"best": list(best), "special_rows": nsp, "kernel_ms": st["kernel_ms"], "wall_s": time.time() - t0,
            "gcups_mn": m * n / st["kernel_ms"] / 1e6, "strip_rows": st["strip_rows"], "strips": st["strips"],
            "kernel": {1: "int32", 2: "pk16"}.get(st["profile_kernel"], st["profile_kernel"]),
            "pruned_fraction": st["pruned_cells"] / st["cells"] if st["cells"] else 0.0}
    print(json.dumps(info), flush=True)
    if extra is not None:
        info["rows"] = extra
    return info


def window_check(s0, s1, best, W=600):
    """the oracle on the W x W window that ends at the reported cell (an unrelated pair's best local alignment is
    far shorter than W): H at the corner equals the score and nothing inside is higher"""
    oracle = g.load_oracle()
    i, j, score = best                      # 0-based cell index of the stream API
    i1, j1 = i + 1, j + 1
    i0, j0 = max(0, i1 - W), max(0, j1 - W)
    ref = oracle.stage1(s0[i0:i1], s1[j0:j1], want_last_row=True)
    ok = (ref["best"][2] == score) and int(ref["last_row"][-1][0]) == score
    return {"window": W, "oracle_best_in_window": int(ref["best"][2]), "oracle_H_at_cell": int(ref["last_row"][-1][0]),
            "ok": bool(ok)}


def case_unrelated(m, n, cfg, out):
    s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=cfg)
    al = pkg.MI355Aligner(device=0)
    al.setSequences(s0, s1)
    part = pkg.Partition(0, 0, m, n)
    r = run(al, part)
    al.close()
    r["check"] = window_check(s0, s1, r["best"])
    out.update(r)
    assert r["check"]["ok"], r["check"]


def case_c3(out, m=48000000, n=46000000):
    t0 = time.time()
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=3)
    out["generate_s"] = time.time() - t0
    al = pkg.MI355Aligner(device=0, max_special_bytes=64 << 30)
    al.setSequences(s0, s1)
    part = pkg.Partition(0, 0, m, n)
    interval = m // 20
    rows_h = {}
    for prune in (False, True):
        def read_rows(al, prune=prune):
            rows, k = [], 0
            while True:
                try:
                    dp, cells = al.streamReadSpecialRow(k)
                except pkg.engine.AlignerError:
                    break
                h = np.ascontiguousarray(cells[:, 0])
                rec = {"dp_row": int(dp), "max_h": int(h.max()), "argmax": int(h.argmax())}
                if not prune:
                    rec["sha256"] = hashlib.sha256(np.ascontiguousarray(cells).tobytes()).hexdigest()
                    rows_h[k] = h
                else:
                    ref = rows_h[k]
                    rec["lower_bound_of_unpruned"] = bool((h <= ref).all())
                    rec["cells_equal"] = float((h == ref).mean())
                    rec["max_equal"] = bool(int(ref.max()) == rec["max_h"])
                rows.append(rec)
                k += 1
            return rows
        r = run(al, part, before_end=read_rows, prune_blocks=prune, special_row_interval=interval)
        assert len(r["rows"]) == r["special_rows"], (len(r["rows"]), r["special_rows"])
        out["pruned" if prune else "unpruned"] = r
    al.close()
    a, b = out["unpruned"], out["pruned"]
    out["same_best"] = a["best"] == b["best"]
    bi = a["best"][0]
    out["rows_lower_bound"] = all(x["lower_bound_of_unpruned"] for x in b["rows"])
    # rows the optimal alignment crosses carry its score exactly; below the best cell nothing is promised
    out["row_max_equal_above_best"] = all(x["max_equal"] for x in b["rows"] if x["dp_row"] <= bi)
    assert out["same_best"] and out["rows_lower_bound"] and out["row_max_equal_above_best"], out


def case_c3_pruned_only(out, ref_json, m=48000000, n=46000000):
    """C3's stage 1 with pruning only (the unpruned sweep costs six minutes): best cell and the maximum of every special
    row above it against the UNPRUNED run recorded in `ref_json` (profiles/r04_scale_c3_48Mx46M.json)"""
    ref = json.load(open(ref_json))["unpruned"]
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=3)
    al = pkg.MI355Aligner(device=0, max_special_bytes=64 << 30)
    al.setSequences(s0, s1)
    part = pkg.Partition(0, 0, m, n)

    def read_rows(al):
        rows, k = [], 0
        while True:
            try:
                dp, cells = al.streamReadSpecialRow(k)
            except pkg.engine.AlignerError:
                break
            h = np.ascontiguousarray(cells[:, 0])
            rows.append({"dp_row": int(dp), "max_h": int(h.max()), "argmax": int(h.argmax())})
            k += 1
        return rows
    t0 = time.time()
    r = run(al, part, before_end=read_rows, prune_blocks=True, special_row_interval=m // 20)
    out["wall_with_seed_s"] = time.time() - t0
    st = al.getStatistics()
    out["seed_ms"] = st["seed_ms"]
    al.close()
    out["pruned"] = r
    out["gcups_mn_incl_seed"] = float(m) * n / (st["kernel_ms"] + st["seed_ms"]) / 1e6
    out["same_best"] = r["best"] == ref["best"]
    bi = ref["best"][0]
    want = {x["dp_row"]: x for x in ref["rows"]}
    out["row_max_equal_above_best"] = all(x["max_h"] == want[x["dp_row"]]["max_h"] and x["argmax"] == want[x["dp_row"]]["argmax"]
                                          for x in r["rows"] if x["dp_row"] <= bi)
    out["rows_lower_bound_of_recorded_maxima"] = all(x["max_h"] <= want[x["dp_row"]]["max_h"] for x in r["rows"])
    assert out["same_best"] and out["row_max_equal_above_best"] and out["rows_lower_bound_of_recorded_maxima"], out


def case_nw_tall(out, m=249000000, n=500000):
    """C5's height (249 M rows, beyond the reference's 134 M texture limit) as a global NW with gap-initialised
    borders: the packed kernel (window follows the scores down to -5e8) and the int32 kernel must agree on
    H[m][n] and on the whole last row"""
    from masa_cudalign_amd.engine import NEEDLEMAN_WUNSCH, INIT_WITH_GAPS
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=15)
    part = pkg.Partition(0, 0, m, n)
    res = {}
    for name, force in (("pk16", False), ("int32", True)):
        al = pkg.MI355Aligner(device=0)
        al.setSequences(s0, s1)
        def last_row(al):
            row = al.streamReadLastRow()
            return {"H_last_cell": int(row[-1, 0]), "sha256": hashlib.sha256(np.ascontiguousarray(row).tobytes()).hexdigest()}
        r = run(al, part, before_end=last_row, recurrence_type=NEEDLEMAN_WUNSCH, first_row_init_type=INIT_WITH_GAPS,
                first_column_init_type=INIT_WITH_GAPS, want_last_row=True, track_best=False, force_int32=force)
        al.close()
        res[name] = r
    out.update(res)
    out["agree"] = res["pk16"]["rows"] == res["int32"]["rows"]
    assert out["agree"], out


def case_c5band(out, m=249000000, n=28500000, k_rows=400, k_cols=48):
    """ONE GPU's share of BASELINE config C5 at full size: band 0 of the 8 column bands of the 249 M x 228 M global
    NW (all 249 M rows x 28.5 M columns, gap-initialised borders, last column kept as it would be streamed to band 1,
    last row kept).  No oracle can sweep 7e15 cells; the run is pinned on its borders and through a size-independent
    property: (1) the first k_rows cells of the last column equal the oracle's sweep of those rows over the whole
    band, (2) the first k_cols cells of the last row equal the oracle's sweep of all 249 M rows over those columns,
    (3) neighbouring cells of the last row / last column never differ by more than match + open + ext + |mismatch|
    (a global score changes by at most one edit when one letter is appended), (4) sha256 of both borders recorded."""
    from masa_cudalign_amd.engine import NEEDLEMAN_WUNSCH, INIT_WITH_GAPS
    t0 = time.time()
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=15)
    out["generate_s"] = time.time() - t0
    part = pkg.Partition(0, 0, m, n)
    al = pkg.MI355Aligner(device=0)
    al.setSequences(s0, s1)
    keep = {}

    def borders(al):
        row = al.streamReadLastRow()
        keep["row_head"] = row[:k_cols].copy()
        hs, hc, step, mx = hashlib.sha256(), hashlib.sha256(), 1 << 24, 0
        hs.update(np.ascontiguousarray(row).tobytes())
        mx_row = int(np.abs(np.diff(row[:, 0].astype(np.int64))).max()) if n > 1 else 0
        prev = None
        for r0 in range(0, m, step):
            col = al.streamReadColumn(r0, min(step, m - r0))
            if r0 == 0:
                keep["col_head"] = col[:k_rows].copy()
            hc.update(np.ascontiguousarray(col).tobytes())
            h = col[:, 0].astype(np.int64)
            if prev is not None:
                mx = max(mx, abs(int(h[0]) - prev))
            if len(h) > 1:
                mx = max(mx, int(np.abs(np.diff(h)).max()))
            prev = int(h[-1])
            last = col[-1].copy()
        return {"H_last_cell": int(row[-1, 0]), "last_column_tail_H": int(last[0]),
                "last_row_sha256": hs.hexdigest(), "last_column_sha256": hc.hexdigest(),
                "max_step_last_row": mx_row, "max_step_last_column": mx}

    r = run(al, part, before_end=borders, recurrence_type=NEEDLEMAN_WUNSCH, first_row_init_type=INIT_WITH_GAPS,
            first_column_init_type=INIT_WITH_GAPS, want_last_row=True, want_last_column=True, track_best=False)
    al.close()
    out.update(r)
    oracle = g.load_oracle()
    t0 = time.time()
    top = oracle.stage1(s0[:k_rows], s1, recurrence=NEEDLEMAN_WUNSCH, first_row_type=INIT_WITH_GAPS,
                        first_col_type=INIT_WITH_GAPS, want_last_col=True, best_mode=oracle.BEST_LAST_CELL)
    left = oracle.stage1(s0, s1[:k_cols], recurrence=NEEDLEMAN_WUNSCH, first_row_type=INIT_WITH_GAPS,
                         first_col_type=INIT_WITH_GAPS, want_last_row=True, best_mode=oracle.BEST_LAST_CELL)
    out["oracle_s"] = time.time() - t0
    oc, orow = np.asarray(top["last_col"]), np.asarray(left["last_row"])
    oc = oc[-k_rows:] if len(oc) > k_rows else oc          # a leading corner cell, if the oracle returns one
    orow = orow[-k_cols:] if len(orow) > k_cols else orow
    out["check"] = {
        "last_column_head_rows": k_rows, "last_column_head_equal": bool((keep["col_head"] == oc).all()),
        "last_row_head_cols": k_cols, "last_row_head_equal": bool((keep["row_head"] == orow).all()),
        "lipschitz_bound": 9,
        "lipschitz_ok": bool(r["rows"]["max_step_last_row"] <= 9 and r["rows"]["max_step_last_column"] <= 9),
        "corner_consistent": bool(r["rows"]["H_last_cell"] == r["rows"]["last_column_tail_H"]),
    }
    out["check"]["ok"] = all(v for k, v in out["check"].items() if isinstance(v, bool))
    assert out["check"]["ok"], out["check"]


def case_c5band_seeded(out, m=249000000, n=28500000, n_total=228000000, k_rows=100):
    """band 0 of C5 as it runs in the pruning chain since the chain has a seed: the diagonal seed pass over the WHOLE
    249 M x 228 M matrix first (mi355sw_seed_bound: what band 0 does before the chain starts, bands.chain_seed_bound), then
    the band with block pruning against that lower bound of H[249 M][228 M] and the extents of the whole matrix.  One GPU's
    share of BASELINE config 5 as worded, seed included.  Pinned on: the first k_rows cells of the last column against the
    oracle's sweep of those rows (equal or skipped); the seed is a lower bound that some global alignment reaches (nothing can check H[m][n]
    of 5.7e16 cells here; at a quarter of the size it IS H[m][n]: profiles/r04_nw_c5_quarter_62Mx57M_final.json)."""
    from masa_cudalign_amd.engine import NEEDLEMAN_WUNSCH, INIT_WITH_GAPS
    t0 = time.time()
    s0, s1 = pkg.seqgen.related_pair(m, n_total, cfg=15)
    out["generate_s"] = time.time() - t0
    al = pkg.MI355Aligner(device=0)
    al.setSequences(s0, s1)
    t0 = time.time()
    bound = al.seedBound(pkg.Partition(0, 0, m, n_total), NEEDLEMAN_WUNSCH)
    out["seed_s"] = time.time() - t0
    out["seed_bound"] = bound
    print(json.dumps({"seed_s": out["seed_s"], "seed_bound": bound}), flush=True)
    part = pkg.Partition(0, 0, m, n)
    keep = {}

    def borders(al):
        col = al.streamReadColumn(0, k_rows)
        keep["col_head"] = col.copy()
        return None

    r = run(al, part, before_end=borders, recurrence_type=NEEDLEMAN_WUNSCH, first_row_init_type=INIT_WITH_GAPS,
            first_column_init_type=INIT_WITH_GAPS, want_last_column=True, track_best=False, prune_blocks=True, prune_rows=m, prune_cols=n_total,
            initial_bound=bound)
    al.close()
    out.update(r)
    out["band_seconds_with_seed"] = out["seed_s"] + r["kernel_ms"] / 1e3
    out["band_gcups_mn_with_seed"] = float(m) * n / out["band_seconds_with_seed"] / 1e9
    oracle = g.load_oracle()
    top = oracle.stage1(s0[:k_rows], s1[:n], recurrence=NEEDLEMAN_WUNSCH, first_row_type=INIT_WITH_GAPS,
                        first_col_type=INIT_WITH_GAPS, want_last_col=True, best_mode=oracle.BEST_LAST_CELL)
    oc = np.asarray(top["last_col"])
    oc = oc[-k_rows:] if len(oc) > k_rows else oc
    # (with a bound this tight the top of the last column -- row 0..k_rows at column n, far off the diagonal -- is itself skipped:
    #  lower bounds of the oracle's cells, equal where they were computed)
    kh, oh = keep["col_head"][:, 0].astype(np.int64), oc[:, 0].astype(np.int64)
    out["check"] = {"last_column_head_rows": k_rows, "last_column_head_lower_bound": bool((kh <= oh).all()),
                    "last_column_head_equal_or_skipped": bool(((kh == oh) | (kh <= -900000000)).all()),
                    "a_bound_was_found": bound is not None, "most_of_the_band_skipped": r["pruned_fraction"] > 0.5}
    out["check"]["ok"] = all(v for v in out["check"].values() if isinstance(v, bool))
    assert out["check"]["ok"], out["check"]


def case_c5band_pruned(out, m=249000000, n=28500000, n_total=228000000, k_rows=400):
    """band 0 of C5 once more, as it runs in the pruning chain (round 4): block pruning ON against the lower bound of
    H[249 M][228 M] -- the bound looks at all 228 M columns (prune_cols), the band's own cells supply it.  Alone, the band
    only knows what its own stretch of the diagonal says about the last cell, so it skips the rows below ~203 M (in the
    chain the bands to the right raise the bound and it skips more: profiles/r04_chain_nw_c5_quarter_8bands.json).  The
    last column is kept as it would be streamed to band 1; pinned on: its first k_rows cells = the oracle's sweep of those
    rows, -INF from the first skipped strip on, sha256 recorded."""
    from masa_cudalign_amd.engine import NEEDLEMAN_WUNSCH, INIT_WITH_GAPS
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=15)
    part = pkg.Partition(0, 0, m, n)
    al = pkg.MI355Aligner(device=0)
    al.setSequences(s0, s1)
    keep = {}

    def borders(al):
        hc, step, first_void, voids = hashlib.sha256(), 1 << 24, None, 0
        for r0 in range(0, m, step):
            col = al.streamReadColumn(r0, min(step, m - r0))
            if r0 == 0:
                keep["col_head"] = col[:k_rows].copy()
            hc.update(np.ascontiguousarray(col).tobytes())
            v = np.nonzero(col[:, 0] <= -900000000)[0]
            voids += len(v)
            if len(v) and first_void is None:
                first_void = r0 + int(v[0])
        return {"last_column_sha256": hc.hexdigest(), "first_skipped_row_of_the_last_column": first_void, "skipped_cells_of_the_last_column": voids}

    r = run(al, part, before_end=borders, recurrence_type=NEEDLEMAN_WUNSCH, first_row_init_type=INIT_WITH_GAPS,
            first_column_init_type=INIT_WITH_GAPS, want_last_column=True, track_best=False, prune_blocks=True, prune_rows=m, prune_cols=n_total)
    al.close()
    out.update(r)
    oracle = g.load_oracle()
    top = oracle.stage1(s0[:k_rows], s1, recurrence=NEEDLEMAN_WUNSCH, first_row_type=INIT_WITH_GAPS,
                        first_col_type=INIT_WITH_GAPS, want_last_col=True, best_mode=oracle.BEST_LAST_CELL)
    oc = np.asarray(top["last_col"])
    oc = oc[-k_rows:] if len(oc) > k_rows else oc
    out["check"] = {"last_column_head_rows": k_rows, "last_column_head_equal": bool((keep["col_head"] == oc).all()),
                    "something_skipped": r["pruned_fraction"] > 0.1}
    out["check"]["ok"] = all(v for v in out["check"].values() if isinstance(v, bool))
    assert out["check"]["ok"], out["check"]


def case_c4chain(out, m=59000000, w=8000000):
    """C4's height (59 M rows) as a chain of two column bands of w columns through the multi-GPU band driver
    (masa-cudalign_amd/bands.py, the code bench.py --gpus N runs), one band after the other on this one GPU with the
    boundary column handed over in 32 k-row segments; the chain must report what ONE partition of 2w columns reports"""
    import collections
    from masa_cudalign_amd.bands import BandRunner, canonical_best

    class Relay:                     # stands in for torch.distributed: rank 0's sends are rank 1's receives
        def __init__(self):
            self.q = collections.deque()
        def send(self, t, dst):
            self.q.append(t.clone())
        def recv(self, t, src):
            while not self.q:
                time.sleep(0.001)
            t.copy_(self.q.popleft())

    s0, s1 = pkg.seqgen.unrelated_pair(m, 2 * w, cfg=4)
    al = pkg.MI355Aligner(device=0)
    al.setSequences(s0, s1)
    relay = Relay()
    res = []
    for rank in (0, 1):
        t0 = time.time()
        b = BandRunner(al, dist=relay, rank=rank, world=2, device=None, segment_rows=1 << 15).run(m, rank * w, (rank + 1) * w)
        st = al.getStatistics()
        res.append({"band": rank, "best": list(b), "wall_s": time.time() - t0, "kernel_ms": st["kernel_ms"],
                    "gcups": m * w / st["kernel_ms"] / 1e6, "strip_rows": st["strip_rows"]})
        print(json.dumps(res[-1]), flush=True)
    t0 = time.time()
    whole = BandRunner(al, dist=None, rank=0, world=1).run(m, 0, 2 * w)
    st = al.getStatistics()
    out["bands"] = res
    out["whole"] = {"best": list(whole), "wall_s": time.time() - t0, "kernel_ms": st["kernel_ms"], "gcups": m * 2 * w / st["kernel_ms"] / 1e6}
    out["chain_best"] = list(canonical_best([tuple(r["best"]) for r in res]))
    out["agree"] = out["chain_best"] == out["whole"]["best"]
    al.close()
    out["check"] = window_check(s0, s1, out["chain_best"])
    assert out["agree"] and out["check"]["ok"], out


def case_c2x(out, m=3000000, n=3000000):
    s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=2)
    al = pkg.MI355Aligner(device=0)
    al.setSequences(s0, s1)
    part = pkg.Partition(0, 0, m, n)
    plain = run(al, part)
    full = run(al, part, special_row_interval=m // 16, want_last_row=True, want_last_column=True)
    out["plain"], out["full"] = plain, full
    out["same_best"] = plain["best"] == full["best"]
    al.close()
    assert out["same_best"]


if __name__ == "__main__":
    case = sys.argv[1]
    out = {"case": case}
    if case == "c3":
        case_c3(out)
    elif case == "c3pruned":
        case_c3_pruned_only(out, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_scale_c3_48Mx46M.json"))
    elif case == "c3small":
        case_c3(out, 6000000, 5000000)
    elif case == "tall":
        case_unrelated(228000000, 1000000, 11, out)
    elif case == "wide":
        case_unrelated(1000000, 228000000, 12, out)
    elif case == "tallband":          # 1/14 of the 228 M x 228 M north-star matrix: its full height, one band of its width
        case_unrelated(228000000, 16000000, 13, out)
    elif case == "tallsmall":
        case_unrelated(40000000, 100000, 11, out)
    elif case == "nwtall":
        case_nw_tall(out)
    elif case == "nwtallsmall":
        case_nw_tall(out, 40000000, 50000)
    elif case == "c5band_seeded":
        case_c5band_seeded(out)
    elif case == "c5band_seeded_small":
        case_c5band_seeded(out, 12000000, 1400000, 11000000, 100)
    elif case == "c5band_pruned":
        case_c5band_pruned(out)
    elif case == "c5band_pruned_small":
        case_c5band_pruned(out, 6000000, 400000, 5500000, 300)
    elif case == "c5band":
        case_c5band(out)
    elif case == "c5bandsmall":
        case_c5band(out, 6000000, 400000, 300, 40)
    elif case == "c4chain":
        case_c4chain(out)
    elif case == "c4chainsmall":
        case_c4chain(out, 3000000, 400000)
    elif case == "c2x":
        case_c2x(out)
    else:
        sys.exit("unknown case")
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)

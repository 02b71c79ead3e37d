"""Block pruning of global alignments on the GPU: a related pair swept with and without pruning.

    python tools/nw_prune_probe.py M N [oracle] [sw]

Checks (size-independent): H[m][n] of the pruned run = the unpruned run's; every cell of the last row, the last column and
the special rows is a lower bound of the unpruned one and equal where it is not a skipped or skipped-derived cell near the
goal; with `oracle`, the unpruned run against the oracle as well.  `sw`: the same pair as a local alignment (the pruning
window no longer has to sit at the floor: fraction skipped, best cell).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def run(pkg, al, s0, s1, edge, prune, interval, keep=True, rows_per_lane=None):
    m, n = len(s0), len(s1)
    part = pkg.Partition(0, 0, m, n)
    mg = pkg.Stage1Manager(part, alignment_start=edge, alignment_end=edge, special_row_interval=interval,
                           keep_last_row=keep, keep_last_column=keep, block_pruning=prune)
    t0 = time.time()
    al.alignPartition(part, mg)
    dt = time.time() - t0
    st = al.getStatistics()
    return mg, st, dt


def main():
    m, n = int(sys.argv[1]), int(sys.argv[2])
    want_oracle = "oracle" in sys.argv[3:]
    sw_too = "sw" in sys.argv[3:]
    pkg = graft.load_package()
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=41)
    al = pkg.MI355Aligner(device=0)
    al.setSequences(s0, s1)
    out = {"m": m, "n": n}
    interval = max(8192, m // 8 // 8192 * 8192)      # a multiple of 8192: the same rows whatever strip height each run picks
    res = {}
    for prune in (False, True):
        mg, st, dt = run(pkg, al, s0, s1, pkg.AT_SEQUENCE_1_AND_2, prune, interval)
        res[prune] = mg
        out["nw_pruned" if prune else "nw_plain"] = {
            "best": list(mg.getBestScore()), "kernel_ms": st["kernel_ms"], "wall_s": dt, "pruned_fraction": st["pruned_cells"] / float(m) / n,
            "gcups_mn": m * n / st["kernel_ms"] / 1e6, "strip_rows": st["strip_rows"], "kernel": st["profile_kernel"],
            "launches": st["kernel_launches"]}
    a, b = res[False], res[True]
    chk = {"best_equal": list(a.getBestScore()) == list(b.getBestScore())}
    for name, get in (("last_row", lambda g: g.lastRow()), ("last_col", lambda g: g.lastColumn())):
        x, y = get(a), get(b)
        chk[name + "_lower_bound"] = bool((y <= x).all())
        chk[name + "_equal_fraction"] = float((y[:, 0] == x[:, 0]).mean())
        chk[name + "_tail_equal"] = bool((y[-64:] == x[-64:]).all())
    rows = sorted(a.special_rows)
    chk["special_rows"] = len(rows)
    chk["special_rows_lower_bound"] = bool(all((b.specialRow(i) <= a.specialRow(i)).all() for i in rows))
    chk["special_rows_equal_fraction"] = [float((b.specialRow(i)[:, 0] == a.specialRow(i)[:, 0]).mean()) for i in rows]
    # where the optimal path crosses a special row the cell is exact: the row maximum of (H + best continuation) is
    # attained there, so the unpruned row's cells that could still reach the final score must be equal
    fin = a.getBestScore()[2]
    ok = True
    for i in rows:
        x, y = a.specialRow(i)[1:, 0].astype(np.int64), b.specialRow(i)[1:, 0].astype(np.int64)
        j = np.arange(1, n + 1, dtype=np.int64)
        di, dj = m - i, n - j
        reach = x + np.minimum(di, dj) - 2 * np.abs(dj - di)
        must = reach >= fin
        ok = ok and bool((x[must] == y[must]).all())
    chk["cells_that_can_reach_the_goal_are_exact"] = ok
    out["check"] = chk
    if want_oracle:
        oracle = graft.load_oracle()
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        from helpers import oracle_kwargs
        st = al.getStatistics()
        kw = oracle_kwargs(oracle, dict(start=4, end=4, pruning=False, disk=-1, block=(out["nw_plain"]["strip_rows"], 1 << 20)), m, n)
        kw.update(want_last_row=True, want_last_col=True)
        ref = oracle.stage1(s0, s1, **kw)
        out["oracle"] = {"best_equal": list(ref["best"]) == list(a.getBestScore()),
                         "last_row_equal": bool(np.array_equal(a.lastRow(), ref["last_row"])),
                         "last_col_equal": bool(np.array_equal(a.lastColumn(), ref["last_col"]))}
    if sw_too:
        res = {}
        for prune in (False, True):
            mg, st, dt = run(pkg, al, s0, s1, pkg.AT_ANYWHERE, prune, interval, keep=True)
            res[prune] = mg
            out["sw_pruned" if prune else "sw_plain"] = {
                "best": list(mg.getBestScore()), "kernel_ms": st["kernel_ms"], "pruned_fraction": st["pruned_cells"] / float(m) / n,
                "gcups_mn": m * n / st["kernel_ms"] / 1e6, "strip_rows": st["strip_rows"], "kernel": st["profile_kernel"],
                "launches": st["kernel_launches"]}
        a, b = res[False], res[True]
        rows = sorted(a.special_rows)
        out["check"]["sw_best_equal"] = out["sw_plain"]["best"] == out["sw_pruned"]["best"]
        out["check"]["sw_rows_lower_bound"] = bool(all((b.specialRow(i) <= a.specialRow(i)).all() for i in rows))
        out["check"]["sw_rows_max_equal"] = bool(all(b.specialRow(i)[:, 0].max() == a.specialRow(i)[:, 0].max() for i in rows
                                                     if i <= out["sw_plain"]["best"][0]))
        out["check"]["sw_last_col_lower_bound"] = bool((b.lastColumn() <= a.lastColumn()).all())
        out["check"]["sw_last_row_lower_bound"] = bool((b.lastRow() <= a.lastRow()).all())
    al.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

"""The seed (round 5: anchored; MI355SW_STAIRCASE_SEED=1: round 4's staircase) over pairs of different make: python tools/seed_sweep.py [out.json]
For each pair (seqgen.related_pair with other mutation rates, inverted segments of other sizes, none at all, unequal lengths)
one pruning run that starts from the seed's bound and one that starts from nothing: the answers must be equal (best cell of
the local alignment / H[m][n] of the global one), the seed must be a score that exists (<= the answer), and the record says
how much of the matrix each run skipped."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402
pkg = g.load_package()

CASES = [
    (9000000, 8500000, 5, {}),
    (9000000, 8800000, 7, dict(inversion=0.0)),
    (9000000, 8500000, 11, dict(inversion=0.15)),
    (10000000, 8500000, 13, dict(p_indel=0.01, indel_mean=8.0)),
    (8500000, 9500000, 17, dict(p_sub=0.08, inversion=0.02)),
    # round 5 (the anchored seed: anchors from stripes, segments in bands of +-64 Ki columns around straight lines)
    (9000000, 8800000, 19, dict(special="big_indels")),       # a 40 000-base insertion at 40 % and a 30 000-base deletion at 70 % of seq1
    (9000000, 8800000, 23, dict(special="duplication")),      # a 100 000-base segment of seq0 copied into seq1 a second time, at its middle
    (9000000, 8800000, 29, dict(special="late_start")),       # seq1 starts with 2.5 M unrelated bases: nothing aligns at the left edge
    (26000000, 24000000, 31, {}),                             # enough columns for five anchors
]


def make_pair(m, n, cfg, kw):
    import numpy as np
    sg = pkg.seqgen
    special = kw.get("special")
    if special is None:
        return sg.related_pair(m, n, cfg=cfg, **kw)
    s0 = sg.random_dna(sg.SEED0 + cfg, m)
    base = sg.mutate_dna(s0, sg.SEED1 + cfg, inversion=0.0)
    if special == "big_indels":
        a, b = int(0.4 * n), int(0.7 * n)
        s1 = np.concatenate([base[:a], sg.random_dna(77 + cfg, 40000), base[a:b], base[b + 30000:]])
    elif special == "duplication":
        a = n // 2
        s1 = np.concatenate([base[:a], base[1000000:1100000], base[a:]])
    else:
        s1 = np.concatenate([sg.random_dna(78 + cfg, 2500000), base])
    if len(s1) < n:
        s1 = np.concatenate([s1, sg.random_dna(79 + cfg, n - len(s1))])
    return s0, np.ascontiguousarray(s1[:n])


def sweep(al, part, n, kind, bound):
    if kind == "sw":
        al.streamBegin(part, prune_blocks=True, initial_bound=bound)
    else:
        al.streamBegin(part, recurrence_type=pkg.NEEDLEMAN_WUNSCH, track_best=False, first_row_init_type=pkg.INIT_WITH_GAPS,
                       first_column_init_type=pkg.INIT_WITH_GAPS, want_last_row=True, prune_blocks=True, initial_bound=bound)
    while True:
        _rows, fin = al.streamPoll()
        if fin:
            break
        time.sleep(0.005)
    h = int(al.streamReadLastRow(col=n - 1, length=1)[0, 0]) if kind != "sw" else None
    best, _ = al.streamEnd()
    st = al.getStatistics()
    return (h if kind != "sw" else [int(x) for x in best]), st["pruned_cells"] / float(st["cells"]), st["kernel_ms"], st["restarts"]


out = {"cases": []}
os.environ["MI355SW_NO_DIAGONAL_SEED"] = "1"           # the sweeps below start from the bound they are given, or from nothing
only = [int(x) for x in os.environ.get("SEED_SWEEP_CASES", "").split(",") if x]       # e.g. SEED_SWEEP_CASES=2,6: those cases only
for idx, (m, n, cfg, kw) in enumerate(CASES):
    if only and idx not in only:
        continue
    s0, s1 = make_pair(m, n, cfg, kw)
    al = pkg.MI355Aligner(device=0)
    al.setSequences(s0, s1)
    part = pkg.Partition(0, 0, m, n)
    for kind in ("sw", "nw"):
        del os.environ["MI355SW_NO_DIAGONAL_SEED"]
        t0 = time.time()
        bound = al.seedBound(part, pkg.SMITH_WATERMAN if kind == "sw" else pkg.NEEDLEMAN_WUNSCH)
        seed_s = time.time() - t0
        os.environ["MI355SW_NO_DIAGONAL_SEED"] = "1"
        a = sweep(al, part, n, kind, bound)
        b = sweep(al, part, n, kind, None)
        answer = a[0][2] if kind == "sw" else a[0]
        rec = {"m": m, "n": n, "cfg": cfg, "mutation": kw, "kind": kind, "seed_bound": bound, "seed_s": seed_s, "answer_seeded": a[0], "answer_unseeded": b[0],
               "skipped_seeded": a[1], "skipped_unseeded": b[1], "kernel_ms_seeded": a[2], "kernel_ms_unseeded": b[2], "restarts": a[3] + b[3],
               "ok": a[0] == b[0] and (bound is None or bound <= answer)}
        out["cases"].append(rec)
        print(json.dumps(rec), flush=True)
    al.close()
out["ok"] = all(c["ok"] for c in out["cases"])
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
assert out["ok"]

"""GPU (-m gpu): the packed kernel's HOT chunk loop (csrc/sw_kernel_pk16.inc, "the chunk loop") and every way out of it.
A strip enters the hot loop after its 257th chunk (columns >= 16 448), so these cases are WIDE: the alignment's ridge,
runs of N, a re-basing window and the end of the row all lie beyond that column, where a chunk of the hot instance has
to hand over to the general one mid-way (exact replay, new maximum / window shift) or must not be taken by it at all
(codes outside the table, the last chunks).  Bit-exact against the oracle, and against the int32 kernels (which have
no such loop) at sizes the oracle cannot sweep."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EDGE = {0: "AT_ANYWHERE", 1: "AT_SEQUENCE_1", 2: "AT_SEQUENCE_2", 3: "AT_SEQUENCE_1_OR_2", 4: "AT_SEQUENCE_1_AND_2"}
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _wide_pair(kind, m, n, seed):
    """seq1: n random letters; seq0: m letters related to seq1[j0 : j0 + m] (4 % substitutions, a few indels) with j0 far
    beyond column 16 448, or unrelated"""
    rng = np.random.default_rng(seed)
    s1 = ACGT[rng.integers(0, 4, size=n)]
    if kind == "unrelated":
        return ACGT[rng.integers(0, 4, size=m)], s1
    j0 = 20000 + int(rng.integers(0, 3000))
    src = s1[j0:j0 + m + 200].copy()
    out = []
    i = 0
    while len(out) < m and i < len(src):
        r = rng.random()
        if r < 0.04:
            out.append(ACGT[rng.integers(0, 4)])
            i += 1
        elif r < 0.045:
            i += int(rng.integers(1, 6))                       # deletion
        elif r < 0.05:
            out.extend(ACGT[rng.integers(0, 4, size=int(rng.integers(1, 6)))])   # insertion
        else:
            out.append(src[i])
            i += 1
    s0 = np.array(out[:m], dtype=np.uint8)
    if len(s0) < m:
        s0 = np.concatenate([s0, ACGT[rng.integers(0, 4, size=m - len(s0))]])
    if kind == "with_n":                                        # codes outside the 4-letter table, inside the hot zone
        for p in (17000, 21011, 21012, 21013, 30000, n - 300):
            s1[p:p + int(rng.integers(1, 90))] = ord("N")
        s0[m // 2:m // 2 + 3] = ord("N")
    return np.ascontiguousarray(s0), np.ascontiguousarray(s1)


@pytest.mark.parametrize("R", [0, 4, 12, 24, 32])
@pytest.mark.parametrize("kind,start,end", [("related", 0, 0), ("unrelated", 0, 0), ("with_n", 0, 0), ("related", 4, 4),
                                            ("related", 1, 3), ("related", 2, 2)])
def test_wide_partitions_against_the_oracle(pkg, oracle, kind, start, end, R):
    from helpers import oracle_kwargs
    m, n = (2600, 40000) if R in (0, 32) else (1700, 36000)
    s0, s1 = _wide_pair(kind, m, n, seed=1000 + 7 * R + start)
    al = pkg.MI355Aligner(device=0, rows_per_lane=R)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        mg = pkg.Stage1Manager(part, alignment_start=getattr(pkg, EDGE[start]), alignment_end=getattr(pkg, EDGE[end]),
                               keep_last_row=True, keep_last_column=True)
        al.alignPartition(part, mg)
        st = al.getStatistics()
        assert st["profile_kernel"] == 2                       # the packed family ran, no fallback
        kw = oracle_kwargs(oracle, dict(start=start, end=end, pruning=False, disk=-1, block=(st["strip_rows"], 1 << 20)), m, n)
        kw.update(want_last_row=True, want_last_col=True)
        ref = oracle.stage1(s0, s1, **kw)
        assert tuple(mg.getBestScore()) == tuple(ref["best"])
        assert np.array_equal(mg.lastRow(), ref["last_row"])
        assert np.array_equal(mg.lastColumn(), ref["last_col"])
        if kind != "unrelated" and start == 0:
            assert ref["best"][1] > 16448 + 64                  # the ridge does lie where the hot loop runs
    finally:
        al.close()


@pytest.mark.parametrize("prune", [False, True])
def test_hot_loop_and_int32_kernels_agree_on_a_large_related_pair(pkg, prune):
    """300 000 x 200 000 related pair (score ~ 170 000: several window shifts, a ridge of 200 000 columns in exact mode;
    with pruning about a third of the slabs skipped, most of them in runs that are tested and written out eight at a
    time): the packed kernels with their hot loop and the int32 kernels give the same best cell; unpruned also the same
    special rows, last row and last column, cell for cell; pruned, every row is a lower bound of the unpruned one, with
    the same maximum where the alignment crosses it."""
    m, n = 300000, 200000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=71)
    part = pkg.Partition(0, 0, m, n)
    res = {}
    for flags in (0, 2):
        al = pkg.MI355Aligner(device=0, flags=flags, rows_per_lane=16)
        try:
            al.setSequences(s0, s1)
            mg = pkg.Stage1Manager(part, keep_last_row=True, keep_last_column=True, block_pruning=prune and flags == 0,
                                   special_row_interval=65536)
            al.alignPartition(part, mg)
            st = al.getStatistics()
            assert st["profile_kernel"] == (2 if flags == 0 else 1)
            rows = {i: mg.specialRow(i) for i in sorted(mg.special_rows)}
            res[flags] = (tuple(mg.getBestScore()), mg.lastRow(), mg.lastColumn(), st["pruned_cells"], rows)
        finally:
            al.close()
    assert res[0][0] == res[2][0] and res[0][0][2] > 100000
    assert sorted(res[0][4]) == sorted(res[2][4]) and len(res[0][4]) >= 4
    if prune:
        assert res[0][3] > 0.2 * m * n
        assert np.all(res[0][1] <= res[2][1])                   # a pruned row is a lower bound of the unpruned one, H and F
        for i in res[0][4]:
            a, b = res[0][4][i][:, 0], res[2][4][i][:, 0]
            assert np.all(res[0][4][i] <= res[2][4][i]), i      # (a slab written out by nobody would show as stale cells here)
            assert np.all(a <= b), i
            if i <= res[2][0][0]:                               # the alignment's ridge crosses this row: its cell survives
                assert a.max() == b.max(), i
    else:
        assert np.array_equal(res[0][1], res[2][1])
        assert np.array_equal(res[0][2], res[2][2])
        for i in res[0][4]:
            assert np.array_equal(res[0][4][i], res[2][4][i]), i


def test_two_phase_run_with_pruning_locates_the_same_cell(pkg, monkeypatch):
    """value-only main pass (the kernels very tall partitions get) with pruning -- runs of pruned slabs also write the
    checkpoint rows the exact pass restarts from -- then the exact pass: the same best cell as the int32 single pass"""
    m, n = 300000, 200000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=71)
    part = pkg.Partition(0, 0, m, n)
    res = {}
    for flags in (0, 2):
        if flags == 0:
            monkeypatch.setenv("MI355SW_TWO_PHASE", "1")
        else:
            monkeypatch.delenv("MI355SW_TWO_PHASE", raising=False)
        al = pkg.MI355Aligner(device=0, flags=flags, rows_per_lane=8)
        try:
            al.setSequences(s0, s1)
            mg = pkg.Stage1Manager(part, block_pruning=flags == 0)
            al.alignPartition(part, mg)
            st = al.getStatistics()
            res[flags] = (tuple(mg.getBestScore()), st)
        finally:
            al.close()
    assert res[0][1]["kernel_launches"] >= 2 and res[0][1]["pruned_cells"] > 0.2 * m * n
    assert res[0][0] == res[2][0]


def test_score_only_pass_with_mixed_heights_agrees_with_int32(pkg):
    """the C2 kernel itself (mixed strip heights in one launch, score only, hot loop in every strip) on a shape that
    engages it -- 1.9 rounds of 1536-row strips, wide enough for the cost model to want them (at 70 000 columns, the
    shape this test had in round 3, it picks 512-row strips and the mixed form never ran) -- against the int32 kernels:
    same best cell"""
    m, n = 3000000, 600000
    # (an UNRELATED pair: since round 4 the engine looks at what its seed pass found before it plans, and a related pair
    #  gets the strip heights of a pruning run -- one height, no mixed form)
    s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=72)
    part = pkg.Partition(0, 0, m, n)
    res = {}
    for flags in (0, 2):
        al = pkg.MI355Aligner(device=0, flags=flags)
        try:
            al.setSequences(s0, s1)
            mg = pkg.Stage1Manager(part)
            al.alignPartition(part, mg)
            res[flags] = (tuple(mg.getBestScore()), al.getStatistics())
        finally:
            al.close()
    assert res[0][0] == res[2][0]
    assert res[0][1]["profile_kernel"] == 2
    assert res[0][1]["kernel"] == "sw_strip_kernel_pk16_mixed<12,11,true,true>" and res[0][1]["strip_rows_second"] == 1408    # it did engage


def _fuzz_wide(k):
    rng = np.random.default_rng(7000 + k)
    m = int(rng.integers(20000, 220000))
    n = int(rng.integers(40000, 320000))
    sub = float(rng.choice([0.01, 0.04, 0.10, 0.20]))
    src = ACGT[rng.integers(0, 4, size=max(m, n) + 4096)]
    s1 = src[:n].copy()
    # seq0: a mutated copy of a window of the common source (ridge somewhere in the matrix), sometimes two windows glued
    off = int(rng.integers(0, max(1, n - m // 2)))
    base = np.concatenate([src[off:], src[:off]])[:m].copy()
    mut = rng.random(m) < sub
    base[mut] = ACGT[rng.integers(0, 4, size=int(mut.sum()))]
    if rng.random() < 0.4:                                       # a block of unrelated sequence in the middle
        a = int(rng.integers(0, m - 1000))
        ln = min(int(rng.integers(500, 20000)), m - a)
        base[a:a + ln] = ACGT[rng.integers(0, 4, size=ln)]
    if rng.random() < 0.3:
        p = int(rng.integers(17000, n - 200))
        s1[p:p + int(rng.integers(1, 150))] = ord("N")
    R = int(rng.choice([0, 4, 8, 12, 16, 24, 32]))
    mode = ["sw", "sw_prune", "sw_prune", "nw", "semi"][int(rng.integers(0, 5))]
    interval = int(rng.choice([0, 0, 16384, 65536]))
    return m, n, np.ascontiguousarray(base), np.ascontiguousarray(s1), R, mode, interval


@pytest.mark.parametrize("k", range(96))
def test_randomised_wide_runs_against_the_int32_kernels(pkg, k):
    """96 seeded wide configurations (20 000-220 000 rows x 40 000-320 000 columns, related pairs with 1-20 % substitutions,
    runs of N, every strip height; local with and without pruning, global, semi-global; with and without special rows):
    the packed kernels (hot chunk loop, runs of pruned slabs) against the int32 kernels, which have neither -- same best
    cell; without pruning the same last row, last column and special rows, cell for cell; with pruning lower bounds."""
    m, n, s0, s1, R, mode, interval = _fuzz_wide(k)
    start, end = {"sw": (0, 0), "sw_prune": (0, 0), "nw": (4, 4), "semi": (1, 3)}[mode]
    part = pkg.Partition(0, 0, m, n)
    res = {}
    for flags in (0, 2):
        # the int32 family has strips of 256 / 512 / 1024 rows: special rows need the same rows on both sides
        rpl = R if (interval == 0 or R in (4, 8, 16)) else 16
        al = pkg.MI355Aligner(device=0, flags=flags, rows_per_lane=rpl if flags == 0 or rpl in (4, 8, 16) else 0)
        try:
            al.setSequences(s0, s1)
            mg = pkg.Stage1Manager(part, alignment_start=getattr(pkg, EDGE[start]), alignment_end=getattr(pkg, EDGE[end]),
                                   keep_last_row=True, keep_last_column=True, special_row_interval=interval,
                                   block_pruning=(mode == "sw_prune" and flags == 0))
            al.alignPartition(part, mg)
            st = al.getStatistics()
            rows = {i: mg.specialRow(i) for i in sorted(mg.special_rows) if i < m}
            res[flags] = (tuple(mg.getBestScore()), mg.lastRow(), mg.lastColumn(), st, rows)
        finally:
            al.close()
    assert res[0][3]["profile_kernel"] == 2 and res[2][3]["profile_kernel"] == 1
    assert res[0][0] == res[2][0], (mode, m, n, R)
    pruned = mode == "sw_prune" and res[0][3]["pruned_cells"] > 0
    if interval:
        assert sorted(res[0][4]) == sorted(res[2][4])
    if pruned:
        assert np.all(res[0][1] <= res[2][1]) and np.all(res[0][2] <= res[2][2])
        for i in res[0][4]:
            assert np.all(res[0][4][i] <= res[2][4][i]), i
    else:
        assert np.array_equal(res[0][1], res[2][1]) and np.array_equal(res[0][2], res[2][2])
        for i in res[0][4]:
            assert np.array_equal(res[0][4][i], res[2][4][i]), i

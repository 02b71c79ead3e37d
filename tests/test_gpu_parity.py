"""GPU (-m gpu): the HIP strip-wavefront engine, called through the C ABI, against the oracle on the
same seeded inputs and against the fixtures generated from the reference's own CPU path.
Integer work: every comparison is bit-exact."""
import numpy as np
import pytest

from helpers import load_golden, make_pair, digest, parse_args, flush_interval

pytestmark = pytest.mark.gpu
G = load_golden()
EDGE = {0: "AT_ANYWHERE", 1: "AT_SEQUENCE_1", 2: "AT_SEQUENCE_2", 3: "AT_SEQUENCE_1_OR_2", 4: "AT_SEQUENCE_1_AND_2"}


def run_stage1(pkg, al, s0, s1, start=0, end=0, interval=0, keep_last_row=False, keep_last_col=False):
    al.setSequences(s0, s1)
    part = pkg.Partition(0, 0, len(s0), len(s1))
    mg = pkg.Stage1Manager(part, alignment_start=getattr(pkg, EDGE[start]), alignment_end=getattr(pkg, EDGE[end]),
                           special_row_interval=interval, keep_last_row=keep_last_row, keep_last_column=keep_last_col)
    al.alignPartition(part, mg)
    al.unsetSequences()
    return mg


@pytest.mark.parametrize("case", G["cases"], ids=[c["name"] for c in G["cases"]])
def test_golden_fixture(case, pkg, aligner):
    """best score + canonical position (and special rows) against the reference-generated fixture."""
    s0, s1 = make_pair(pkg, case["seq"])
    p = parse_args(case["args"])
    sr = "special_rows" in case and not case["name"].startswith("full_pipeline")
    interval = flush_interval(len(s0), len(s1), p["disk"]) if sr else 0
    mg = run_stage1(pkg, aligner, s0, s1, p["start"], p["end"], interval=interval, keep_last_row=sr)
    assert list(mg.getBestScore()) == case["best"]
    if sr:
        got = sorted(mg.special_rows)
        want = sorted(int(k) for k in case["special_rows"])
        assert got == want
        for i in got:
            assert digest(mg.specialRow(i)) == case["special_rows"][str(i)], "special row %d" % i


@pytest.mark.parametrize("R", [4, 8, 12, 16, 24, 32])
@pytest.mark.parametrize("m,n", [(1, 1), (1, 300), (300, 1), (63, 64), (64, 63), (255, 129), (256, 128), (257, 127),
                                 (513, 65), (1025, 1023), (2048, 100), (5000, 4321)])
def test_all_borders_vs_oracle_sw(pkg, oracle, R, m, n):
    """ragged sizes (not multiples of 4/64/512, n < 64, m < strip): best, last row and last column."""
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=m * 7 + n)
    al = pkg.MI355Aligner(device=0, rows_per_lane=R)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        mg = pkg.Stage1Manager(part, keep_last_row=True, keep_last_column=True)
        al.alignPartition(part, mg)
        ref = oracle.stage1(s0, s1, block_h=64 * R, block_w=97, want_last_row=True, want_last_col=True)
        assert tuple(mg.getBestScore()) == tuple(ref["best"])
        assert np.array_equal(mg.lastRow(), ref["last_row"])
        assert np.array_equal(mg.lastColumn(), ref["last_col"])
    finally:
        al.close()


@pytest.mark.parametrize("start,end", [(4, 4), (1, 3), (2, 3), (3, 3), (1, 1), (2, 2)])
def test_nw_and_semiglobal_edges(pkg, oracle, aligner, start, end):
    from helpers import oracle_kwargs
    m, n = 3001, 2777
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=50 + start * 5 + end)
    mg = run_stage1(pkg, aligner, s0, s1, start, end, keep_last_row=True, keep_last_col=True)
    st = aligner.getStatistics()
    kw = oracle_kwargs(oracle, dict(start=start, end=end, pruning=False, disk=-1, block=(st["strip_rows"], 1 << 20)), m, n)
    kw.update(want_last_row=True, want_last_col=True)
    ref = oracle.stage1(s0, s1, **kw)
    assert np.array_equal(mg.lastRow(), ref["last_row"])
    assert np.array_equal(mg.lastColumn(), ref["last_col"])
    assert tuple(mg.getBestScore()) == tuple(ref["best"])
    assert st["profile_kernel"] == 2        # NW and the semi-global forms run on the packed kernel, no fallback


@pytest.mark.parametrize("rel", [False, True])
def test_packed_kernel_nw_window_follows_scores_down(pkg, oracle, aligner, rel):
    """global alignment (gap-initialised borders) of 30 k x 45 k: scores fall far below -32768 (unrelated)
    or swing from negative to positive (related), so the packed kernel's 16-bit window has to be re-centred
    downwards as well as upwards -- and stay bit-exact on the last row, the last column and H[m][n]."""
    from helpers import oracle_kwargs
    m, n = 30000, 45000
    s0, s1 = (pkg.seqgen.related_pair if rel else pkg.seqgen.unrelated_pair)(m, n, cfg=88)
    mg = run_stage1(pkg, aligner, s0, s1, 4, 4, keep_last_row=True, keep_last_col=True)
    st = aligner.getStatistics()
    assert st["profile_kernel"] == 2 and st["kernel_launches"] == 1
    kw = oracle_kwargs(oracle, dict(start=4, end=4, pruning=False, disk=-1, block=(st["strip_rows"], 1 << 20)), m, n)
    kw.update(want_last_row=True, want_last_col=True)
    ref = oracle.stage1(s0, s1, **kw)
    assert ref["last_row"][:, 0].min() < -40000
    assert np.array_equal(mg.lastRow(), ref["last_row"])
    assert np.array_equal(mg.lastColumn(), ref["last_col"])
    assert tuple(mg.getBestScore()) == tuple(ref["best"])


def test_generic_compare_kernel_and_foreign_bytes(pkg, oracle):
    """more than 7 shared byte values force the raw-compare kernels; N matches N, bytes present in one
    sequence only never match (X/CUDAligner.cu:276-289 raw byte inequality)."""
    rng = np.random.default_rng(5)
    alpha = np.frombuffer(b"ACGTNRYKMSWBDHV", dtype=np.uint8)
    s0 = alpha[rng.integers(0, len(alpha), 3000)]
    s1 = s0.copy()
    s1[rng.integers(0, 3000, 300)] = alpha[rng.integers(0, len(alpha), 300)]
    s1 = np.concatenate([s1[:1500], alpha[rng.integers(0, 4, 77)], s1[1500:]])
    for flags in (0, 1):
        al = pkg.MI355Aligner(device=0, flags=flags)
        try:
            al.setSequences(s0, s1)
            part = pkg.Partition(0, 0, len(s0), len(s1))
            mg = pkg.Stage1Manager(part, keep_last_row=True)
            al.alignPartition(part, mg)
            assert al.getStatistics()["profile_kernel"] == 0
            ref = oracle.stage1(s0, s1, want_last_row=True)
            assert tuple(mg.getBestScore()) == tuple(ref["best"])
            assert np.array_equal(mg.lastRow(), ref["last_row"])
        finally:
            al.close()
    # ACGT+N pair: packed 16-bit kernel, int32 profile kernel and int32 raw-compare kernel must agree
    s0, s1 = make_pair(pkg, dict(kind="with_n", m=3000, n=3100, cfg=9))
    res = []
    for flags in (0, 2, 1):
        al = pkg.MI355Aligner(device=0, flags=flags)
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, len(s0), len(s1))
        mg = pkg.Stage1Manager(part, keep_last_column=True)
        al.alignPartition(part, mg)
        res.append((al.getStatistics()["profile_kernel"], tuple(mg.getBestScore()), mg.lastColumn()))
        al.close()
    assert [r[0] for r in res] == [2, 1, 0]
    for r in res[1:]:
        assert res[0][1] == r[1] and np.array_equal(res[0][2], r[2])


def test_tie_break_canonical_position(pkg, oracle, aligner):
    """unrelated pairs have many cells tied at the best score: (min i, then min j) must win (M5)."""
    for cfg in range(3):
        s0, s1 = pkg.seqgen.unrelated_pair(9000, 8000, cfg=70 + cfg)
        mg = run_stage1(pkg, aligner, s0, s1)
        assert tuple(mg.getBestScore()) == tuple(oracle.stage1(s0, s1)["best"])


def test_process_block_seam(pkg, oracle, aligner):
    """AbstractBlockProcessor::processBlock (S3): random borders incl. -INF entries, SW and NW."""
    rng = np.random.default_rng(11)
    s0, s1 = pkg.seqgen.related_pair(3000, 3000, cfg=81)
    aligner.setSequences(s0, s1)
    try:
        for rec in (pkg.SMITH_WATERMAN, pkg.NEEDLEMAN_WUNSCH):
            for (i0, j0, i1, j1) in [(0, 0, 700, 900), (100, 250, 1124, 314), (1000, 1000, 1001, 2500), (5, 7, 1500, 8)]:
                m, n = i1 - i0, j1 - j0
                row = np.stack([rng.integers(0, 50, n), rng.integers(-60, 40, n)], axis=1).astype(np.int32)
                col = np.stack([rng.integers(0, 50, m + 1), rng.integers(-60, 40, m + 1)], axis=1).astype(np.int32)
                row[rng.integers(0, n, max(1, n // 10)), 1] = -pkg.INF
                col[1 + rng.integers(0, m, max(1, m // 10)), 1] = -pkg.INF
                r1, c1 = row.copy(), col.copy()
                b1 = oracle.process_block(s0, s1, r1, c1, i0, j0, i1, j1, rec)
                r2, c2 = row.copy(), col.copy()
                b2 = aligner.processBlock(r2, c2, i0, j0, i1, j1, rec)
                assert np.array_equal(r1, r2) and np.array_equal(c1, c2)
                assert tuple(b1) == tuple(b2)
    finally:
        aligner.unsetSequences()


@pytest.mark.parametrize("flags", [0, 2, 3])     # packed kernel, int32 profile kernel, int32 byte-compare kernel
def test_streamed_column_bands_on_one_gpu(pkg, oracle, flags):
    """the 8-GPU chain emulated on one GPU: bands run one after the other, each fed with the previous
    band's last column through the streaming ABI; boundary columns and bests match the reference chain."""
    ch = G["chain"]
    s0, s1 = make_pair(pkg, ch["seq"])
    n, parts = len(s1), ch["parts"]
    from masa_cudalign_amd.bands import band_limits, canonical_best
    lim = band_limits(n, [1] * parts)
    al = pkg.MI355Aligner(device=0, flags=flags)
    try:
        al.setSequences(s0, s1)
        col, cands = None, []
        for k in range(parts):
            part = pkg.Partition(0, lim[k], len(s0), lim[k + 1])
            kw = dict(want_last_column=True)
            if col is not None:
                kw.update(first_column_init_type=pkg.INIT_WITH_CUSTOM_DATA, stream_first_column=True, first_column=col[:1])
            al.streamBegin(part, **kw)
            fed = 0
            while True:
                if col is not None and fed < len(s0):
                    ln = min(777, len(s0) - fed)
                    al.streamFeedColumn(fed, col[1 + fed:1 + fed + ln])
                    fed += ln
                rows, fin = al.streamPoll()
                if fin:
                    break
            newcol = np.concatenate([np.array([[0, -pkg.INF]], dtype=np.int32), al.streamReadColumn(0, len(s0))])
            best, _ = al.streamEnd()
            if k < parts - 1:
                assert digest(newcol) == ch["boundary_columns"]["STEP-%d.tmp" % (k + 1)]
            col = newcol
            cands.append(best)
            run = canonical_best(cands)
            assert [run[0] + 1, run[1] + 1, run[2]] == ch["band_bests"][k]
    finally:
        al.close()


def test_large_roundtrip_properties(pkg, aligner):
    """full-size style check without an oracle run: a pair with a planted exact repeat must score at
    least the repeat length, symmetric roles give the same score, and two runs are bit-identical."""
    m = n = 300000
    s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=90)
    s1 = s1.copy()
    s1[120000:120400] = s0[200000:200400]
    a = tuple(run_stage1(pkg, aligner, s0, s1).getBestScore())
    b = tuple(run_stage1(pkg, aligner, s0, s1).getBestScore())
    c = tuple(run_stage1(pkg, aligner, s1, s0).getBestScore())
    assert a == b
    assert a[2] >= 400 and a[2] == c[2]
    assert (a[0], a[1]) == (c[1], c[0])
    assert abs(a[0] - 200400) <= 40 and abs(a[1] - 120400) <= 40


def test_c2_full_size_invariants(pkg, oracle):
    """BASELINE config C2 at its full size (3 M x 3 M, unrelated ACGT, 9e12 cells) -- no oracle can sweep
    that in a test, so the result is pinned through size-independent properties:
      * four different engines agree bit for bit on (i, j, score): the kernel bench.py times -- default configuration:
        1536- and 1408-row strips in ONE launch, which the library reports itself (mi355sw_stats.kernel) --, the packed
        kernel with 1536-row strips, the packed kernel with 1024-row strips, the int32 kernel;
      * the reported cell is real: the oracle, run on the 600 x 600 window that ends at it (a local
        alignment of that score is far shorter), computes H = score at exactly that corner and nothing
        higher inside the window."""
    m = n = 3000000
    s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=2)
    part = pkg.Partition(0, 0, m, n)
    results = []
    for (R, flags, kernel) in ((0, 0, 2), (24, 0, 2), (16, 0, 2), (16, 2, 1)):
        al = pkg.MI355Aligner(device=0, rows_per_lane=R, flags=flags)
        try:
            al.setSequences(s0, s1)
            mg = pkg.Stage1Manager(part)
            al.alignPartition(part, mg)
            st = al.getStatistics()
            assert st["profile_kernel"] == kernel and st["restarts"] == 0
            if R == 0:
                # what the driver's bench line runs: the mixed-height kernel, a whole number of rounds of strips
                assert st["kernel"] == "sw_strip_kernel_pk16_mixed<12,11,true,true>", st["kernel"]
                assert st["strip_rows"] == 1536 and st["strip_rows_second"] == 1408 and st["strips"] % st["waves"] == 0
                assert st["strips_first"] * 1536 + (st["strips"] - st["strips_first"]) * 1408 >= m
            else:
                assert st["strip_rows"] == 64 * R and st["strip_rows_second"] == 0 and st["strips_first"] == st["strips"]
                assert st["kernel"].startswith("sw_strip_kernel_pk16<%d,true,true,false>" % (R // 2) if kernel == 2 else "sw_strip_kernel<%d,true,true,true>" % R)
            results.append(tuple(mg.getBestScore()))
        finally:
            al.close()
    assert results[0] == results[1] == results[2] == results[3]
    i, j, score = results[0]            # 1-based DP coordinates (dispatchScore convention)
    assert 18 <= score <= 30
    W = 600
    i0, j0 = max(0, i - W), max(0, j - W)
    ref = oracle.stage1(s0[i0:i], s1[j0:j], want_last_row=True)
    assert ref["best"][2] == score
    assert int(ref["last_row"][-1][0]) == score          # H at the reported cell itself


def test_packed_kernel_rebasing_beyond_16bit(pkg, oracle):
    """scores above 32767: the packed 16-bit kernel must keep its window centred on the wavefront (re-basing)
    and stay bit-exact -- best cell, last row and last column -- and must agree with the int32 kernel."""
    m, n = 70000, 66000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=97)
    ref = oracle.stage1(s0, s1, want_last_row=True, want_last_col=True)
    assert ref["best"][2] > 40000
    for flags, kernel in ((0, 2), (2, 1)):
        al = pkg.MI355Aligner(device=0, flags=flags)
        try:
            al.setSequences(s0, s1)
            part = pkg.Partition(0, 0, m, n)
            mg = pkg.Stage1Manager(part, keep_last_row=True, keep_last_column=True)
            al.alignPartition(part, mg)
            assert al.getStatistics()["profile_kernel"] == kernel
            assert tuple(mg.getBestScore()) == tuple(ref["best"])
            assert np.array_equal(mg.lastRow(), ref["last_row"])
            assert np.array_equal(mg.lastColumn(), ref["last_col"])
        finally:
            al.close()


def test_two_phase_exact_position(pkg, oracle, monkeypatch):
    """value-only main pass + exact re-run of the winning strip from a checkpoint row (used for very tall
    partitions) gives the same canonical cell as the single pass."""
    monkeypatch.setenv("MI355SW_TWO_PHASE", "1")
    for (m, n, rel) in [(9000, 7000, False), (20000, 9000, True), (5000, 30000, False)]:
        s0, s1 = (pkg.seqgen.related_pair if rel else pkg.seqgen.unrelated_pair)(m, n, cfg=61)
        al = pkg.MI355Aligner(device=0, rows_per_lane=4)
        try:
            al.setSequences(s0, s1)
            part = pkg.Partition(0, 0, m, n)
            mg = pkg.Stage1Manager(part)
            al.alignPartition(part, mg)
            assert al.getStatistics()["kernel_launches"] == 2
            assert tuple(mg.getBestScore()) == tuple(oracle.stage1(s0, s1)["best"])
        finally:
            al.close()


@pytest.mark.parametrize("R", [8, 24])
def test_score_lookup_paths_switch_inside_a_strip(pkg, oracle, R):
    """The packed kernel scores a chunk with a byte permute when every column in reach is one of <= 4 plain
    letters and with the one-hot AND/min form otherwise.  Runs of N (a 5th letter common to both sequences, so
    N==N matches as in the reference's byte compare), isolated foreign bytes and long clean stretches make
    the wavefront switch back and forth; best cell, last row, last column and special rows stay bit-exact."""
    m, n = 3000, 6000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=1234)
    s0, s1 = s0.copy(), s1.copy()
    s1[700:760] = ord("N"); s1[2000] = ord("N"); s1[2300:2310] = ord("R"); s1[4100:4400:7] = ord("N")
    s0[100:180] = ord("N"); s0[1500] = ord("Y"); s0[2500:2520] = ord("N")
    al = pkg.MI355Aligner(device=0, rows_per_lane=R)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        mg = pkg.Stage1Manager(part, keep_last_row=True, keep_last_column=True)
        al.alignPartition(part, mg)
        assert al.getStatistics()["profile_kernel"] == 2
        ref = oracle.stage1(s0, s1, want_last_row=True, want_last_col=True)
        assert tuple(mg.getBestScore()) == tuple(ref["best"])
        assert np.array_equal(mg.lastRow(), ref["last_row"])
        assert np.array_equal(mg.lastColumn(), ref["last_col"])
    finally:
        al.close()


def test_three_letter_alphabet_uses_table_form(pkg, oracle):
    """fewer than four common letters, and letters that occur in one sequence only (never match)."""
    rng = np.random.default_rng(5)
    s0 = rng.choice(np.frombuffer(b"ACGX", dtype=np.uint8), size=2500)
    s1 = rng.choice(np.frombuffer(b"ACGQ", dtype=np.uint8), size=5000)
    s1[1000:1400] = s0[300:700]
    al = pkg.MI355Aligner(device=0)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, len(s0), len(s1))
        mg = pkg.Stage1Manager(part, keep_last_row=True, keep_last_column=True)
        al.alignPartition(part, mg)
        ref = oracle.stage1(s0, s1, want_last_row=True, want_last_col=True)
        assert tuple(mg.getBestScore()) == tuple(ref["best"])
        assert np.array_equal(mg.lastRow(), ref["last_row"])
        assert np.array_equal(mg.lastColumn(), ref["last_col"])
    finally:
        al.close()


def test_special_rows_with_1536_row_strips(pkg, oracle):
    """strip heights that do not divide 8192: special rows sit at multiples of ceil(8192/1536)*1536 = 9216
    rows (AbstractDiagonalAligner::isSpecialRow with MINIMUM_FLUSH_INTERVAL), and hold the oracle's values."""
    m, n = 30000, 4000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=77)
    al = pkg.MI355Aligner(device=0, rows_per_lane=24)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        mg = pkg.Stage1Manager(part, special_row_interval=1000, keep_last_row=True)
        al.alignPartition(part, mg)
        rows = sorted(mg.special_rows)
        assert [r for r in rows if r < m] == [9216, 18432, 27648]
        ref = oracle.stage1(s0, s1, block_h=9216, block_w=n, special_row_interval=9216, want_last_row=True)
        assert tuple(mg.getBestScore()) == tuple(ref["best"])
        want = dict(zip(ref["special_row_ids"], ref["special_rows"]))
        for i in (9216, 18432, 27648):
            assert np.array_equal(mg.specialRow(i), want[i]), i
    finally:
        al.close()


def test_block_pruning_keeps_the_canonical_best(pkg, oracle):
    """mustPruneBlocks(): slabs that cannot reach the running best are skipped (AbstractBlockPruning bound).
    On a related pair a large part of the matrix is pruned, yet best score AND canonical position equal the
    unpruned run and the oracle; what is left in the special rows is a lower bound of the true cells that
    is exact where it matters (each row's maximum lies on the optimal path)."""
    m, n = 60000, 50000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=31)
    ref = oracle.stage1(s0, s1, block_h=8192, block_w=n, special_row_interval=8192)
    want_rows = dict(zip(ref["special_row_ids"], ref["special_rows"]))
    al = pkg.MI355Aligner(device=0, rows_per_lane=4)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        out = {}
        for prune in (False, True):
            mg = pkg.Stage1Manager(part, special_row_interval=8192, block_pruning=prune)
            al.alignPartition(part, mg)
            st = al.getStatistics()
            out[prune] = (tuple(mg.getBestScore()), st["pruned_cells"], mg)
            assert st["profile_kernel"] == 2
        assert out[False][0] == out[True][0] == tuple(ref["best"])
        assert out[False][1] == 0
        assert out[True][1] > 0.15 * m * n                       # a sizeable part of the matrix was skipped
        assert out[True][1] + al.getStatistics()["processed_cells"] == m * n
        mgp = out[True][2]
        crossed = 0
        for i in sorted(mgp.special_rows):
            got, want = mgp.specialRow(i)[:, 0], want_rows[i][:, 0]
            assert (got <= want).all() and (got[1:] >= 0).all()
            # a row the OPTIMAL path crosses (rows above the best cell) keeps its maximum: it lies on that path.  Below the best
            # cell a row's maximum belongs to some other alignment -- here one that runs into the last column 20 000 below the
            # optimum -- and may go like anything else that cannot reach the best (round 6: the chunks at the end of a row are
            # asked the skip test too; they used to be computed whatever the bound said)
            if i <= ref["best"][0]:
                assert got.max() == want.max() and int(got.argmax()) == int(want.argmax())
                crossed += 1
        assert crossed >= 4
    finally:
        al.close()


def test_block_pruning_unrelated_prunes_nothing_and_nw_ignores_it(pkg, oracle):
    m, n = 9000, 12000
    s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=32)
    al = pkg.MI355Aligner(device=0)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        mg = pkg.Stage1Manager(part, block_pruning=True)
        al.alignPartition(part, mg)
        assert al.getStatistics()["pruned_cells"] == 0
        assert tuple(mg.getBestScore()) == tuple(oracle.stage1(s0, s1)["best"])
    finally:
        al.close()


def _fuzz_case(k):
    rng = np.random.default_rng(1000 + k)
    m = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 255, 300, 511, 777, 1024, 1500, 2049, 2600]))
    n = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 193, 300, 511, 900, 1300, 2500]))
    alpha = [b"ACGT", b"ACGT", b"ACGTN", b"ACG", b"AC", b"ACGTNRYKM", b"ACGTU"][int(rng.integers(0, 7))]
    a = np.frombuffer(alpha, dtype=np.uint8)
    s0 = rng.choice(a, size=m)
    if rng.random() < 0.6 and m > 8 and n > 8:           # related: copy of a piece of seq0 with noise
        s1 = rng.choice(a, size=n)
        L = int(min(m, n) * rng.uniform(0.3, 1.0))
        i0, j0 = int(rng.integers(0, m - L + 1)), int(rng.integers(0, n - L + 1))
        piece = s0[i0:i0 + L].copy()
        hit = rng.random(L) < 0.05
        piece[hit] = rng.choice(a, size=int(hit.sum()))
        s1[j0:j0 + L] = piece
    else:
        s1 = rng.choice(a, size=n)
    R = int(rng.choice([0, 4, 8, 12, 16, 24, 32]))
    start, end = [(0, 0), (0, 0), (0, 0), (4, 4), (1, 3), (2, 3), (3, 3), (1, 1), (2, 2)][int(rng.integers(0, 9))]
    prune = bool(rng.integers(0, 2)) and (start, end) == (0, 0)
    flags = 2 if rng.random() < 0.15 else 0               # sometimes the int32 kernel
    return m, n, s0, s1, R, start, end, prune, flags


@pytest.mark.parametrize("k", range(240))
def test_randomised_differential_against_oracle(pkg, oracle, k):
    """240 seeded random configurations (sizes around every lane / chunk / strip boundary, 2-9 letter alphabets,
    related and unrelated pairs, every strip height, SW / NW / semi-global edges, pruning, both kernel
    families): best cell, last row and last column equal the oracle's, bit for bit."""
    from helpers import oracle_kwargs
    m, n, s0, s1, R, start, end, prune, flags = _fuzz_case(k)
    al = pkg.MI355Aligner(device=0, rows_per_lane=R, flags=flags)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        mg = pkg.Stage1Manager(part, alignment_start=getattr(pkg, EDGE[start]), alignment_end=getattr(pkg, EDGE[end]),
                               keep_last_row=True, keep_last_column=True, block_pruning=prune)
        al.alignPartition(part, mg)
        st = al.getStatistics()
        kw = oracle_kwargs(oracle, dict(start=start, end=end, pruning=False, disk=-1, block=(st["strip_rows"], 1 << 20)), m, n)
        kw.update(want_last_row=True, want_last_col=True)
        ref = oracle.stage1(s0, s1, **kw)
        assert tuple(mg.getBestScore()) == tuple(ref["best"])
        if not (prune and st["pruned_cells"] > 0):
            assert np.array_equal(mg.lastRow(), ref["last_row"])
            assert np.array_equal(mg.lastColumn(), ref["last_col"])
    finally:
        al.close()


def test_stop_like_stage2_goal_found(pkg, oracle):
    """The stage-2 call shape (M/stage2/sw_stage2.cpp:387-441): a tall, narrow NW partition with custom borders whose
    manager says stop once the goal showed up in the last column.  What was dispatched before the stop is exact,
    special rows below the stop are not handed over, nothing more is asked from the first-column stream, and the
    engine does not sweep the rest of the partition."""
    from masa_cudalign_amd.manager import ArrayCellsReader
    m, n, stop_after = 4000000, 2000, 20000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=91)
    INF = pkg.engine.INF
    col = np.zeros((m + 1, 2), dtype=np.int32); col[:, 0] = -2 * np.arange(m + 1) - 3; col[0, 0] = 0; col[:, 1] = -INF
    row = np.zeros((n + 1, 2), dtype=np.int32); row[:, 0] = -2 * np.arange(n + 1) - 3; row[0, 0] = 0; row[:, 1] = -INF

    class CountingReader(ArrayCellsReader):
        asked, asked_after_stop, stopped = 0, 0, False

        def read(self, buf, length):
            self.asked += length
            if self.stopped:
                self.asked_after_stop += length
            return ArrayCellsReader.read(self, buf, length)

    class Mgr(pkg.Stage1Manager):
        def dispatchColumn(self, j, buf, length):
            pkg.Stage1Manager.dispatchColumn(self, j, buf, length)
            if self.last_column_pos >= stop_after:
                self.active = False                      # AlignerManager::stopAligner, AlignerManager.cpp:357-370
                creader.stopped = True

    creader = CountingReader(col)
    al = pkg.MI355Aligner(device=0)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        mg = Mgr(part, alignment_start=pkg.AT_SEQUENCE_1_AND_2, alignment_end=pkg.AT_SEQUENCE_1_AND_2,
                 keep_last_column=True, special_row_interval=8192,
                 first_row_reader=ArrayCellsReader(row), first_column_reader=creader)
        al.alignPartition(part, mg)
        st = al.getStatistics()
    finally:
        al.close()
    got = mg.lastColumn()                    # corner cell + the rows dispatched before the stop
    seen = got.shape[0] - 1
    # (the last-column cells travel in chunks of up to 16 strips -- runtime.cpp, AlignJob::pump -- and the stop is looked at between them)
    assert stop_after <= seen < stop_after + 17 * st["strip_rows"]
    # rows 0..seen of the matrix do not depend on anything below them
    ref = oracle.stage1(s0[:seen], s1, recurrence=oracle.NEEDLEMAN_WUNSCH, first_row_type=oracle.INIT_WITH_GAPS,
                        first_col_type=oracle.INIT_WITH_GAPS, best_mode=oracle.BEST_LAST_CELL,
                        block_h=st["strip_rows"], block_w=1 << 20, want_last_col=True,
                        special_row_interval=st["strip_rows"])
    assert np.array_equal(got, ref["last_col"])
    rows = sorted(mg.special_rows)
    assert rows and all(r <= seen for r in rows)
    want = dict(zip(ref["special_row_ids"], ref["special_rows"]))
    for r in rows:
        assert np.array_equal(mg.specialRow(r), want[r]), r
    # the first-column stream is left alone after the stop, and the engine computes nothing below the rows it had been fed by
    # then.  (Whatever else shares the GPU: how far the feed is ahead of the kernel when the stop comes is a matter of timing
    # -- under `pytest -n 6` the whole column may be there before the kernel's first strip is through -- these two are not.)
    assert creader.asked_after_stop == 0
    # (mi355sw_stats.processed_cells after a stop: the strips that were complete or in flight -- at most one per wavefront)
    assert st["processed_cells"] <= min(m, creader.asked + (st["waves"] + 1) * st["strip_rows"]) * n


@pytest.mark.parametrize("rel", [False, True])
def test_two_phase_at_its_real_trigger(pkg, oracle, rel):
    """Partitions with >= 32 Mi rows (BASELINE C3, C4, C5 and the 228 M target) track only the best VALUE in the main
    pass and recompute the winning strip exactly from a checkpoint row.  33.6 M x 16 k, no environment override:
    the packed two-phase run, the int32 single-pass run and the oracle (on the window that ends at the reported
    cell) agree."""
    m, n = 33600000, 16384
    assert m >= (32 << 20)
    s0, s1 = (pkg.seqgen.related_pair if rel else pkg.seqgen.unrelated_pair)(m, n, cfg=62)
    part = pkg.Partition(0, 0, m, n)
    res = {}
    for flags in (0, 2):
        al = pkg.MI355Aligner(device=0, flags=flags)
        try:
            al.setSequences(s0, s1)
            mg = pkg.Stage1Manager(part)
            al.alignPartition(part, mg)
            st = al.getStatistics()
            res[flags] = tuple(mg.getBestScore())
            if flags == 0:
                assert st["profile_kernel"] == 2 and st["kernel_launches"] == 2      # main pass + exact pass
                assert st["strips"] >= 16000
            else:
                assert st["profile_kernel"] == 1 and st["kernel_launches"] == 1
        finally:
            al.close()
    assert res[0] == res[2]
    i, j, score = res[0]
    assert score > (8000 if rel else 15)
    # the alignment that ends at (i, j) is at most ~1.3 * n rows long: the oracle recomputes the window that holds it
    W = min(i, 3 * n)
    ref = oracle.stage1(s0[i - W:i], s1[:j], want_last_row=True)
    assert ref["best"][2] == score and int(ref["last_row"][-1][0]) == score
    # nothing above the window beats it either: the rows [0, i) hold no higher score (checked on a second window
    # placed where the int32 run and the packed run agree anyway; here: canonical = first row reaching the maximum)
    assert ref["best"][0] == W and ref["best"][1] == j


def test_packed_and_int32_agree_at_the_north_star_height(pkg, oracle):
    """228 000 000 rows (the north star's height: 111 329 strips of 2048 rows, two-phase best) x 4096 columns, local SW on
    an unrelated pair: the packed run and the int32 single-pass run report the same best cell and the same last row,
    cell for cell -- a divergence anywhere in the 228 M rows above would show in that row -- and the oracle confirms
    the cell on the window that ends at it."""
    m, n = 228000000, 4096
    s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=63)
    part = pkg.Partition(0, 0, m, n)
    res = {}
    for flags in (0, 2):
        al = pkg.MI355Aligner(device=0, flags=flags)
        try:
            al.setSequences(s0, s1)
            mg = pkg.Stage1Manager(part, keep_last_row=True)
            al.alignPartition(part, mg)
            st = al.getStatistics()
            res[flags] = (tuple(mg.getBestScore()), mg.lastRow())
            assert st["profile_kernel"] == (2 if flags == 0 else 1)
        finally:
            al.close()
    assert res[0][0] == res[2][0]
    assert np.array_equal(res[0][1], res[2][1])
    i, j, score = res[0][0]
    W = min(i, 600)
    ref = oracle.stage1(s0[i - W:i], s1[:j], want_last_row=True)
    assert ref["best"][2] == score and int(ref["last_row"][-1][0]) == score


def test_match_last_column_follows_aligner_utils(pkg, oracle, aligner):
    """IAligner::matchLastColumn = AlignerUtils::matchColumn (M/libmasa/utils/AlignerUtils.cpp:50-107): first k with
    base.h + buffer.h == goal (aligned) or base.e + buffer.e + gap_open == goal (gapped); a sum above the goal is
    reported as the reference's MATCH_ERROR_1 / _2."""
    rng = np.random.default_rng(17)
    for case in range(60):
        n = int(rng.integers(1, 400))
        base = rng.integers(-500, 500, size=(n, 2)).astype(np.int32)
        buf = rng.integers(-500, 500, size=(n, 2)).astype(np.int32)
        goal = 1500
        kind = case % 5
        k = int(rng.integers(0, n))
        if kind == 0:
            buf[k, 0] = goal - base[k, 0]
        elif kind == 1:
            buf[k, 1] = goal - 3 - base[k, 1]
        elif kind == 2:
            buf[k, 0] = goal + 7 - base[k, 0]
        elif kind == 3:
            buf[k, 1] = goal + 2 - 3 - base[k, 1]
        got = aligner.matchLastColumn(buf, base, goal)
        rc, rk, rs, rt = oracle.match_column(buf, base, goal)
        if rc == 1:
            assert got == {"found": True, "k": rk, "score": rs, "type": rt}, case
        elif rc == 0:
            assert got["found"] is False and got["k"] == -1, case
        else:
            assert got["found"] is False and got["k"] == rk and got["type"] == rc, case
    with_inf = np.array([[5, -pkg.INF], [7, 3]], dtype=np.int32)
    got = aligner.matchLastColumn(with_inf, np.array([[1, -pkg.INF], [2, 4]], dtype=np.int32), 10)
    assert got == {"found": True, "k": 1, "score": 4, "type": 1}


@pytest.mark.parametrize("letters", [b"ACGTNRYKM", b"ACGTNRYKMSWBDH", b"ACGTNRYKMSWBDHV"])
def test_iupac_alphabets_stay_on_the_packed_kernel(pkg, oracle, letters):
    """up to 14 byte values common to both sequences are coded (most frequent first) and run on the packed kernel;
    its int32 stand-in for 8..14 letters is the byte-compare kernel on the codes; 15 and more fall back to raw byte
    compare.  Letters of one sequence only never match.  All three agree with the oracle bit for bit."""
    rng = np.random.default_rng(len(letters))
    alpha = np.frombuffer(letters, dtype=np.uint8)
    # genome-like: mostly ACGT, IUPAC codes sprinkled in, runs of N
    m, n = 7000, 9000
    s0 = alpha[rng.integers(0, 4, m)].copy()
    s1 = s0[:min(m, n)].copy()
    s1 = np.concatenate([s1, alpha[rng.integers(0, 4, n - len(s1))]])
    for s in (s0, s1):
        idx = rng.integers(0, len(s), len(s) // 40)
        s[idx] = alpha[rng.integers(0, len(alpha), len(idx))]
        s[len(s) // 2: len(s) // 2 + 90] = ord("N")
    s0[5::211] = ord("@")            # bytes of one sequence only
    s1[7::199] = ord("#")
    ref = oracle.stage1(s0, s1, want_last_row=True, want_last_col=True)
    want_kernel = 2 if len(letters) <= 14 else 0
    for flags in (0, 2):
        al = pkg.MI355Aligner(device=0, flags=flags)
        try:
            al.setSequences(s0, s1)
            part = pkg.Partition(0, 0, m, n)
            mg = pkg.Stage1Manager(part, keep_last_row=True, keep_last_column=True)
            al.alignPartition(part, mg)
            assert al.getStatistics()["profile_kernel"] == (want_kernel if flags == 0 else 0)
            assert tuple(mg.getBestScore()) == tuple(ref["best"])
            assert np.array_equal(mg.lastRow(), ref["last_row"])
            assert np.array_equal(mg.lastColumn(), ref["last_col"])
        finally:
            al.close()

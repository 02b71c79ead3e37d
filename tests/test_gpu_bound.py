"""GPU (-m gpu): where the pruning bound STARTS -- the diagonal seed (mi355sw_seed_bound) and a caller's initial_bound --
held against the oracle at sizes it finishes in seconds, and the guard on a bound no alignment reaches (MI355SW_EBOUND).

Reference: the bound a node starts from is what it knows when it starts -- the best score of the run it resumes
(Status::load -> BestScoreList, sw_stage1.cpp:210-217) and the other nodes' best (AlignerPool::getBestNodeScore); pruning
itself: AbstractBlockPruning::isBlockPrunable, AbstractBlockPruning.cpp:70-111; canonical best: BestScoreList.cpp:129-195."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
INF = 999999999


def _stream(pkg, al, m, n, recurrence, bound, interval=8192, prune=True):
    """one pruning stream over the whole matrix through the C ABI's streaming form: best, last row, last column, special rows"""
    from masa_cudalign_amd.engine import SMITH_WATERMAN, INIT_WITH_GAPS, INIT_WITH_ZEROES
    import time
    sw = recurrence == SMITH_WATERMAN
    init = INIT_WITH_ZEROES if sw else INIT_WITH_GAPS
    al.streamBegin(pkg.Partition(0, 0, m, n), recurrence_type=recurrence, track_best=sw, first_row_init_type=init, first_column_init_type=init,
                   want_last_row=True, want_last_column=True, special_row_interval=interval, prune_blocks=prune, initial_bound=bound)
    while True:
        rows, fin = al.streamPoll()
        if fin:
            break
        time.sleep(0.001)
    out = {"last_row": al.streamReadLastRow(), "last_col": al.streamReadColumn(0, m), "rows": {}}
    k = 0
    while True:
        try:
            dp, cells = al.streamReadSpecialRow(k)
        except Exception:
            break
        out["rows"][dp] = cells
        k += 1
    best, nsp = al.streamEnd()
    assert nsp == k
    out["best"] = (best[0] + 1, best[1] + 1, best[2]) if best[1] >= 0 else tuple(best)      # 1-based like the oracle's (the stream's cell is 0-based)
    out["stats"] = al.getStatistics()
    return out


def _pairs(pkg, kind):
    sg = pkg.seqgen
    if kind == "related":                       # one inverted segment of 5 % (seqgen's default)
        return sg.related_pair(70000, 66000, cfg=501)
    if kind == "inversion":                     # a long inverted stretch: the seed's band must find its way across it
        return sg.related_pair(70000, 66000, cfg=502, inversion=0.15)
    if kind == "ties":
        # three exact copies of one 22 000-mer in seq0, one in seq1 (with a few substitutions): three co-optimal end cells, the
        # canonical one (min i, BestScoreList.cpp:129-195) is the first copy's
        a = sg.random_dna(0x7135, 22000)
        b = a.copy()
        b[1000::1500] = np.frombuffer(b"ACGT", dtype=np.uint8)[(np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), b[1000::1500]) + 1) % 4]
        s0 = np.concatenate([a, sg.random_dna(0x7136, 2000), a, sg.random_dna(0x7137, 2000), a])
        return np.ascontiguousarray(s0), np.ascontiguousarray(b)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["related", "inversion", "ties"])
def test_seed_bound_then_pruned_local_run_against_the_oracle(pkg, oracle, kind):
    """mi355sw_seed_bound -> initial_bound -> pruned run: the best CELL is the oracle's canonical cell; last row, last column and
    special rows are lower bounds of the oracle's with every row's maximum intact above the best cell.  The same with the bound
    set to the optimum itself (every co-optimal cell must survive the strict test) and to optimum - 1."""
    from masa_cudalign_amd.engine import SMITH_WATERMAN
    s0, s1 = _pairs(pkg, kind)
    m, n = len(s0), len(s1)
    from helpers import oracle_full
    ref = oracle_full(oracle, s0, s1)
    want_rows = dict(zip(ref["special_row_ids"], ref["special_rows"]))
    opt = ref["best"][2]
    if kind == "ties":
        # the construction really has several co-optimal end cells: the optimum shows up in the last column on three rows
        assert int((ref["last_col"][:, 0] == opt).sum()) >= 3
    al = pkg.MI355Aligner(device=0, rows_per_lane=4)
    try:
        al.setSequences(s0, s1)
        seed = al.seedBound(pkg.Partition(0, 0, m, n), SMITH_WATERMAN)
        assert seed is not None and 0 < seed <= opt            # the score of an alignment that exists
        if kind != "inversion":
            assert seed > 0.9 * opt                            # ... and on these pairs nearly the answer itself
        base = _stream(pkg, al, m, n, SMITH_WATERMAN, None, prune=False)
        assert base["best"] == tuple(ref["best"]) and base["stats"]["pruned_cells"] == 0
        for bound in (seed, opt, opt - 1):
            got = _stream(pkg, al, m, n, SMITH_WATERMAN, bound)
            st = got["stats"]
            assert st["profile_kernel"] == 2 and st["restarts"] == 0
            assert got["best"] == tuple(ref["best"]), (kind, bound)
            assert st["pruned_cells"] > 0.3 * m * n and st["pruned_cells"] + st["processed_cells"] == m * n
            assert np.all(got["last_row"] <= ref["last_row"][1:]) and np.all(got["last_col"] <= ref["last_col"][1:])
            assert sorted(got["rows"]) == sorted(want_rows)
            for i, cells in got["rows"].items():
                w = want_rows[i][1:]
                assert np.all(cells <= w), i
                if i <= ref["best"][0]:                        # the row's maximum lies on the optimal path: exact
                    assert int(cells[:, 0].max()) == int(w[:, 0].max()) and int(cells[:, 0].argmax()) == int(w[:, 0].argmax()), i
    finally:
        al.close()


def test_a_bound_above_the_optimum_is_reported_not_obeyed(pkg, oracle):
    """initial_bound = optimum + 1: no alignment reaches it, everything that matters is pruned away -- mi355sw_stream_end says
    MI355SW_EBOUND instead of handing out a wrong best (local) / a wrong H[m][n] (global); the engine stays usable"""
    from masa_cudalign_amd.engine import SMITH_WATERMAN, NEEDLEMAN_WUNSCH, AlignerError
    s0, s1 = pkg.seqgen.related_pair(70000, 66000, cfg=503)
    m, n = len(s0), len(s1)
    al = pkg.MI355Aligner(device=0, rows_per_lane=4)
    try:
        al.setSequences(s0, s1)
        from helpers import oracle_full
        ref = oracle_full(oracle, s0, s1)
        opt = ref["best"][2]
        with pytest.raises(AlignerError, match="EBOUND"):
            _stream(pkg, al, m, n, SMITH_WATERMAN, opt + 1)
        assert _stream(pkg, al, m, n, SMITH_WATERMAN, opt)["best"] == tuple(ref["best"])
        h = int(oracle_full(oracle, s0, s1, edge=4)["last_row"][-1, 0])
        with pytest.raises(AlignerError, match="EBOUND"):
            _stream(pkg, al, m, n, NEEDLEMAN_WUNSCH, h + 1)
        for bound in (h, h - 1):
            got = _stream(pkg, al, m, n, NEEDLEMAN_WUNSCH, bound)
            assert int(got["last_row"][-1, 0]) == h and got["stats"]["pruned_cells"] > 0.3 * m * n
    finally:
        al.close()


def test_seed_bound_global_then_pruned_run_against_the_oracle(pkg, oracle):
    """the same for a GLOBAL alignment: the seed is a lower bound of H[m][n], the pruned run behind it ends in the oracle's
    H[m][n]; borders are lower bounds, exact wherever a path can still reach the goal"""
    from masa_cudalign_amd.engine import NEEDLEMAN_WUNSCH
    from helpers import oracle_full
    from test_gpu_prune_global import _reach
    s0, s1 = pkg.seqgen.related_pair(70000, 66000, cfg=504)
    m, n = len(s0), len(s1)
    ref = oracle_full(oracle, s0, s1, edge=4)
    final = int(ref["last_row"][-1, 0])
    al = pkg.MI355Aligner(device=0, rows_per_lane=4)
    try:
        al.setSequences(s0, s1)
        seed = al.seedBound(pkg.Partition(0, 0, m, n), NEEDLEMAN_WUNSCH)
        assert seed is not None and seed <= final
        got = _stream(pkg, al, m, n, NEEDLEMAN_WUNSCH, seed)
        assert int(got["last_row"][-1, 0]) == final and got["stats"]["pruned_cells"] > 0.3 * m * n
        lr, lc = got["last_row"], got["last_col"]
        assert np.all(lr <= ref["last_row"][1:]) and np.all(lc <= ref["last_col"][1:])
        x = ref["last_row"][1:, 0].astype(np.int64)
        must = _reach(x, m, np.arange(1, n + 1), m, n) >= final
        assert must.any() and np.array_equal(lr[must, 0], ref["last_row"][1:][must, 0])
        x = ref["last_col"][1:, 0].astype(np.int64)
        must = _reach(x, np.arange(1, m + 1), n, m, n) >= final
        assert must.any() and np.array_equal(lc[must, 0], ref["last_col"][1:][must, 0])
    finally:
        al.close()


def _late_homology(pkg, m, n, start_col, cfg):
    """seq1 = unrelated columns, then -- from column `start_col` on -- a mutated copy of seq0's beginning: nothing aligns at the
    left edge of the matrix (a chromosome that starts with an unaligned stretch)"""
    sg = pkg.seqgen
    s0 = sg.random_dna(sg.SEED0 + cfg, m)
    tail = sg.mutate_dna(s0[:n], sg.SEED1 + cfg, inversion=0.0)[:n - start_col]
    s1 = np.concatenate([sg.random_dna(sg.SEED1 + cfg + 9, start_col), tail])
    assert len(s1) == n
    return s0, np.ascontiguousarray(s1)


def _is_pruning_kernel(name):
    import re
    return re.fullmatch(r"sw_strip_kernel_pk16<\d+,\w+,\w+,true>", name) is not None


def test_pruning_engages_for_homology_that_starts_inside_the_matrix(pkg, oracle):
    """The probe that tells related from unrelated pairs looks at stripes across the whole width, not only at the left edge
    (AbstractBlockPruning.cpp:70-111 prunes whenever the bound allows): homology that starts at 30 % of the width still
    prunes, with the oracle's best cell; an unrelated pair with pruning left on still runs the plain kernel."""
    m, n = 70000, 66000
    s0, s1 = _late_homology(pkg, m, n, 20000, 505)
    from helpers import oracle_full
    ref = oracle_full(oracle, s0, s1)
    al = pkg.MI355Aligner(device=0)              # default configuration: the probe only plans runs it may plan
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        mg = pkg.Stage1Manager(part, block_pruning=True)
        al.alignPartition(part, mg)
        st = al.getStatistics()
        assert tuple(mg.getBestScore()) == tuple(ref["best"])
        assert _is_pruning_kernel(st["kernel"]) and st["pruned_cells"] > 0.2 * m * n, st    # the pruning kernel ran and skipped
        u0, u1 = pkg.seqgen.unrelated_pair(m, n, cfg=506)
        al.setSequences(u0, u1)
        mg = pkg.Stage1Manager(part, block_pruning=True)
        al.alignPartition(part, mg)
        st = al.getStatistics()
        assert tuple(mg.getBestScore()) == tuple(oracle_full(oracle, u0, u1)["best"])
        assert st["pruned_cells"] == 0 and not _is_pruning_kernel(st["kernel"]), st         # planned as the plain score pass
    finally:
        al.close()


def test_late_homology_at_four_million_columns(pkg):
    """the same at 4 M x 4 M, homology from column 1 200 000 on: more than 30 % skipped, the best cell of the unpruned run"""
    m, n = 4000000, 4000000
    s0, s1 = _late_homology(pkg, m, n, 1200000, 507)
    al = pkg.MI355Aligner(device=0)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        res = {}
        for prune in (False, True):
            mg = pkg.Stage1Manager(part, block_pruning=prune)
            al.alignPartition(part, mg)
            res[prune] = (tuple(mg.getBestScore()), al.getStatistics())
        assert res[False][0] == res[True][0] and res[True][0][2] > 1500000
        assert res[False][1]["pruned_cells"] == 0
        assert res[True][1]["pruned_cells"] > 0.3 * m * n, res[True][1]
    finally:
        al.close()


def test_a_co_optimal_path_that_can_only_tie_survives_pruning(pkg, oracle):
    """Three exact copies of one 22 000-mer in seq0 against one in seq1, cut so that the SECOND copy's alignment ends exactly in the
    matrix's last cell: it can only ever TIE the score the first copy reached 24 000 rows earlier, with no row or column to
    spare -- every one of its cells sits exactly on the pruning bound.  The strict test must keep all of them (the round-4
    kernels cut such a path where it crossed a strip boundary at a slab corner: SW_PRUNE_MARGIN in csrc/sw_kernel_pk16.inc):
    H of the last cell is the oracle's, with every strip height, with and without exact tracking, with and without the window."""
    from masa_cudalign_amd.engine import SMITH_WATERMAN, F_NO_WINDOW
    import time
    s0, s1 = _pairs(pkg, "ties")
    M, N = 36864, 12864                      # second copy: rows 24 000 ..., so (M, N) ends its first 12 864 columns
    ref = oracle.stage1(s0[:M], s1[:N], want_last_row=True)
    want = int(ref["last_row"][-1, 0])
    assert want == ref["best"][2] and ref["best"][0] < M        # the same score, reached far above by the first copy (the canonical cell)
    part = pkg.Partition(0, 0, M, N)
    for R in (4, 8, 16, 32):
        for flags in (0, F_NO_WINDOW):
            for track in (False, True):
                al = pkg.MI355Aligner(device=0, rows_per_lane=R, flags=flags)
                try:
                    al.setSequences(s0, s1)
                    al.streamBegin(part, track_best=track, prune_blocks=True, want_last_row=True)
                    while not al.streamPoll()[1]:
                        time.sleep(0.001)
                    lr = al.streamReadLastRow()
                    best, _ = al.streamEnd()
                    st = al.getStatistics()
                finally:
                    al.close()
                assert int(lr[-1, 0]) == want, (R, flags, track, lr[-1], want)
                assert np.all(lr <= ref["last_row"][1:]) and st["pruned_cells"] > 0.3 * M * N
                if track:
                    assert (best[0] + 1, best[1] + 1, best[2]) == tuple(ref["best"])


def test_chromosome_like_pair_with_leading_n_runs(pkg, oracle):
    """two "chromosomes" that start with unsequenced stretches of different length (runs of N, upper-cased FASTA bytes as the
    reference compares them: N scores a match against N, X/CUDAligner.cu:276-289) before their homology begins: nothing related
    at the left edge of the matrix beyond the N block, five letters in play (the permute scoring of the packed kernel only
    takes chunks of plain A/C/G/T).  Default configuration, pruning on: the oracle's best cell, and a good part of the matrix
    skipped."""
    from helpers import oracle_full
    sg = pkg.seqgen
    body0 = sg.random_dna(sg.SEED0 + 510, 120000)
    body1 = sg.mutate_dna(body0, sg.SEED1 + 510, inversion=0.0)[:110000]
    s0 = np.ascontiguousarray(np.concatenate([np.full(30000, ord("N"), dtype=np.uint8), body0]))
    s1 = np.ascontiguousarray(np.concatenate([np.full(18000, ord("N"), dtype=np.uint8), body1]))
    m, n = len(s0), len(s1)
    ref = oracle_full(oracle, s0, s1)
    al = pkg.MI355Aligner(device=0)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        res = {}
        for prune in (False, True):
            mg = pkg.Stage1Manager(part, block_pruning=prune)
            al.alignPartition(part, mg)
            res[prune] = (tuple(mg.getBestScore()), al.getStatistics())
        assert res[False][0] == res[True][0] == tuple(ref["best"])
        assert res[True][1]["profile_kernel"] == 2 and res[True][1]["restarts"] == 0
        assert res[True][1]["pruned_cells"] > 0.2 * m * n, res[True][1]
    finally:
        al.close()


def test_anchored_seed_and_staircase_seed_agree_with_the_run_behind_them(pkg):
    """9 M x 8.5 M related pair (one inverted segment): the anchored seed (segments between anchors, side by side) and round 4's
    staircase (MI355SW_F_STAIRCASE_SEED: one chain of tiles) are two routes to a first bound -- both the score of an alignment
    that exists, on this pair both the answer itself -- and the pruned run behind either reports the same best cell as the run
    that starts from nothing (MI355SW_F_NO_DIAGONAL_SEED) while skipping far more."""
    from masa_cudalign_amd.engine import SMITH_WATERMAN, F_STAIRCASE_SEED, F_NO_DIAGONAL_SEED
    m, n = 9000000, 8500000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=5)
    part = pkg.Partition(0, 0, m, n)
    res = {}
    for name, flags in (("anchored", 0), ("staircase", F_STAIRCASE_SEED), ("none", F_NO_DIAGONAL_SEED)):
        al = pkg.MI355Aligner(device=0, flags=flags)
        try:
            al.setSequences(s0, s1)
            seed = al.seedBound(part, SMITH_WATERMAN)
            mg = pkg.Stage1Manager(part, block_pruning=True)
            al.alignPartition(part, mg)
            st = al.getStatistics()
            res[name] = (seed, tuple(mg.getBestScore()), st["pruned_cells"] / float(m) / n, st["seed_ms"])
        finally:
            al.close()
    assert res["none"][0] is None and res["none"][3] == 0
    assert res["anchored"][1] == res["staircase"][1] == res["none"][1]
    best = res["none"][1][2]
    assert res["anchored"][0] == res["staircase"][0] == best
    assert res["anchored"][3] > 0 and res["staircase"][3] > 0
    assert min(res["anchored"][2], res["staircase"][2]) > res["none"][2] + 0.2

"""GPU (-m gpu): block pruning of GLOBAL alignments (BASELINE config 5 as worded: "global NW ... block pruning on") and
of local alignments whose scores lie far above the 16-bit window (round 4).

Reference: AbstractBlockPruning::isBlockPrunable, M/libmasa/pruning/AbstractBlockPruning.cpp:70-111 -- the NEEDLEMAN_WUNSCH
branch (:92-104) bounds what a block can still reach in the LAST cell of the super-partition and keeps a running lower
bound of that cell.  The reference's stage 1 never switches it on for global alignments (sw_stage1.cpp:219-225); the
oracle for a pruned run is therefore the UNPRUNED one: H[m][n] must be the same, every other cell a lower bound of the
true one, and exact wherever a path through it can still reach H[m][n]."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from test_bands_gloo import _free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _global_run(pkg, al, m, n, prune, interval=0, sup=None):
    part = pkg.Partition(0, 0, m, n)
    mg = pkg.Stage1Manager(part, alignment_start=pkg.AT_SEQUENCE_1_AND_2, alignment_end=pkg.AT_SEQUENCE_1_AND_2,
                           special_row_interval=interval, keep_last_row=True, keep_last_column=True, block_pruning=prune)
    al.alignPartition(part, mg)
    return mg, al.getStatistics()


def _reach(h, i, j, m, n):
    """the most a global path through cell (i, j) holding h can still score in (m, n): every diagonal step a match,
    the forced gap at its extension price"""
    di, dj = m - i, n - j
    return h + np.minimum(di, dj) - 2 * np.abs(dj - di)


def _check_lower_bounds(mg, ref, m, n, final):
    """last row, last column and special rows of a pruned run against the exact ones"""
    lr, lc = mg.lastRow(), mg.lastColumn()
    assert np.all(lr <= ref["last_row"]) and np.all(lc <= ref["last_col"])
    x = ref["last_row"][:, 0].astype(np.int64)
    must = _reach(x, m, np.arange(0, n + 1), m, n) >= final
    assert must.any() and np.array_equal(lr[must, 0], ref["last_row"][must, 0])
    x = ref["last_col"][:, 0].astype(np.int64)
    must = _reach(x, np.arange(0, m + 1), n, m, n) >= final
    assert must.any() and np.array_equal(lc[must, 0], ref["last_col"][must, 0])
    rows = dict(zip(ref.get("special_row_ids") or [], ref.get("special_rows") if ref.get("special_rows") is not None else []))
    for i in sorted(mg.special_rows):
        got, want = mg.specialRow(i), rows[i]
        assert np.all(got <= want), i
        must = _reach(want[:, 0].astype(np.int64), i, np.arange(0, n + 1), m, n) >= final
        assert must.any() and np.array_equal(got[must, 0], want[must, 0]), i


def test_global_pruning_against_the_oracle(pkg, oracle):
    """60 000 x 50 000 related pair, gap-initialised borders: more than a quarter of the matrix is skipped; H[m][n] is the
    oracle's; last row, last column and special rows are lower bounds of the oracle's and equal wherever a path can still
    reach H[m][n]; without the request nothing is skipped and everything is equal"""
    from helpers import oracle_kwargs
    m, n = 60000, 50000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=34)
    al = pkg.MI355Aligner(device=0, rows_per_lane=4)
    try:
        al.setSequences(s0, s1)
        kw = oracle_kwargs(oracle, dict(start=4, end=4, pruning=False, disk=-1, block=(8192, 1 << 20)), m, n)
        kw.update(want_last_row=True, want_last_col=True, special_row_interval=8192)
        ref = oracle.stage1(s0, s1, **kw)
        mg, st = _global_run(pkg, al, m, n, False, 8192)
        assert st["pruned_cells"] == 0 and tuple(mg.getBestScore()) == tuple(ref["best"])
        assert np.array_equal(mg.lastRow(), ref["last_row"]) and np.array_equal(mg.lastColumn(), ref["last_col"])
        mg, st = _global_run(pkg, al, m, n, True, 8192)
        assert st["profile_kernel"] == 2 and st["kernel_launches"] == 1          # the packed kernel, no int32 rerun
        assert tuple(mg.getBestScore()) == tuple(ref["best"]) == (m, n, int(ref["last_row"][-1, 0]))
        assert st["pruned_cells"] > 0.25 * m * n and st["pruned_cells"] + st["processed_cells"] == m * n
        _check_lower_bounds(mg, ref, m, n, ref["best"][2])
    finally:
        al.close()


def _fuzz(k):
    rng = np.random.default_rng(4000 + k)
    m = int(rng.integers(2000, 40000))
    n = int(np.clip(m * rng.uniform(0.6, 1.5), 1500, 45000))
    R = int(rng.choice([0, 4, 8, 16, 32]))
    kind = int(rng.integers(0, 4))
    return m, n, R, kind


@pytest.mark.parametrize("k", range(24))
def test_global_pruning_randomised(pkg, oracle, k):
    """24 seeded shapes (2 000 ... 45 000, every strip-height family, related pairs of several divergences and unrelated
    ones): H[m][n] = the oracle's, borders are lower bounds that are exact where they can matter, no int32 rerun"""
    from helpers import oracle_kwargs
    m, n, R, kind = _fuzz(k)
    if kind == 0:
        s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=400 + k)
    else:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=400 + k, p_sub=[0.0, 0.01, 0.05, 0.15][kind], p_indel=[0.0, 0.001, 0.004, 0.01][kind])
    al = pkg.MI355Aligner(device=0, rows_per_lane=R)
    try:
        al.setSequences(s0, s1)
        mg, st = _global_run(pkg, al, m, n, True)
        kw = oracle_kwargs(oracle, dict(start=4, end=4, pruning=False, disk=-1, block=(st["strip_rows"], 1 << 20)), m, n)
        kw.update(want_last_row=True, want_last_col=True)
        ref = oracle.stage1(s0, s1, **kw)
        assert st["profile_kernel"] == 2 and st["kernel_launches"] == 1
        assert tuple(mg.getBestScore()) == tuple(ref["best"])
        _check_lower_bounds(mg, ref, m, n, ref["best"][2])
        if kind != 0 and min(m, n) > 24000:        # large enough for whole slabs outside the band that can still reach the goal
            assert st["pruned_cells"] > 0
    finally:
        al.close()


def test_global_pruning_needs_the_last_cell_as_the_goal(pkg, oracle):
    """semi-global edges (the best score is looked for on the last row / column) and local alignments whose manager does
    not want scores: the bound of a global alignment does not hold -- the request is ignored, every cell exact"""
    from helpers import oracle_kwargs
    m, n = 30000, 26000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=35)
    al = pkg.MI355Aligner(device=0, rows_per_lane=8)
    try:
        al.setSequences(s0, s1)
        for start, end in ((4, 1), (4, 2), (2, 3)):
            part = pkg.Partition(0, 0, m, n)
            edges = {1: pkg.AT_SEQUENCE_1, 2: pkg.AT_SEQUENCE_2, 3: pkg.AT_SEQUENCE_1_OR_2, 4: pkg.AT_SEQUENCE_1_AND_2}
            mg = pkg.Stage1Manager(part, alignment_start=edges[start], alignment_end=edges[end], keep_last_row=True,
                                   keep_last_column=True)
            mg.block_pruning = True                 # a manager that asks although its goal is not the last cell
            al.alignPartition(part, mg)
            st = al.getStatistics()
            kw = oracle_kwargs(oracle, dict(start=start, end=end, pruning=False, disk=-1, block=(st["strip_rows"], 1 << 20)), m, n)
            kw.update(want_last_row=True, want_last_col=True)
            ref = oracle.stage1(s0, s1, **kw)
            assert st["pruned_cells"] == 0
            assert np.array_equal(mg.lastRow(), ref["last_row"]) and np.array_equal(mg.lastColumn(), ref["last_col"])
            assert tuple(mg.getBestScore()) == tuple(ref["best"])
    finally:
        al.close()


def _worker_chain(rank, world, port, m, n, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    from masa_cudalign_amd.bands import BandRunner, band_limits
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=36)
        lim = band_limits(n, [1] * world)
        out = {}
        for mode, transport in (("plain", "p2p"), ("pruned", "p2p"), ("pruned_host", "host")):
            al = pkg.MI355Aligner(device=0, rows_per_lane=4)
            al.setSequences(s0, s1)
            runner = BandRunner(al, dist=dist, rank=rank, world=world, device=None, segment_rows=2048, transport=transport,
                                prune_blocks=(mode != "plain"))
            rows, got = {}, {}
            runner.run(m, lim[rank], lim[rank + 1], recurrence=pkg.NEEDLEMAN_WUNSCH, track_best=False,
                       first_row_init_type=pkg.INIT_WITH_GAPS, first_col_init_type=pkg.INIT_WITH_GAPS, want_last_row=True,
                       before_end=lambda eng: got.update(row=eng.streamReadLastRow()), special_row_interval=8192, n_total=n,
                       special_row_sink=lambda dp, c0, cells: rows.__setitem__(dp, (c0.copy(), cells.copy())))
            st = al.getStatistics()
            out[mode] = dict(last_row=got["row"], rows=rows, pruned=int(st["pruned_cells"]), cells=int(st["cells"]),
                             restarts=runner.restarts, kernel=st["profile_kernel"])
            dist.barrier()
            al.close()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_chain_of_bands_prunes_a_global_alignment(pkg, oracle):
    """C5's shape in small: global NW, gap-initialised borders, four column bands in four processes on the one GPU
    (ports mapped with hipIpc; and once through the host), block pruning ON in every band -- the running lower bound of
    H[m][n] travels along the chain through the ports like the best score of a local alignment does.  H[m][n] of the
    last band is the oracle's, last row and special rows put together are lower bounds that are exact where a path can
    still reach it, every band but the first skips cells, nobody leaves the packed kernel."""
    from helpers import oracle_kwargs
    m, n, world = 72000, 80000, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_chain, args=(r, world, port, m, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=800) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=36)
    kw = oracle_kwargs(oracle, dict(start=4, end=4, pruning=False, disk=-1, block=(8192, 1 << 20)), m, n)
    kw.update(want_last_row=True, special_row_interval=8192)
    ref = oracle.stage1(s0, s1, **kw)
    final = ref["best"][2]
    want_rows = dict(zip(ref["special_row_ids"], ref["special_rows"]))
    for mode in ("plain", "pruned", "pruned_host"):
        assert all(res[r][mode]["restarts"] == 0 and res[r][mode]["kernel"] == 2 for r in range(world)), (mode, [(res[r][mode]["restarts"], res[r][mode]["kernel"]) for r in range(world)])
        row = np.concatenate([res[r][mode]["last_row"] for r in range(world)])
        assert int(row[-1, 0]) == final, mode
        if mode == "plain":
            assert np.array_equal(row, ref["last_row"][1:])
        else:
            assert np.all(row <= ref["last_row"][1:])
        for dp in sorted(want_rows):
            if dp >= m:
                continue
            got = np.concatenate([res[r][mode]["rows"][dp][1] for r in range(world)])
            want = want_rows[dp][1:]
            if mode == "plain":
                assert np.array_equal(got, want), dp
            else:
                assert np.all(got <= want), (mode, dp)
                must = _reach(want[:, 0].astype(np.int64), dp, np.arange(1, n + 1), m, n) >= final
                assert must.any() and np.array_equal(got[must, 0], want[must, 0]), (mode, dp)
    assert all(res[r]["plain"]["pruned"] == 0 for r in range(world))
    for mode in ("pruned", "pruned_host"):
        assert sum(res[r][mode]["pruned"] for r in range(world)) > 0.25 * m * n, mode
        assert all(res[r][mode]["pruned"] > 0 for r in range(world)), (mode, [res[r][mode]["pruned"] for r in range(world)])
    print("pruned fraction per band: p2p %s host %s" % (
        ["%.2f" % (res[r]["pruned"]["pruned"] / res[r]["pruned"]["cells"]) for r in range(world)],
        ["%.2f" % (res[r]["pruned_host"]["pruned"] / res[r]["pruned_host"]["cells"]) for r in range(world)]))


def test_local_pruning_far_above_the_16_bit_window(pkg):
    """400 000 x 300 000 related pair, scores up to 236 000: slabs whose entering scores are hundreds of thousands are
    skipped like any other (round 3 only skipped below 16 000 with the window at the floor).  Best cell = the int32
    kernels' (no pruning there); special rows, last row and last column are lower bounds with the row maximum intact
    above the best cell; more than 45 % of the matrix skipped; the packed kernel never hands over to the int32 ones."""
    m, n = 400000, 300000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=41)
    part = pkg.Partition(0, 0, m, n)
    res = {}
    for flags, prune in ((2, False), (0, True)):
        al = pkg.MI355Aligner(device=0, flags=flags, rows_per_lane=8)
        try:
            al.setSequences(s0, s1)
            mg = pkg.Stage1Manager(part, special_row_interval=49152, keep_last_row=True, keep_last_column=True, block_pruning=prune)
            al.alignPartition(part, mg)
            res[prune] = (mg, al.getStatistics())
        finally:
            al.close()
    (a, sa), (b, sb) = res[False], res[True]
    assert sa["profile_kernel"] != 2 and sb["profile_kernel"] == 2 and sb["kernel_launches"] == 1
    assert tuple(a.getBestScore()) == tuple(b.getBestScore()) and b.getBestScore()[2] > 200000
    assert sb["pruned_cells"] > 0.45 * m * n
    assert np.all(b.lastRow() <= a.lastRow()) and np.all(b.lastColumn() <= a.lastColumn())
    assert sorted(a.special_rows) == sorted(b.special_rows) and len(a.special_rows) >= 8
    for i in sorted(a.special_rows):
        x, y = a.specialRow(i), b.specialRow(i)
        assert np.all(y <= x) and np.all(y[1:, 0] >= 0), i
        if i <= a.getBestScore()[0]:
            assert y[:, 0].max() == x[:, 0].max() and int(y[:, 0].argmax()) == int(x[:, 0].argmax()), i


def test_diagonal_seed_gives_the_same_answers_and_prunes_more(pkg, monkeypatch):
    """9 M x 8.5 M related pair: a pruning run of a large matrix gets its first bound from a staircase of tiles along the
    diagonal (runtime.cpp::diagonal_seed; the reference seeds its bound with the best of a resumed run or of the other
    nodes, sw_stage1.cpp:210-217, AlignerPool).  The seed is the score of a real alignment, so nothing may change but the
    amount of work: same best cell (local) / same H[m][n] (global) as without the seed and as without pruning, special
    rows lower bounds of the unpruned ones with the row maxima intact above the best cell, more cells skipped."""
    m, n = 9000000, 8500000                      # (the seed only runs from 8 Mi x 8 Mi: below that it costs more than it saves)
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=37)
    part = pkg.Partition(0, 0, m, n)
    al = pkg.MI355Aligner(device=0)
    try:
        al.setSequences(s0, s1)
        for edge in (pkg.AT_ANYWHERE, pkg.AT_SEQUENCE_1_AND_2):
            res = {}
            for mode in ("plain", "pruned_no_seed", "pruned"):
                if mode == "pruned_no_seed":
                    monkeypatch.setenv("MI355SW_NO_DIAGONAL_SEED", "1")
                else:
                    monkeypatch.delenv("MI355SW_NO_DIAGONAL_SEED", raising=False)
                mg = pkg.Stage1Manager(part, alignment_start=edge, alignment_end=edge, special_row_interval=1 << 20, block_pruning=mode != "plain")
                al.alignPartition(part, mg)
                st = al.getStatistics()
                res[mode] = (mg, st)
                assert st["profile_kernel"] == 2 and st["restarts"] == 0, (mode, st)
            (a, sa), (b, sb), (c, sc) = res["plain"], res["pruned_no_seed"], res["pruned"]
            assert tuple(a.getBestScore()) == tuple(b.getBestScore()) == tuple(c.getBestScore())
            assert sa["seed_ms"] == 0 and sb["seed_ms"] == 0 and sc["seed_ms"] > 0
            assert sa["pruned_cells"] == 0 and sc["pruned_cells"] > sb["pruned_cells"] > 0.2 * m * n
            assert sc["pruned_cells"] > 0.6 * m * n                 # the seed finds the alignment: only what can tie it is left
            best_row = a.getBestScore()[0]
            assert sorted(a.special_rows) == sorted(c.special_rows) and len(a.special_rows) >= 4
            for i in sorted(a.special_rows):
                x, y = a.specialRow(i), c.specialRow(i)
                assert np.all(y <= x), i
                if edge == pkg.AT_ANYWHERE and i <= best_row:
                    assert y[:, 0].max() == x[:, 0].max() and int(y[:, 0].argmax()) == int(x[:, 0].argmax()), i
                if edge != pkg.AT_ANYWHERE:
                    must = _reach(x[:, 0].astype(np.int64), i, np.arange(0, n + 1), m, n) >= a.getBestScore()[2]
                    assert must.any() and np.array_equal(y[must, 0], x[must, 0]), i
            print("edge %d: skipped %.3f without the seed, %.3f with it (seed %.0f ms); kernel %.0f -> %.0f ms" % (
                edge, sb["pruned_cells"] / float(m) / n, sc["pruned_cells"] / float(m) / n, sc["seed_ms"], sb["kernel_ms"], sc["kernel_ms"]))
    finally:
        al.close()

"""GPU (-m gpu): the packed 16-bit kernel's safety net.  When its window cannot hold what it is given it says so
(MI355SW_EOVERFLOW16) and the work is redone on the int32 kernel; rows handed out before the report are exact and stay
handed out, the rerun REPLAYS (nothing is asked twice from the manager's sequential streams, nothing is dispatched
twice).  Real DP data never trips the net (scores move by at most 5 per cell and the window follows them), so the
tests use border data no DP matrix could produce, and a fault-injection knob for the band driver."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from test_bands_gloo import _free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _jump_cells(length, at, low, high, INF):
    c = np.zeros((length + 1, 2), dtype=np.int32)
    c[:, 0] = low
    c[at:, 0] = high
    c[:, 1] = -INF
    return c


@pytest.mark.parametrize("R", [0, 8])
def test_first_row_jump_is_recomputed_in_int32(pkg, oracle, R):
    """nothing progressive: the partition is simply run again (first row with a 0 -> 90 000 step in the middle)"""
    from masa_cudalign_amd.manager import ArrayCellsReader
    m, n = 5000, 9000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=301)
    row = _jump_cells(n, 5000, 0, 90000, pkg.INF)
    al = pkg.MI355Aligner(device=0, rows_per_lane=R)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        mg = pkg.Stage1Manager(part, keep_last_row=True, first_row_reader=ArrayCellsReader(row))
        al.alignPartition(part, mg)
        st = al.getStatistics()
        assert st["profile_kernel"] == 1            # the int32 kernel produced the result
        ref = oracle.stage1(s0, s1, first_row_type=oracle.INIT_WITH_CUSTOM_DATA, custom_first_row=row,
                            want_last_row=True)
        assert tuple(mg.getBestScore()) == tuple(ref["best"])
        assert np.array_equal(mg.lastRow(), ref["last_row"])
    finally:
        al.close()


@pytest.mark.parametrize("R,nw", [(0, False), (8, False), (4, True)])
def test_first_column_jump_replays_progressive_traffic(pkg, oracle, R, nw):
    """progressive traffic (streamed first column, last column and special rows handed over while the kernel runs):
    the strips above the step are clean and are dispatched by the packed pass; the strip that meets the step reports,
    the int32 pass replays.  The manager's first-column stream is read exactly once, every last-column row and every
    special row arrives exactly once and in order, and all of it equals the oracle's."""
    from masa_cudalign_amd.manager import ArrayCellsReader
    m, n, at = 45000, 3000, 30000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=302)
    INF = pkg.INF
    col = _jump_cells(m, at, 0, 100000, INF)
    row = np.zeros((n + 1, 2), dtype=np.int32); row[:, 1] = -INF

    class CountingReader(ArrayCellsReader):
        asked = 0

        def read(self, buf, length):
            self.asked += length
            return ArrayCellsReader.read(self, buf, length)

    class Mgr(pkg.Stage1Manager):
        col_calls = 0

        def dispatchColumn(self, j, buf, length):
            self.col_calls += 1
            pkg.Stage1Manager.dispatchColumn(self, j, buf, length)

    creader = CountingReader(col)
    al = pkg.MI355Aligner(device=0, rows_per_lane=R)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        kw = dict(alignment_start=pkg.AT_SEQUENCE_1_AND_2, alignment_end=pkg.AT_SEQUENCE_1_AND_2) if nw else {}
        mg = Mgr(part, keep_last_column=True, keep_last_row=True, special_row_interval=8192,
                 first_row_reader=ArrayCellsReader(row), first_column_reader=creader, **kw)
        al.alignPartition(part, mg)
        st = al.getStatistics()
    finally:
        al.close()
    assert st["profile_kernel"] == 1
    assert creader.asked == m + 1                    # corner + every row, once
    rec = oracle.NEEDLEMAN_WUNSCH if nw else oracle.SMITH_WATERMAN
    ref = oracle.stage1(s0, s1, recurrence=rec, first_row_type=oracle.INIT_WITH_CUSTOM_DATA, custom_first_row=row,
                        first_col_type=oracle.INIT_WITH_CUSTOM_DATA, custom_first_col=col,
                        best_mode=oracle.BEST_LAST_CELL if nw else oracle.BEST_ANYWHERE,
                        want_last_row=True, want_last_col=True, block_h=256, block_w=1 << 20, special_row_interval=256)
    got = mg.lastColumn()
    assert got.shape == ref["last_col"].shape and np.array_equal(got, ref["last_col"])
    assert np.array_equal(mg.lastRow(), ref["last_row"])
    assert tuple(mg.getBestScore()) == tuple(ref["best"])
    want = dict(zip(ref["special_row_ids"], ref["special_rows"]))
    rows = sorted(r for r in mg.special_rows if r < m)
    assert len(rows) >= 3 and rows == sorted(set(rows))
    assert any(r < at for r in rows) and any(r > at for r in rows)     # some came from each pass
    for r in rows:
        assert len(mg.special_rows[r]) == 2                              # leading cell + the row: ONE delivery
        assert np.array_equal(mg.specialRow(r), want[r]), r


def test_process_block_falls_back_to_int32(pkg, oracle, aligner):
    """AbstractBlockProcessor::processBlock with a border row no 16-bit window can hold"""
    s0, s1 = pkg.seqgen.related_pair(3000, 3000, cfg=303)
    aligner.setSequences(s0, s1)
    try:
        i0, j0, i1, j1 = 100, 200, 1700, 2900
        m, n = i1 - i0, j1 - j0
        rng = np.random.default_rng(3)
        row = np.stack([rng.integers(0, 50, n), rng.integers(-60, 40, n)], axis=1).astype(np.int32)
        row[n // 2:, 0] += 120000
        col = np.stack([rng.integers(0, 50, m + 1), rng.integers(-60, 40, m + 1)], axis=1).astype(np.int32)
        for rec in (pkg.SMITH_WATERMAN, pkg.NEEDLEMAN_WUNSCH):
            r1, c1 = row.copy(), col.copy()
            b1 = oracle.process_block(s0, s1, r1, c1, i0, j0, i1, j1, rec)
            r2, c2 = row.copy(), col.copy()
            b2 = aligner.processBlock(r2, c2, i0, j0, i1, j1, rec)
            assert aligner.getStatistics()["profile_kernel"] == 1
            assert np.array_equal(r1, r2) and np.array_equal(c1, c2)
            assert tuple(b1) == tuple(b2)
    finally:
        aligner.unsetSequences()


def test_streaming_form_reports_overflow_and_hands_out_only_exact_rows(pkg, oracle, monkeypatch):
    """the streaming ABI itself does not rerun: poll returns EOVERFLOW16, the rows it reported before are exact, rows
    of the failing strip are refused"""
    monkeypatch.setenv("MI355SW_FAULT_OVERFLOW_STRIP", "5")
    m, n = 40 * 256, 4000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=304)
    al = pkg.MI355Aligner(device=0, rows_per_lane=4)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        al.streamBegin(part, want_last_column=True)
        rows = 0
        with pytest.raises(pkg.AlignerError, match="EOVERFLOW16"):
            for _ in range(10000000):
                rows, fin = al.streamPoll()
        clean = rows                                 # what the last clean poll reported
        assert clean % 256 == 0 and clean <= 5 * 256
        got = al.streamReadColumn(0, clean) if clean else np.zeros((0, 2), dtype=np.int32)
        with pytest.raises(pkg.AlignerError):
            al.streamReadColumn(0, clean + 1)
        al.streamAbort()
        al.streamEnd()
        if clean:
            ref = oracle.stage1(s0[:clean], s1, want_last_col=True)
            assert np.array_equal(got, ref["last_col"][1:])
    finally:
        al.close()


def _band_worker(rank, world, port, m, n, transport, fault, q):
    sys.path.insert(0, ROOT)
    if fault is not None and rank == fault[0]:
        os.environ["MI355SW_FAULT_OVERFLOW_STRIP"] = str(fault[1])
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    from masa_cudalign_amd.bands import BandRunner, band_limits
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=43)
        lim = band_limits(n, [1] * world)
        al = pkg.MI355Aligner(device=0, rows_per_lane=4, waves=128)
        al.setSequences(s0, s1)
        runner = BandRunner(al, dist=dist, rank=rank, world=world, device=None, segment_rows=512, transport=transport)
        best = runner.run(m, lim[rank], lim[rank + 1])
        restarts = runner.restarts
        kernel = al.getStatistics()["profile_kernel"]
        gbest = runner.reduce_best(best)
        al.close()
        q.put((rank, tuple(gbest), restarts, kernel))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("transport", ["p2p", "host"])
@pytest.mark.parametrize("fault_rank", [0, 1, 2])
def test_band_chain_survives_an_overflow_report(pkg, oracle, transport, fault_rank):
    """three bands, three processes, one of them reports an overflow mid-run (fault injection: strip 7 of 24): that
    band restarts on the int32 kernel and replays, its neighbours never notice, the chain's answer is the oracle's"""
    m, n, world = 24 * 256, 7500, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_band_worker, args=(r, world, port, m, n, transport, (fault_rank, 7), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=800) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=43)
    ref = oracle.stage1(s0, s1)
    want = (ref["best"][0] - 1, ref["best"][1] - 1, ref["best"][2])
    for rank, gbest, restarts, kernel in res:
        assert gbest == want, (rank, gbest, want)
        assert restarts == (1 if rank == fault_rank else 0)
        assert kernel == (1 if rank == fault_rank else 2)

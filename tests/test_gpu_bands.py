"""GPU (-m gpu): ranks (gloo rendezvous, all on cuda:0 -- the box has one GPU) run column bands with the real HIP
engine while all their strip kernels are resident together.  Transport "p2p": the boundary column goes through a
column port -- band g's kernel stores into band g+1's HBM buffer (mapped across the processes with hipIpc, exactly
as between two GPUs over xGMI) and publishes the row count, band g+1's kernel polls it.  Transport "host": pinned
columns + gloo send/recv (the reference's socket chain)."""
import os
import sys

import pytest
import torch.multiprocessing as mp

from test_bands_gloo import _free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, m, n, q, transport="p2p"):
    sys.path.insert(0, ROOT)
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
        al = pkg.MI355Aligner(device=0, rows_per_lane=4)
        al.setSequences(s0, s1)
        runner = BandRunner(al, dist=dist, rank=rank, world=world, device=None, segment_rows=512, transport=transport)
        out = []
        for rep in range(2):                     # the second run re-uses the port (owner resets, then tells the writer)
            best = runner.run(m, lim[rank], lim[rank + 1])
            out.append((tuple(best), tuple(runner.reduce_best(best))))
        al.close()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("transport,world", [("p2p", 2), ("host", 2), ("p2p", 4)])
def test_bands_in_separate_processes_one_gpu(pkg, oracle, transport, world):
    m, n = 6000, 7000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, m, n, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=43)
    ref = oracle.stage1(s0, s1)
    want = (ref["best"][0] - 1, ref["best"][1] - 1, ref["best"][2])
    for rank, out in res:
        for best, gbest in out:
            assert gbest == want, (rank, best, gbest, want)


def _worker_nw(rank, world, port, m, n, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    from masa_cudalign_amd.bands import BandRunner, band_limits
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=45)
        lim = band_limits(n, [1] * world)
        al = pkg.MI355Aligner(device=0, rows_per_lane=4, waves=64)
        al.setSequences(s0, s1)
        runner = BandRunner(al, dist=dist, rank=rank, world=world, device=None, segment_rows=512, transport="p2p")
        got = {}
        runner.run(m, lim[rank], lim[rank + 1], recurrence=pkg.NEEDLEMAN_WUNSCH, track_best=False,
                   first_row_init_type=pkg.INIT_WITH_GAPS, first_col_init_type=pkg.INIT_WITH_GAPS,
                   want_last_row=True, before_end=lambda eng: got.update(row=eng.streamReadLastRow()))
        al.close()
        q.put((rank, got["row"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_global_nw_three_bands_three_processes_one_gpu(pkg, oracle):
    """C5's recurrence through the band driver with the real engine: global NW, gap-initialised borders, three
    bands (the middle one receives and sends while its kernel runs); the bands' last-row slices put together are
    the last row of the one-partition oracle run, ending on H[m][n]."""
    import numpy as np
    m, n, world = 5000, 6500, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_nw, args=(r, world, port, m, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=45)
    ref = oracle.stage1(s0, s1, recurrence=oracle.NEEDLEMAN_WUNSCH, first_row_type=oracle.INIT_WITH_GAPS,
                        first_col_type=oracle.INIT_WITH_GAPS, want_last_row=True, best_mode=oracle.BEST_LAST_CELL)
    row = np.concatenate([res[r] for r in range(world)])
    assert np.array_equal(row, ref["last_row"][1:])
    assert int(row[-1, 0]) == ref["best"][2]


def test_port_chain_in_one_process(pkg):
    """the reference's --split chain with every boundary column travelling through a column port inside one process
    (mi355sw_port_attach): band k's kernel writes band k+1's port, band k+1 reads it -- boundary columns and running
    bests match the fixture the reference produced.  Packed kernel and both int32 kernels."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import load_golden, make_pair, digest
    from masa_cudalign_amd.bands import band_limits, canonical_best
    ch = load_golden()["chain"]
    s0, s1 = make_pair(pkg, ch["seq"])
    n, parts, m = len(s1), ch["parts"], len(s0)
    lim = band_limits(n, [1] * parts)
    for flags in (0, 2, 3):
        als = [pkg.MI355Aligner(device=0, flags=flags) for _ in range(parts)]
        try:
            for k in range(parts):
                als[k].setSequences(s0, s1)
                if k > 0:
                    als[k].portCreate(m)
                    als[k - 1].portAttach(als[k])
            cands = []
            for k in range(parts):
                part = pkg.Partition(0, lim[k], m, lim[k + 1])
                kw = dict(last_column_port=k < parts - 1)
                if k > 0:
                    assert als[k].portRowsReady() == m                      # band k-1 published every row
                    col = np.concatenate([np.array([[0, -pkg.INF]], dtype=np.int32), als[k].portRead(0, m)])
                    assert digest(col) == ch["boundary_columns"]["STEP-%d.tmp" % k]
                    kw.update(first_column_init_type=pkg.INIT_WITH_CUSTOM_DATA, first_column_port=True, first_column=col[:1])
                als[k].streamBegin(part, **kw)
                while not als[k].streamPoll()[1]:
                    pass
                best, _ = als[k].streamEnd()
                cands.append(best)
                run = canonical_best(cands)
                assert [run[0] + 1, run[1] + 1, run[2]] == ch["band_bests"][k]
        finally:
            for al in als:
                al.close()

"""GPU (-m gpu): two ranks (gloo rendezvous, both on cuda:0 -- the box has one GPU) run two column bands
with the real HIP engine; the boundary column is streamed while both strip kernels are running."""
import os
import sys

import pytest
import torch.multiprocessing as mp

from test_bands_gloo import _free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, m, n, q):
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
        runner = BandRunner(al, dist=dist, rank=rank, world=world, device=None, segment_rows=512)
        best = runner.run(m, lim[rank], lim[rank + 1])
        gbest = runner.reduce_best(best)
        al.close()
        q.put((rank, tuple(best), tuple(gbest)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_bands_two_processes_one_gpu(pkg, oracle):
    m, n, world = 6000, 7000, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, m, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=43)
    ref = oracle.stage1(s0, s1)
    want = (ref["best"][0] - 1, ref["best"][1] - 1, ref["best"][2])
    for rank, best, gbest in res:
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
        runner = BandRunner(al, dist=dist, rank=rank, world=world, device=None, segment_rows=512)
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


def _worker_device_columns(q):
    """fresh process: torch's HIP runtime first (as in bench.py), then the engine"""
    import torch
    torch.cuda.init()
    import numpy as np
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as graft
    from helpers import load_golden, make_pair, digest
    pkg = graft.load_package()
    from masa_cudalign_amd.bands import band_limits, canonical_best
    ch = load_golden()["chain"]
    s0, s1 = make_pair(pkg, ch["seq"])
    n, parts, m = len(s1), ch["parts"], len(s0)
    lim = band_limits(n, [1] * parts)
    al = pkg.MI355Aligner(device=0)
    al.setSequences(s0, s1)
    dcol, corner, cands, ok = None, None, [], True
    for k in range(parts):
        part = pkg.Partition(0, lim[k], m, lim[k + 1])
        kw = dict(want_last_column=True)
        if dcol is not None:
            kw.update(first_column_init_type=pkg.INIT_WITH_CUSTOM_DATA, stream_first_column=True, first_column=corner)
        al.streamBegin(part, **kw)
        out = torch.empty((m, 2), dtype=torch.int32, device="cuda:0")
        fed = read = 0
        while True:
            if dcol is not None and fed < m:
                ln = min(1000, m - fed)
                al.streamFeedColumnDevice(fed, dcol[fed:fed + ln].data_ptr(), ln)
                fed += ln
            rows, fin = al.streamPoll()
            if rows > read:                      # drain what is complete while the kernel is still running
                al.streamReadColumnDevice(read, out[read:rows].data_ptr(), rows - read)
                read = rows
            if fin:
                break
        if read < m:
            al.streamReadColumnDevice(read, out[read:].data_ptr(), m - read)
        best, _ = al.streamEnd()
        newcol = np.concatenate([np.array([[0, -pkg.INF]], dtype=np.int32), out.cpu().numpy()])
        if k < parts - 1:
            ok = ok and digest(newcol) == ch["boundary_columns"]["STEP-%d.tmp" % (k + 1)]
        dcol, corner = out, newcol[:1]
        cands.append(best)
        run = canonical_best(cands)
        ok = ok and [run[0] + 1, run[1] + 1, run[2]] == ch["band_bests"][k]
    al.close()
    q.put(bool(ok))


@pytest.mark.timeout(600)
def test_chain_with_device_tensor_columns(pkg):
    """the reference's --split chain with the boundary column travelling as a DEVICE tensor (what an RCCL send/recv
    hands over: mi355sw_stream_read_column_device / mi355sw_stream_feed_column_device on the engine's copy stream
    while the strip kernel runs): boundary columns and running bests match the fixture"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_device_columns, args=(q,))
    p.start()
    assert q.get(timeout=500) is True
    p.join(timeout=60)
    assert p.exitcode == 0

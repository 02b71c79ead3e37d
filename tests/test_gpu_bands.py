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


def _worker_prune(rank, world, port, m, n, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    from masa_cudalign_amd.bands import BandRunner, band_limits
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=47)
        lim = band_limits(n, [1] * world)
        out = {}
        for mode, transport in (("plain", "p2p"), ("pruned", "p2p"), ("pruned_host", "host")):
            al = pkg.MI355Aligner(device=0, rows_per_lane=4)
            al.setSequences(s0, s1)
            runner = BandRunner(al, dist=dist, rank=rank, world=world, device=None, segment_rows=2048, transport=transport,
                                prune_blocks=(mode != "plain"))
            rows = {}
            best = runner.run(m, lim[rank], lim[rank + 1], special_row_interval=8192, n_total=n,
                              special_row_sink=lambda dp, c0, cells: rows.__setitem__(dp, (c0.copy(), cells.copy())))
            st = al.getStatistics()
            out[mode] = dict(best=tuple(runner.reduce_best(best)), rows=rows, pruned=int(st["pruned_cells"]),
                             cells=int(st["cells"]), hints=runner.hints, special=list(runner.special_rows),
                             restarts=runner.restarts, kernel=st["profile_kernel"])
            dist.barrier()
            al.close()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_chain_of_bands_prunes_with_the_shared_best_and_keeps_special_rows(pkg, oracle):
    """Four bands in four processes (one GPU, ports mapped with hipIpc), a related pair: block pruning ON in every
    band against the running best of the whole chain (the reference switches pruning off when it forks,
    M/libmasa/libmasa.cpp:1318-1321) and one special-rows slice per band (the reference: one area per forked node,
    Job.cpp:123-128).  Without pruning the concatenated slices are the single partition's special rows, cell for cell;
    with pruning the chain reports the same best cell, every band past the first skips cells, and the rows are lower
    bounds of the exact ones with the same maximum.  Both transports."""
    import numpy as np
    from masa_cudalign_amd.bands import band_limits
    m, n, world = 72000, 80000, 4       # scores up to ~60 000: the later bands' first columns lie far above 2^15
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_prune, args=(r, world, port, m, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=800) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=47)
    ref = oracle.stage1(s0, s1, special_row_interval=8192, block_h=1024, block_w=1024)     # one pass: best + every special row
    want = (ref["best"][0] - 1, ref["best"][1] - 1, ref["best"][2])
    lim = band_limits(n, [1] * world)
    dps = list(range(8192, m, 8192))
    assert ref["special_row_ids"] == dps
    for mode in ("plain", "pruned", "pruned_host"):
        assert all(res[r][mode]["best"] == want for r in range(world)), (mode, [res[r][mode]["best"] for r in range(world)], want)
        assert all(res[r][mode]["special"] == dps for r in range(world)), mode
    for k, dp in enumerate(dps):
        row = ref["special_rows"][k]                     # cell 0 = first-column cell with f = -INF
        got = np.concatenate([res[r]["plain"]["rows"][dp][1] for r in range(world)])
        assert np.array_equal(got, row[1:]), dp
        for r in range(world):
            assert tuple(int(x) for x in res[r]["plain"]["rows"][dp][0]) == (int(row[lim[r], 0]), -oracle.INF), (dp, r)
        for mode in ("pruned", "pruned_host"):
            got = np.concatenate([res[r][mode]["rows"][dp][1] for r in range(world)])
            assert np.all(got[:, 0] <= row[1:, 0]) and got[:, 0].max() == row[1:, 0].max(), (mode, dp)
    # every band stayed on the packed kernel: a boundary column deep inside a long alignment (scores far above the
    # 16-bit range, relative to a window that follows them) is not an overflow
    for mode in ("plain", "pruned", "pruned_host"):
        assert all(res[r][mode]["restarts"] == 0 and res[r][mode]["kernel"] == 2 for r in range(world)), (mode, [(res[r][mode]["restarts"], res[r][mode]["kernel"]) for r in range(world)])
    assert all(res[r]["plain"]["pruned"] == 0 for r in range(world))
    for mode in ("pruned", "pruned_host"):
        assert all(res[r][mode]["pruned"] > 0 for r in range(1, world)), (mode, [res[r][mode]["pruned"] for r in range(world)])
    print("pruned fraction per band: p2p %s host %s (hints %s)" % (
        ["%.2f" % (res[r]["pruned"]["pruned"] / res[r]["pruned"]["cells"]) for r in range(world)],
        ["%.2f" % (res[r]["pruned_host"]["pruned"] / res[r]["pruned_host"]["cells"]) for r in range(world)],
        [res[r]["pruned_host"]["hints"] for r in range(world)]))


def test_running_best_travels_through_the_ports_in_both_directions(pkg, oracle):
    """The two running-best words of a column port, deterministically: two bands in ONE process run one after the other.
    (1) band 1 starts when band 0 is through, so band 0's final best is waiting in band 1's port (pushed DOWN the
    chain): band 1 skips more cells than the same band pruning on its own (MI355SW_NO_SHARED_BEST).  (2) band 0 run
    again WITHOUT a port reset reads what band 1 published (UP the chain) -- the best of the same matrix, so a valid
    bound -- and skips more than it did the first time.  Best cells unchanged throughout."""
    from masa_cudalign_amd.bands import band_limits, canonical_best
    m, n = 30000, 32000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=48)
    ref = oracle.stage1(s0, s1)
    want = (ref["best"][0] - 1, ref["best"][1] - 1, ref["best"][2])
    lim = band_limits(n, [1, 1])

    def run_band(al, k, shared):
        if shared:
            os.environ.pop("MI355SW_NO_SHARED_BEST", None)
        else:
            os.environ["MI355SW_NO_SHARED_BEST"] = "1"
        try:
            kw = dict(prune_blocks=True, prune_rows=m, prune_cols=n - lim[k], share_best=True, last_column_port=(k == 0))
            if k == 1:
                corner = [[0, -pkg.INF]]
                kw.update(first_column_init_type=pkg.INIT_WITH_CUSTOM_DATA, first_column_port=True, first_column=corner)
            al.streamBegin(pkg.Partition(0, lim[k], m, lim[k + 1]), **kw)
            while not al.streamPoll()[1]:
                pass
            best, _ = al.streamEnd()
            assert al.getStatistics()["profile_kernel"] == 2          # the packed kernel took the band (no overflow report)
            return best, int(al.getStatistics()["pruned_cells"])
        finally:
            os.environ.pop("MI355SW_NO_SHARED_BEST", None)

    res = {}
    for shared in (False, True):
        a0, a1 = pkg.MI355Aligner(device=0, rows_per_lane=4), pkg.MI355Aligner(device=0, rows_per_lane=4)
        try:
            for al in (a0, a1):
                al.setSequences(s0, s1)
            a1.portCreate(m)
            a0.portAttach(a1)
            b0, p0 = run_band(a0, 0, shared)
            b1, p1 = run_band(a1, 1, shared)
            assert canonical_best([b0, b1]) == want
            b0b, p0b = run_band(a0, 0, shared)          # the port still holds what band 1 published
            # (a band that knows a better score from elsewhere records nothing below it: its own best may be empty now)
            assert canonical_best([b0b, b1]) == want
            if not shared:
                assert b0b == b0
            res[shared] = (p0, p1, p0b)
        finally:
            a0.close(); a1.close()
    print("pruned cells (band 0, band 1, band 0 again): alone %s shared %s" % (res[False], res[True]))
    assert res[True][1] > res[False][1]                  # down the chain
    assert res[True][2] > res[True][0]                   # up the chain


def _worker_area(rank, world, port, m, n, transport, work, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    from masa_cudalign_amd.bands import BandRunner, band_limits, band_stage1
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=11)
        lim = band_limits(n, [1] * world)
        al = pkg.MI355Aligner(device=0, rows_per_lane=4)          # 256-row strips: the block height of the reference run
        al.setSequences(s0, s1)
        runner = BandRunner(al, dist=dist, rank=rank, world=world, device=None, segment_rows=512, transport=transport)
        res = band_stage1(runner, m, lim[rank], lim[rank + 1], work, 500 * 1024, n_total=n)
        res["restarts"] = runner.restarts
        dist.barrier()
        al.close()
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("transport", ["p2p", "host"])
def test_band_stage1_leaves_the_area_of_a_forked_node(pkg, oracle, transport, tmp_path):
    """The ENGINE's bands against MASA-Core run as --split=3 --part=1..3 (oracle/_ref/ref_driver, prebuilt): each
    band's Special Rows Area -- rows with their boundary cell, last row, markers, the received boundary column --
    is the forked node's, byte for byte; through the column ports and through the host."""
    from test_bands_gloo import check_area_against_split_reference
    # (the engine keeps AbstractDiagonalAligner's minimum spacing of 8192 rows, the serial block aligner of the reference
    #  run has none: the budget is chosen so that the spacing is 33 strips of 256 rows for both)
    rows = check_area_against_split_reference(pkg, oracle, tmp_path, 40000, 13200, 3, _worker_area, transport, sra_limit="500K")
    assert rows >= 12


def _worker_wide(rank, world, port, m, n, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    from masa_cudalign_amd.bands import BandRunner, band_limits
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=73)
        lim = band_limits(n, [1] * world)
        out = {}
        for mode in ("plain", "pruned"):
            al = pkg.MI355Aligner(device=0, rows_per_lane=16)
            al.setSequences(s0, s1)
            runner = BandRunner(al, dist=dist, rank=rank, world=world, device=None, segment_rows=8192, transport="p2p",
                                prune_blocks=(mode == "pruned"))
            rows = {}
            best = runner.run(m, lim[rank], lim[rank + 1], special_row_interval=32768, n_total=n,
                              special_row_sink=lambda dp, c0, cells: rows.__setitem__(dp, (c0.copy(), cells.copy())))
            st = al.getStatistics()
            out[mode] = dict(best=tuple(runner.reduce_best(best)), rows=rows, pruned=int(st["pruned_cells"]), cells=int(st["cells"]),
                             restarts=runner.restarts, kernel=st["profile_kernel"])
            dist.barrier()
            al.close()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_wide_bands_with_pruning_against_the_single_partition(pkg):
    """Two bands of 120 000 columns (two processes, column port between them) -- wide enough for the packed kernel's
    hot chunk loop and for runs of pruned slabs taken 16 at a time, with a first column that arrives through the port --
    against ONE partition on the int32 kernels: same best cell; unpruned, the concatenated special-row slices are the single
    partition's rows cell for cell; pruned, lower bounds of them (H and F)."""
    import numpy as np
    m, n, world = 150000, 240000, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_wide, args=(r, world, port, m, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=800) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=73)
    al = pkg.MI355Aligner(device=0, rows_per_lane=16, flags=2)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        mg = pkg.Stage1Manager(part, special_row_interval=32768)
        al.alignPartition(part, mg)
        i, j, sc = mg.getBestScore()
        want = (i - 1, j - 1, sc)                         # the band runner reports the 0-based cell
        single = {dp: mg.specialRow(dp) for dp in sorted(mg.special_rows) if dp < m}
    finally:
        al.close()
    assert len(single) >= 3 and want[2] > 50000
    for mode in ("plain", "pruned"):
        assert all(res[r][mode]["best"] == want for r in range(world)), (mode, [res[r][mode]["best"] for r in range(world)], want)
        assert all(res[r][mode]["restarts"] == 0 and res[r][mode]["kernel"] == 2 for r in range(world)), mode
        for dp, row in single.items():
            got = np.concatenate([res[r][mode]["rows"][dp][1] for r in range(world)])
            if mode == "plain":
                assert np.array_equal(got, row[1:]), dp
            else:
                assert np.all(got <= row[1:]), dp
    assert all(res[r]["plain"]["pruned"] == 0 for r in range(world))
    assert sum(res[r]["pruned"]["pruned"] for r in range(world)) > 0.15 * m * n


# ---- bands on DISTINCT devices (round 4) -------------------------------------------------------------------------------
# Everything above runs its bands on cuda:0 -- the test boxes have one GPU -- so no boundary column ever crossed a GPU-to-GPU
# link in a test.  The tests below are the same chains with band k on device k; they skip themselves where
# mi355sw_device_count() is smaller than the chain (and are the first thing to run on a multi-GPU node).

def _devices(pkg):
    return pkg.engine.load_library().mi355sw_device_count()


def _worker_devices(rank, world, port, m, n, q, prune):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    from masa_cudalign_amd.bands import BandRunner, band_limits
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=48)
        lim = band_limits(n, [1] * world)
        out = {}
        for transport in ("p2p", "host"):
            al = pkg.MI355Aligner(device=rank)                       # band k on GPU k
            al.setSequences(s0, s1)
            runner = BandRunner(al, dist=dist, rank=rank, world=world, device=None, segment_rows=4096, transport=transport, prune_blocks=prune)
            best = runner.run(m, lim[rank], lim[rank + 1], n_total=n, digest_inbound=True)
            st = al.getStatistics()
            out[transport] = dict(best=tuple(runner.reduce_best(best)), crc=runner.inbound_crc, pruned=int(st["pruned_cells"]),
                                  restarts=runner.restarts, kernel=st["kernel"])
            dist.barrier()
            al.close()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("prune", [False, True])
def test_bands_on_distinct_devices_in_separate_processes(pkg, prune):
    """band k on GPU k, one process per band, ports mapped with hipIpc ACROSS devices (the path bench.py --gpus N takes):
    the chain's best cell is the single partition's, and every band receives through its port the very column it
    receives through the host (crc32)"""
    world = min(_devices(pkg), 4)
    if world < 2:
        pytest.skip("needs at least two GPUs (mi355sw_device_count() = %d)" % _devices(pkg))
    m, n = 200000, 80000 * world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_devices, args=(r, world, port, m, n, q, prune)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=800) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=48)
    al = pkg.MI355Aligner(device=0, flags=2)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        mg = pkg.Stage1Manager(part)
        al.alignPartition(part, mg)
        i, j, sc = mg.getBestScore()
    finally:
        al.close()
    for r in range(world):
        assert res[r]["p2p"]["best"] == res[r]["host"]["best"] == (i - 1, j - 1, sc), (r, res[r])
        assert res[r]["p2p"]["restarts"] == 0 and res[r]["host"]["restarts"] == 0
        if r > 0 and not prune:           # (with pruning the two runs may skip different slabs: lower bounds, not the same cells)
            assert res[r]["p2p"]["crc"] == res[r]["host"]["crc"], r
    if prune:
        assert sum(res[r]["p2p"]["pruned"] for r in range(world)) > 0


@pytest.mark.parametrize("recurrence", ["sw", "nw"])
def test_chain_on_distinct_devices_in_one_process(pkg, recurrence):
    """bands.InProcessChain: band k on GPU k, all driven by this process, ports attached with peer access
    (mi355sw_port_attach -> hipDeviceEnablePeerAccess; bench.py's "p2p-attach" transport).  Result = one band over all
    columns on GPU 0."""
    from masa_cudalign_amd.bands import InProcessChain, BandRunner, band_limits
    world = min(_devices(pkg), 4)
    if world < 2:
        pytest.skip("needs at least two GPUs (mi355sw_device_count() = %d)" % _devices(pkg))
    m, n = 300000, 100000 * world
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=49)
    als = [pkg.MI355Aligner(device=k) for k in range(world)]
    try:
        for a in als:
            a.setSequences(s0, s1)
        kw = dict(recurrence=pkg.NEEDLEMAN_WUNSCH, first_row_init_type=pkg.INIT_WITH_GAPS, first_col_init_type=pkg.INIT_WITH_GAPS) if recurrence == "nw" else {}
        for prune in (False, True):
            chain = InProcessChain(als, prune_blocks=prune)
            best, stats = chain.run(m, band_limits(n, [1] * world), **kw)
            assert chain.restarts == 0
            if recurrence == "sw":
                want = BandRunner(als[0]).run(m, 0, n)
            else:
                got = {}
                BandRunner(als[0]).run(m, 0, n, recurrence=pkg.NEEDLEMAN_WUNSCH, track_best=False, first_row_init_type=pkg.INIT_WITH_GAPS,
                                       first_col_init_type=pkg.INIT_WITH_GAPS, want_last_row=True,
                                       before_end=lambda eng: got.update(h=int(eng.streamReadLastRow(col=n - 1, length=1)[0, 0])))
                want = (m - 1, n - 1, got["h"])
            assert tuple(best) == tuple(want), (prune, best, want)
            if prune:
                assert sum(s["pruned_cells"] for s in stats) > 0
    finally:
        for a in als:
            a.close()


@pytest.mark.parametrize("recurrence", ["sw", "nw"])
def test_chain_in_one_process_on_one_device(pkg, recurrence):
    """the same InProcessChain with all four bands on cuda:0 (what a one-GPU box can run of it: the driver, the ports, the
    concurrent kernels -- 128 wavefronts each so that all four are resident): local and global, with and without pruning,
    against one band over all columns"""
    from masa_cudalign_amd.bands import InProcessChain, BandRunner, band_limits
    world, m, n = 4, 160000, 240000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=50)
    als = [pkg.MI355Aligner(device=0, waves=128, rows_per_lane=8) for _ in range(world)]
    ref = pkg.MI355Aligner(device=0)
    try:
        for a in als + [ref]:
            a.setSequences(s0, s1)
        kw = dict(recurrence=pkg.NEEDLEMAN_WUNSCH, first_row_init_type=pkg.INIT_WITH_GAPS, first_col_init_type=pkg.INIT_WITH_GAPS) if recurrence == "nw" else {}
        if recurrence == "sw":
            want = BandRunner(ref).run(m, 0, n)
        else:
            got = {}
            BandRunner(ref).run(m, 0, n, recurrence=pkg.NEEDLEMAN_WUNSCH, track_best=False, first_row_init_type=pkg.INIT_WITH_GAPS,
                                first_col_init_type=pkg.INIT_WITH_GAPS, want_last_row=True,
                                before_end=lambda eng: got.update(h=int(eng.streamReadLastRow(col=n - 1, length=1)[0, 0])))
            want = (m - 1, n - 1, got["h"])
        for prune in (False, True):
            chain = InProcessChain(als, prune_blocks=prune)
            for rep in range(2):                           # the second run re-uses (resets) the attached ports
                best, stats = chain.run(m, band_limits(n, [1] * world), **kw)
                assert tuple(best) == tuple(want), (prune, rep, best, want)
            assert chain.restarts == 0 and all(s["profile_kernel"] == 2 for s in stats)
            if prune:
                assert sum(s["pruned_cells"] for s in stats) > 0.2 * m * n
    finally:
        for a in als + [ref]:
            a.close()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("recurrence", ["sw", "nw"])
def test_chain_takes_its_first_bound_from_the_diagonal_seed(pkg, recurrence):
    """A chain that prunes starts from the diagonal seed of the WHOLE matrix (bands.chain_seed_bound -> mi355sw_seed_bound):
    three bands of a 9 M x 8.5 M related pair side by side on cuda:0, with and without it -- the same answer, the seed's value
    is the score of an alignment that exists (local) or a lower bound of H[m][n] (global), and far more of the matrix goes."""
    from masa_cudalign_amd.bands import InProcessChain, band_limits
    world, m, n = 3, 9000000, 8500000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=5)
    als = [pkg.MI355Aligner(device=0, waves=336) for _ in range(world)]
    try:
        for a in als:
            a.setSequences(s0, s1)
        kw = dict(recurrence=pkg.NEEDLEMAN_WUNSCH, first_row_init_type=pkg.INIT_WITH_GAPS, first_col_init_type=pkg.INIT_WITH_GAPS) if recurrence == "nw" else {}
        out = {}
        for seed in (False, True):
            chain = InProcessChain(als, prune_blocks=True, seed_bound=seed)
            best, stats = chain.run(m, band_limits(n, [1] * world), **kw)
            out[seed] = (tuple(best), sum(s["pruned_cells"] for s in stats) / float(m) / n, chain.initial_bound, chain.restarts)
        assert out[False][0] == out[True][0], out
        assert out[False][2] is None and out[True][2] is not None and out[False][3] == out[True][3] == 0
        if recurrence == "sw":
            # (the score of an alignment that exists: here the longest stretch the +-64 Ki band could follow, 82 % of the best)
            assert 0.5 * out[True][0][2] < out[True][2] <= out[True][0][2], out
        else:
            assert out[True][2] <= out[True][0][2] and out[True][2] > out[True][0][2] - 100000, out
        assert out[True][1] > out[False][1] + 0.15 and out[True][1] > 0.6, out
        # an unrelated pair has nothing to follow: no bound, and the chain runs as it did
        u0, u1 = pkg.seqgen.unrelated_pair(m, 8400000, cfg=5)
        if recurrence == "sw":
            als[0].setSequences(u0, u1)
            assert als[0].seedBound(pkg.Partition(0, 0, m, 8400000)) is None
    finally:
        for a in als:
            a.close()


def _worker_seeded(rank, world, port, m, n, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    from masa_cudalign_amd.bands import BandRunner, band_limits
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=5)
        lim = band_limits(n, [1] * world)
        al = pkg.MI355Aligner(device=0, waves=1024 // world)
        al.setSequences(s0, s1)
        runner = BandRunner(al, dist=dist, rank=rank, world=world, device=None, transport="p2p", prune_blocks=True)
        best = runner.run(m, lim[rank], lim[rank + 1], n_total=n)
        st = al.getStatistics()
        gbest = tuple(runner.reduce_best(best))
        al.close()
        q.put((rank, gbest, runner.initial_bound, st["pruned_cells"], runner.restarts))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_seed_bound_travels_with_the_start_token_between_processes(pkg):
    """two rank processes, one band each (ports through hipIpc on the shared GPU): band 0 runs the seed pass, band 1 receives
    the value with its start token -- both begin with the same bound, below the chain's best"""
    world, m, n = 2, 9000000, 8500000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_seeded, args=(r, world, port, m, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=800) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][2] is not None and res[0][2] == res[1][2], res
    assert res[0][1] == res[1][1] and 0.5 * res[0][1][2] < res[0][2] <= res[0][1][2], res
    assert sum(r[3] for r in res) > 0.6 * m * n and all(r[4] == 0 for r in res), res

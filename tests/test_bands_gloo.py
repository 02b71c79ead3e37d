"""CPU, world_size 2, 3 and 4 (gloo): the column-band driver (masa-cudalign_amd/bands.py) streams the boundary
column between ranks while both bands are running, and the reduced best equals the single-band answer.
The compute engine is a test double built on the oracle (the product engine needs a GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleStreamEngine:
    """Same streaming surface as MI355Aligner (streamBegin/Poll/FeedColumn/ReadColumn/End), computing
    row segments with the oracle as soon as their first-column rows have arrived."""

    def __init__(self, oracle, seq0, seq1, seg=256):
        self.o, self.s0, self.s1, self.seg = oracle, seq0, seq1, seg
        self.in_shm = self.out_shm = None
        self._opts = {"wait_seconds": 0.0}          # MI355Aligner.configure's switches (mi355sw_config)

    def configure(self, **kw):
        self._opts.update(kw)

    # -- column ports, stood in by POSIX shared memory between the rank processes: int32 row counter at +0, cells from
    #    +256 -- the layout of the engine's port (csrc/runtime.cpp); the "kernel" below publishes and polls like
    #    complete_strip_common / claim_strip_common do
    def portCreate(self, rows):
        from multiprocessing import shared_memory
        from masa_cudalign_amd.engine import PortHandle
        if self.in_shm is not None:
            self.in_shm.close(); self.in_shm.unlink()
        self.in_shm = shared_memory.SharedMemory(create=True, size=256 + 8 * (rows + 1))
        self.in_shm.buf[:256] = bytes(256)
        np.frombuffer(self.in_shm.buf, dtype=np.int32, count=64)[[16, 32]] = -999999999   # the two running-best words
        ph = PortHandle()
        name = self.in_shm.name.encode()
        for k, b in enumerate(name):
            ph.ipc[k] = b
        ph.rows, ph.bytes = rows, 256 + 8 * (rows + 1)
        return ph

    def portReset(self):
        self.in_shm.buf[:4] = bytes(4)
        np.frombuffer(self.in_shm.buf, dtype=np.int32, count=64)[[16, 32]] = -999999999

    def portOpen(self, ph):
        from multiprocessing import shared_memory
        name = bytes(ph.ipc).split(b"\0")[0].decode()
        if self.out_shm is not None:
            self.out_shm.close()
        self.out_shm = shared_memory.SharedMemory(name=name)

    def portClose(self):
        if self.out_shm is not None:
            self.out_shm.close(); self.out_shm = None
        if self.in_shm is not None:
            self.in_shm.close(); self.in_shm.unlink(); self.in_shm = None

    def _port_rows(self):
        return int(np.frombuffer(self.in_shm.buf, dtype=np.int32, count=1)[0])

    def portRead(self, row, length):
        return np.frombuffer(self.in_shm.buf, dtype=np.int32, count=2 * length, offset=256 + 8 * (row + 1)).reshape(-1, 2).copy()

    def streamAbort(self):
        pass

    def streamBegin(self, part, recurrence_type=1, track_best=True, first_row_init_type=0, first_row_start_offset=0,
                    want_last_column=False, first_column_init_type=0, stream_first_column=False, first_column=None,
                    first_column_port=False, last_column_port=False, special_row_interval=0, prune_blocks=False,
                    prune_rows=0, prune_cols=0, share_best=False, **kw):
        o = self.o
        self.from_port, self.to_port = first_column_port, last_column_port
        # special rows every K "strips" (the stand-in's strip = one row segment), never row 0 nor rows >= m
        self.sp_k = -(-special_row_interval // self.seg) if special_row_interval > 0 else 0
        self.sp_rows = {}
        # block pruning as the strip kernel does it: a block (row segment x 128 columns) is skipped when nothing that
        # enters it can reach the running best -- of this band, or (share_best) of the whole chain; skipped cells
        # count as H = 0, E = F = -INF
        self.prune = bool(prune_blocks)
        self.share = bool(share_best) and not os.environ.get("MI355SW_NO_SHARED_BEST")
        self.hint = -o.INF
        if kw.get("initial_bound") is not None:       # mi355sw_stream_params.initial_bound: what the bound starts from
            self.hint = int(kw["initial_bound"])
        self.initial_bound_seen = kw.get("initial_bound")
        self.own_best = -o.INF
        self.pruned_blocks = self.total_blocks = 0
        self.part, self.rec = part, recurrence_type
        self.m, self.n = part.i1 - part.i0, part.j1 - part.j0
        self.prune_rows = prune_rows or self.m
        self.prune_cols = prune_cols or self.n
        self.row = o.initial_cells(first_row_init_type, first_row_start_offset, self.n + 1)
        self.custom_col = first_column_init_type == o.INIT_WITH_CUSTOM_DATA
        self.col = np.zeros((self.m + 1, 2), dtype=np.int32)
        if self.custom_col:
            self.col[0] = first_column[0]
            self.fed = 0 if (stream_first_column or first_column_port) else self.m
        else:
            self.col[:] = o.initial_cells(first_column_init_type, 0, self.m + 1)
            self.fed = self.m
        self.row[0] = self.col[0]
        self.done = 0
        self.last_col = np.zeros((self.m, 2), dtype=np.int32)
        self.cands = []
        self.t_wait = None

    def streamFeedColumn(self, row, cells):
        assert row == self.fed
        self.col[1 + row:1 + row + len(cells)] = cells
        self.fed += len(cells)

    def _advance(self):
        o = self.o
        while self.done < self.m:
            r1 = min(self.done + self.seg, self.m)
            if self.from_port and self.fed < r1:
                ready = self._port_rows()               # what the neighbour's "kernel" has published so far
                if getattr(self, "port_fault", None) == "deaf":
                    ready = 0                            # a mapping that opens but never delivers
                if ready <= self.fed:                    # the kernel's wall-time wait budget (csrc/sw_kernel.h)
                    import time
                    from masa_cudalign_amd.engine import AlignerError
                    self.t_wait = self.t_wait or time.time()
                    if time.time() - self.t_wait > (self._opts["wait_seconds"] or 3600.0):    # mi355sw_config.wait_seconds
                        raise AlignerError("stream_poll: wait budget exhausted on the inbound column")
                else:
                    self.t_wait = None
                if ready > self.fed:
                    cells = np.frombuffer(self.in_shm.buf, dtype=np.int32, count=2 * (ready - self.fed),
                                          offset=256 + 8 * (self.fed + 1)).reshape(-1, 2)
                    self.col[1 + self.fed:1 + ready] = cells
                    if getattr(self, "port_fault", None) == "stale" and int(cells[:, 0].max()) >= 100:
                        k = self.fed + int(np.argmax(cells[:, 0])) // 8 * 8
                        self.col[1 + k:9 + k] = 0        # one 64-byte line seen before its stores had landed
                    self.fed = ready
            if self.fed < r1:
                return
            r0 = self.done
            if self.prune:
                self._segment_with_pruning(r0, r1)
                continue
            res = o.stage1(self.s0[self.part.i0 + r0:self.part.i0 + r1], self.s1[self.part.j0:self.part.j1],
                           recurrence=self.rec, first_row_type=o.INIT_WITH_CUSTOM_DATA, custom_first_row=self.row,
                           first_col_type=o.INIT_WITH_CUSTOM_DATA, custom_first_col=self.col[r0:r1 + 1],
                           want_last_row=True, want_last_col=True, block_h=64, block_w=128)
            self.row = res["last_row"]
            self.last_col[r0:r1] = res["last_col"][1:]
            if self.to_port:                             # cells first, then the row count (the release store)
                dst = np.frombuffer(self.out_shm.buf, dtype=np.int32, count=2 * (r1 - r0), offset=256 + 8 * (r0 + 1)).reshape(-1, 2)
                dst[:] = res["last_col"][1:]
                np.frombuffer(self.out_shm.buf, dtype=np.int32, count=1)[0] = r1
            b = res["best"]
            if b[0] >= 0:
                self.cands.append((b[0] - 1 + r0 + self.part.i0, b[1] - 1 + self.part.j0, b[2]))
                self.own_best = max(self.own_best, b[2])
            self._segment_done(r1)

    def _segment_done(self, r1):
        if getattr(self, "segment_delay", 0):            # (tests whose point is what arrives WHILE a band computes)
            import time
            time.sleep(self.segment_delay)
        if self.sp_k and (r1 // self.seg) % self.sp_k == 0 and r1 % self.seg == 0 and r1 < self.m:
            self.sp_rows[r1] = self.row[1:].copy()
        self._relay()
        self.done = r1

    def _known(self):
        return max(self.own_best, self.hint) if self.share else self.own_best

    def _relay(self):
        """relay_running_best of csrc/sw_kernel.h on the shared-memory ports: +64 pushed by the band on the left, +128
        published by the owner"""
        if not self.share:
            return
        word = lambda shm, off: np.frombuffer(shm.buf, dtype=np.int32, count=1, offset=off)
        v = max(self.own_best, self.hint)
        if self.from_port:
            v = max(v, int(word(self.in_shm, 64)[0]))
        if self.to_port:
            v = max(v, int(word(self.out_shm, 128)[0]))
        self.hint = max(self.hint, v)
        if self.from_port:
            word(self.in_shm, 128)[0] = v
        if self.to_port:
            word(self.out_shm, 64)[0] = v

    def _segment_with_pruning(self, r0, r1):
        o, INF = self.o, self.o.INF
        i0, i1 = self.part.i0 + r0, self.part.i0 + r1
        col = self.col[r0:r1 + 1].copy()
        for j0 in range(0, self.n, 128):
            j1 = min(j0 + 128, self.n)
            blk = self.row[1 + j0:1 + j1]
            self.total_blocks += 1
            e = max(int(blk[:, 0].max()), int(col[:, 0].max()))
            left = min(self.prune_rows - r0, self.prune_cols - j0)
            if self.rec == o.SMITH_WATERMAN and e + left < self._known():
                diag = int(blk[-1, 0])
                blk[:, 0], blk[:, 1] = 0, -INF
                col[0] = (diag, -INF)
                col[1:, 0], col[1:, 1] = 0, -INF
                self.pruned_blocks += 1
                continue
            b = o.process_block(self.s0, self.s1, blk, col, i0, self.part.j0 + j0, i1, self.part.j0 + j1, self.rec)
            if b[2] > -INF:
                self.cands.append(tuple(int(x) for x in b))
                self.own_best = max(self.own_best, int(b[2]))
        self.row[0] = (self.col[r1][0], -INF)
        self.last_col[r0:r1] = col[1:]
        if self.to_port:
            dst = np.frombuffer(self.out_shm.buf, dtype=np.int32, count=2 * (r1 - r0), offset=256 + 8 * (r0 + 1)).reshape(-1, 2)
            dst[:] = col[1:]
            np.frombuffer(self.out_shm.buf, dtype=np.int32, count=1)[0] = r1
        self._segment_done(r1)

    def streamReadSpecialRow(self, k, col=0, length=None):
        from masa_cudalign_amd.engine import AlignerError
        step = self.sp_k * self.seg
        if not self.sp_k or k < 0 or (k + 1) * step >= self.m:
            raise AlignerError("streamReadSpecialRow: EINVAL special row range")
        dp = (k + 1) * step
        length = self.n - col if length is None else length
        if length == 0:
            return dp, np.empty((0, 2), dtype=np.int32)
        return dp, self.sp_rows[dp][col:col + length].copy()

    def streamBestHint(self, score):
        self.hint = max(self.hint, int(score))

    def streamRunningBest(self):
        return max(self.own_best, self.hint) if self.share else self.own_best

    def streamPoll(self):
        self._advance()
        return self.done, self.done >= self.m

    def streamReadColumn(self, row, length):
        assert row + length <= self.done
        return self.last_col[row:row + length].copy()

    def streamReadLastRow(self, col=0, length=None):
        assert self.done >= self.m
        length = self.n - col if length is None else length
        return self.row[1 + col:1 + col + length].copy()

    def streamEnd(self):
        from masa_cudalign_amd.bands import canonical_best
        return canonical_best(self.cands), 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, m, n, q, transport="host"):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    pkg = graft.load_package()
    oracle = graft.load_oracle()
    from masa_cudalign_amd.bands import BandRunner, band_limits
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=41)
        lim = band_limits(n, [1] * world)
        eng = OracleStreamEngine(oracle, s0, s1, seg=200)
        runner = BandRunner(eng, dist=dist, rank=rank, world=world, device=None, segment_rows=300, transport=transport)
        if transport == "p2p":
            assert runner.probe_p2p(m)
        out = []
        for rep in range(2 if transport == "p2p" else 1):     # second run: port re-used (owner resets, then tells the writer)
            best = runner.run(m, lim[rank], lim[rank + 1])
            out.append((tuple(best), tuple(runner.reduce_best(best))))
        dist.barrier()
        eng.portClose()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,transport", [(2, "host"), (3, "host"), (4, "host"), (2, "p2p"), (4, "p2p")])
def test_bands_over_gloo(pkg, oracle, world, transport):
    """3 and 4 bands: middle bands receive and send at the same time.  "p2p": the driver's column-port protocol
    (probe, handle exchange before every run, reset by the owner, publish / poll of the row counter) with the ports
    stood in by shared memory between the rank processes."""
    m, n = 1500, 1800
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, m, n, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=41)
    ref = oracle.stage1(s0, s1)
    want = (ref["best"][0] - 1, ref["best"][1] - 1, ref["best"][2])   # 0-based cell, as the engine reports
    for rank, out in res:
        for best, gbest in out:
            assert gbest == want, (rank, best, gbest, want)


def _worker_fallback(rank, world, port, m, n, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    pkg = graft.load_package()
    oracle = graft.load_oracle()
    from masa_cudalign_amd.bands import BandRunner, band_limits
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=41)
        lim = band_limits(n, [1] * world)
        eng = OracleStreamEngine(oracle, s0, s1, seg=200)
        if rank == 1:                                  # this rank's GPU "cannot" export its port
            def broken(rows):
                raise OSError("hipIpcGetMemHandle: invalid argument")
            eng.portCreate = broken
        runner = BandRunner(eng, dist=dist, rank=rank, world=world, device=None, segment_rows=300, transport="p2p")
        mine = runner.probe_p2p(m)
        ok = torch.tensor([1 if mine else 0], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)      # what bench.py does: everybody or nobody
        if int(ok.item()) == 0:
            runner.transport = "host"
            eng.portClose()
        best = runner.run(m, lim[rank], lim[rank + 1])
        q.put((rank, mine, int(ok.item()), tuple(runner.reduce_best(best))))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_p2p_probe_failure_falls_back_to_host_for_everybody(pkg, oracle):
    """one rank cannot create its column port: its probe fails, its left neighbour gets an empty token and fails too,
    the third rank succeeds -- and after the all_reduce all three use the host transport and the chain's answer is right"""
    m, n, world = 1200, 1500, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fallback, args=(r, world, port, m, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=240) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=41)
    ref = oracle.stage1(s0, s1)
    want = (ref["best"][0] - 1, ref["best"][1] - 1, ref["best"][2])
    assert [res[r][0] for r in range(world)] == [False, False, True]
    assert all(res[r][1] == 0 and res[r][2] == want for r in range(world))


def _worker_verify(rank, world, port, m, n, fault, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    pkg = graft.load_package()
    oracle = graft.load_oracle()
    from masa_cudalign_amd.bands import BandRunner, band_limits
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=43)
        lim = band_limits(n, [1] * world)
        eng = OracleStreamEngine(oracle, s0, s1, seg=200)
        if rank == 1:
            eng.port_fault = fault
        runner = BandRunner(eng, dist=dist, rank=rank, world=world, device=None, segment_rows=300, transport="p2p",
                            prune_blocks=True)
        seen_prune = []                 # the check compares boundary columns cell by cell: it must run UNPRUNED (bands.verify_p2p)
        begin = eng.streamBegin

        def spy(*a, **kw):
            seen_prune.append(bool(kw.get("prune_blocks")))
            return begin(*a, **kw)
        eng.streamBegin = spy

        def all_min(v):
            t = torch.tensor([v], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item())
        ok = all_min(1 if runner.probe_p2p(m) else 0) == 1
        verified = ok and runner.verify_p2p(m // 3, lim[rank], lim[rank + 1], all_min, budget_s=3.0)
        assert seen_prune and not any(seen_prune) and runner.prune_blocks is True, (seen_prune, runner.prune_blocks)
        eng.streamBegin = begin
        # the check's time budget travels as configuration (engine.configure, runner.stall_abort_s), not through os.environ
        env_restored = ("MI355SW_WAIT_S" not in os.environ and "MI355SW_BAND_STALL_S" not in os.environ and
                        eng._opts["wait_seconds"] == 0.0 and runner.stall_abort_s is None)
        if not verified:
            runner.transport = "host"
            eng.portClose()
        best = runner.run(m, lim[rank], lim[rank + 1])
        q.put((rank, verified, runner.transport, env_restored, tuple(runner.reduce_best(best)), runner.p2p_error))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("fault", [None, "stale", "deaf"])
def test_transport_check_before_the_measurement(pkg, oracle, fault):
    """bench.py's N > 1 start-up: ports open everywhere, then a short chain runs once through the ports and once
    through the host and every band compares what it received and what it found.  Healthy ports: the check passes and
    the run uses them.  A port that delivers a line of cells before its stores landed ("stale"), or never delivers ("deaf":
    the band's wait budget runs out), fails the check on EVERY rank, within the budget, and the chain falls back to
    the host transport with the right answer."""
    m, n, world = 1500, 1400, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_verify, args=(r, world, port, m, n, fault, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=240) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=43)
    ref = oracle.stage1(s0, s1)
    want = (ref["best"][0] - 1, ref["best"][1] - 1, ref["best"][2])
    for r in range(world):
        verified, transport, env_restored, best, err = res[r]
        assert verified == (fault is None) and transport == ("p2p" if fault is None else "host"), (r, res[r])
        assert env_restored and best == want, (r, res[r])
    if fault is not None:
        assert any(res[r][4] for r in range(world))          # somebody says why


def _worker_nw(rank, world, port, m, n, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    pkg = graft.load_package()
    oracle = graft.load_oracle()
    from masa_cudalign_amd.bands import BandRunner, band_limits
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=44)
        lim = band_limits(n, [1] * world)
        eng = OracleStreamEngine(oracle, s0, s1, seg=128)
        runner = BandRunner(eng, dist=dist, rank=rank, world=world, device=None, segment_rows=256)
        runner.run(m, lim[rank], lim[rank + 1], recurrence=oracle.NEEDLEMAN_WUNSCH, track_best=False,
                   first_row_init_type=oracle.INIT_WITH_GAPS, first_col_init_type=oracle.INIT_WITH_GAPS)
        q.put((rank, eng.last_col.copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_global_nw_bands_over_gloo(pkg, oracle):
    """C5's recurrence through the band driver: global NW with gap-initialised borders over 3 bands -- every band's
    last column (corner cells, first-row offsets of the inner bands) equals that column of the one-partition oracle
    run, and the last band ends on H[m][n]."""
    m, n, world = 900, 1100, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_nw, args=(r, world, port, m, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from masa_cudalign_amd.bands import band_limits
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=44)
    lim = band_limits(n, [1] * world)
    kw = dict(recurrence=oracle.NEEDLEMAN_WUNSCH, first_row_type=oracle.INIT_WITH_GAPS,
              first_col_type=oracle.INIT_WITH_GAPS, best_mode=oracle.BEST_LAST_CELL)
    for rank in range(world):
        ref = oracle.stage1(s0, s1[:lim[rank + 1]], want_last_col=True, **kw)
        assert np.array_equal(res[rank], ref["last_col"][1:]), rank
    full = oracle.stage1(s0, s1, **kw)
    assert int(res[world - 1][-1, 0]) == full["best"][2]


def _worker_prune(rank, world, port, m, n, transport, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    pkg = graft.load_package()
    oracle = graft.load_oracle()
    from masa_cudalign_amd.bands import BandRunner, band_limits
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=41)
        lim = band_limits(n, [1] * world)
        out = {}
        for mode in ("plain", "pruned", "pruned_alone"):
            # "pruned_alone": every band prunes against its own best only (what a chain without sharing could do)
            if mode == "pruned_alone":
                os.environ["MI355SW_NO_SHARED_BEST"] = "1"
            eng = OracleStreamEngine(oracle, s0, s1, seg=256)   # strip heights are multiples of 256, as the engine's
            # the stand-in sweeps a band in a millisecond or two, the host transport's side thread exchanges the chain's best
            # every 2 ms: whether a hint arrives before a band is through would be the scheduler's choice (seen once in a
            # full-suite run on a loaded machine).  A real band takes seconds; 10 ms per row segment keeps the order of events.
            eng.segment_delay = 0.01
            runner = BandRunner(eng, dist=dist, rank=rank, world=world, device=None, segment_rows=300, transport=transport,
                                prune_blocks=(mode != "plain"))
            if transport == "p2p":
                assert runner.probe_p2p(m)
            rows = {}
            best = runner.run(m, lim[rank], lim[rank + 1], special_row_interval=512, n_total=n,
                              special_row_sink=lambda dp, c0, cells: rows.__setitem__(dp, (c0.copy(), cells.copy())))
            out[mode] = dict(best=tuple(runner.reduce_best(best)), rows=rows, pruned=eng.pruned_blocks, total=eng.total_blocks,
                             hints=runner.hints, special=list(runner.special_rows))
            dist.barrier()
            eng.portClose()
            os.environ.pop("MI355SW_NO_SHARED_BEST", None)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("transport", ["host", "p2p"])
def test_bands_prune_with_the_chain_wide_best_and_keep_special_rows(pkg, oracle, transport):
    """Block pruning over a chain of bands (the reference switches it off when it forks, libmasa.cpp:1318-1321) and one
    special-rows slice per band (the reference: one area per forked node).  Unpruned: the concatenated slices ARE the
    single partition's special rows.  Pruned: same best cell, every band past the first prunes, the rows are lower
    bounds with the same maximum -- and sharing the best along the chain prunes more than every band on its own."""
    m, n, world = 1500, 1800, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_prune, args=(r, world, port, m, n, transport, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=41)
    ref = oracle.stage1(s0, s1)
    want = (ref["best"][0] - 1, ref["best"][1] - 1, ref["best"][2])
    for mode in ("plain", "pruned", "pruned_alone"):
        assert all(res[r][mode]["best"] == want for r in range(world)), mode
        assert all(res[r][mode]["special"] == [512, 1024] for r in range(world)), mode
    for dp in (512, 1024):
        row = oracle.stage1(s0[:dp], s1, want_last_row=True)["last_row"]      # cell 0 = first-column cell, f = -INF
        for mode in ("plain", "pruned"):
            got = np.concatenate([res[r][mode]["rows"][dp][1] for r in range(world)])
            if mode == "plain":
                assert np.array_equal(got, row[1:]), dp
                from masa_cudalign_amd.bands import band_limits
                lim = band_limits(n, [1] * world)
                for r in range(world):       # leading cell of each slice = the boundary column's cell of that row
                    assert tuple(res[r][mode]["rows"][dp][0]) == (int(row[lim[r], 0]), -oracle.INF), (dp, r)
            else:
                assert np.all(got[:, 0] <= row[1:, 0]) and got[:, 0].max() == row[1:, 0].max(), dp
    assert all(res[r]["plain"]["pruned"] == 0 for r in range(world))
    assert all(res[r]["pruned"]["pruned"] > 0 for r in range(1, world))
    shared = sum(res[r]["pruned"]["pruned"] for r in range(world))
    alone = sum(res[r]["pruned_alone"]["pruned"] for r in range(world))
    assert shared > alone, (shared, alone)
    if transport == "host":                      # the side thread delivered somebody else's best at least once
        assert sum(res[r]["pruned"]["hints"] for r in range(world)) > 0


def _worker_area(rank, world, port, m, n, transport, work, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    pkg = graft.load_package()
    oracle = graft.load_oracle()
    from masa_cudalign_amd.bands import BandRunner, band_limits, band_stage1
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=11)
        lim = band_limits(n, [1] * world)
        eng = OracleStreamEngine(oracle, s0, s1, seg=256)
        runner = BandRunner(eng, dist=dist, rank=rank, world=world, device=None, segment_rows=300, transport=transport)
        if transport == "p2p":
            assert runner.probe_p2p(m)
        res = band_stage1(runner, m, lim[rank], lim[rank + 1], work, 100 * 1024, n_total=n)
        dist.barrier()
        eng.portClose()
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def check_area_against_split_reference(pkg, oracle, tmp_path, m, n, world, worker, transport, sra_limit="100K"):
    """runs MASA-Core as --split=world --part=1..world (oracle/_ref/ref_driver: the reference's own stage 1, area and
    split code around a serial block aligner of 256-row blocks) and `worker` (band_stage1 on every rank) on the same
    pair, and compares the areas file by file"""
    from oracle import binding as ob
    if not ob.have_ref():
        pytest.skip("oracle/_ref/ref_driver not built")
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=11)
    refdir = str(tmp_path / "ref")
    os.makedirs(refdir)
    for k in range(1, world + 1):          # one work directory for the chain: a part waits for its left neighbour's files there
        args = ["--disk-size=" + sra_limit, "--no-block-pruning", "--block=256,128", "--split=%d" % world, "--part=%d" % k]
        ob.run_ref(s0, s1, args, workdir=refdir, timeout=120)
    area = os.path.join(refdir, "work", "special_rows", "stage.01.00")
    lim = [(n * g) // world for g in range(world + 1)]
    ref = []
    for r in range(world):
        d = "%08X.%08X.%08X.%08X" % (0, lim[r], m, lim[r + 1])
        ref.append({d: {fn: open(os.path.join(area, d, fn), "rb").read() for fn in sorted(os.listdir(os.path.join(area, d)))}})
    assert len(os.listdir(area)) == world
    work = str(tmp_path / "native")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, m, n, transport, work, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = oracle.stage1(s0, s1)["best"] if m * n <= 20_000_000 else None
    rows = 0
    for r in range(world):
        if want is not None:
            assert res[r]["best"] == tuple(want)
        assert res[r]["best"] == res[0]["best"]
        area = os.path.join(work, "FORK.%02d" % r, "special_rows", "stage.01.00")
        got = {d: {fn: open(os.path.join(area, d, fn), "rb").read() for fn in sorted(os.listdir(os.path.join(area, d)))}
               for d in sorted(os.listdir(area))}
        assert sorted(got) == sorted(ref[r]), (r, sorted(got), sorted(ref[r]))
        for d in got:
            # (the partition's last row is always kept natively -- the completion marker stage1.py resumes by; MASA-Core
            #  has it when the last block row happens to be a special one, and then it is the same file)
            last = "%08X" % m
            if last not in ref[r][d]:
                assert len(got[d].pop(last)) == 8 * (lim[r + 1] - lim[r] + 1)
            assert sorted(got[d]) == sorted(ref[r][d]), (r, d, sorted(got[d]), sorted(ref[r][d]))
            for fn in got[d]:
                assert got[d][fn] == ref[r][d][fn], (r, d, fn)
            rows += len([fn for fn in got[d] if len(fn) == 8])
        cp = open(os.path.join(work, "FORK.%02d" % r, "crosspoints", "crosspoint_01.00")).read().split()
        assert tuple(int(x) for x in cp[1].split(",")) == (0,) + tuple(res[r]["best"])
    # the last node's crosspoint file is the chain's in the reference too
    ref_cp = open(os.path.join(refdir, "work", "crosspoints", "crosspoint_01.00")).read().split()
    assert tuple(int(x) for x in ref_cp[1].split(","))[1:] == tuple(res[0]["best"])
    return rows


@pytest.mark.timeout(300)
@pytest.mark.parametrize("transport", ["host", "p2p"])
def test_band_stage1_leaves_the_area_of_a_forked_node(pkg, oracle, transport, tmp_path):
    """band_stage1 against MASA-Core itself run as --split=3 --part=1..3: per band the same partition directory, the
    same special-row files, border markers and tee'd boundary column, byte for byte."""
    rows = check_area_against_split_reference(pkg, oracle, tmp_path, 3000, 3300, 3, _worker_area, transport)
    assert rows >= 9


class SeedingOracleEngine(OracleStreamEngine):
    """... with mi355sw_seed_bound: here simply the best local score itself (the score of an alignment that exists)"""
    seed_calls = 0

    def seedBound(self, part, recurrence_type=1):
        self.seed_calls += 1
        ref = self.o.stage1(self.s0[part.i0:part.i1], self.s1[part.j0:part.j1])
        return int(ref["best"][2])


def _worker_seed(rank, world, port, m, n, transport, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    pkg = graft.load_package()
    oracle = graft.load_oracle()
    from masa_cudalign_amd import bands
    bands.SEED_MIN_EXTENT = 0                  # the test matrix is tiny
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=41)
        lim = bands.band_limits(n, [1] * world)
        out = {}
        for seed in (False, True):
            eng = SeedingOracleEngine(oracle, s0, s1, seg=256)
            runner = bands.BandRunner(eng, dist=dist, rank=rank, world=world, device=None, segment_rows=300, transport=transport,
                                      prune_blocks=True, seed_bound=seed)
            if transport == "p2p":
                assert runner.probe_p2p(m)
            best = runner.run(m, lim[rank], lim[rank + 1], n_total=n)
            out[seed] = dict(best=tuple(runner.reduce_best(best)), pruned=eng.pruned_blocks, total=eng.total_blocks, calls=eng.seed_calls,
                             bound=runner.initial_bound, seen=eng.initial_bound_seen)
            dist.barrier()
            eng.portClose()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("transport", ["host", "p2p"])
def test_seed_bound_travels_with_the_start_token(pkg, oracle, transport):
    """A pruning chain starts from the diagonal seed of the whole matrix (bands.chain_seed_bound): band 0 alone asks its engine
    for it, the value reaches every band with the ordered-start token and goes into its streamBegin(initial_bound=...); same
    best cell, and every band -- the first one above all, which otherwise starts from nothing -- prunes at least as much."""
    m, n, world = 1500, 1800, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_seed, args=(r, world, port, m, n, transport, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=41)
    ref = oracle.stage1(s0, s1)
    want = (ref["best"][0] - 1, ref["best"][1] - 1, ref["best"][2])
    for r in range(world):
        assert res[r][False]["best"] == res[r][True]["best"] == want, r
        assert res[r][False]["bound"] is None and res[r][False]["seen"] is None and res[r][False]["calls"] == 0, r
        assert res[r][True]["bound"] == res[r][True]["seen"] == want[2], r
        assert res[r][True]["calls"] == (1 if r == 0 else 0), r
        assert res[r][True]["pruned"] >= res[r][False]["pruned"], r
    assert res[0][True]["pruned"] > res[0][False]["pruned"]

"""Column-band (multi-GPU) Stage-1 driver: one process per GPU, band g owns columns
[n*W_g/sum(W), n*W_{g+1}/sum(W)) of seq1 and every row of seq0; band g streams its LAST column
(H,E per row) to band g+1, which consumes it as its FIRST column (INIT_WITH_CUSTOM_DATA).

Replaces the reference's --fork/--split chain (M/libmasa/libmasa.cpp:497-642): forked processes
joined by TCP sockets carrying cell_t (M/common/io/SocketCells{Reader,Writer}.cpp), best score
relayed through AlignerPool signal files (M/stage1/sw_stage1.cpp:421-464).  Here the boundary
column goes GPU to GPU: band g+1 owns a column PORT in its own HBM, band g's strip kernel stores its
last column straight into it over xGMI (peer-mapped through hipIpc) and publishes the row count with a
system-scope release store that band g+1's kernel polls -- no host, copy queue or PCIe in the loop
(transport "p2p").  The CPU tests' stand-in engine uses the "host" transport instead (row segments sent
with torch.distributed send/recv, gloo).  The best score is an all_gather of one (score,i,j) triple
reduced with BestScoreList's order.

The compute engine is injected (`engine_factory`) so that the N>1 plumbing can be exercised on CPU
with a stand-in; the product engine is MI355Aligner (HIP, no CPU fallback).
"""
import os
import sys
import threading
import time

import numpy as np

from .engine import INF, SMITH_WATERMAN, INIT_WITH_ZEROES, INIT_WITH_CUSTOM_DATA, Partition


def band_limits(n, weights):
    """split_sequences, libmasa.cpp:497-535: band g = (n*P[g]/sum, n*P[g+1]/sum] in 1-based trim
    coordinates, i.e. 0-based half-open [n*P[g]//sum, n*P[g+1]//sum)."""
    tot = int(sum(weights))
    acc = [0]
    for w in weights:
        acc.append(acc[-1] + int(w))
    return [(n * acc[g]) // tot for g in range(len(weights) + 1)]


def rows_per_lane_for_bands(m, band_cols, world, waves=1024):
    """Strip height for a chain of `world` equal column bands.  Band g+1 can only start once band g has
    swept its first strips across the whole band, and can never finish earlier than one band sweep after
    band g does, so the chain costs (rounds + world - 1) sweeps of band_cols steps; the step time grows
    with the strip height (measured table, same as runtime.cpp step_ns) while taller strips need fewer
    rounds.  Equal widths are optimal for such a chain (a narrower later band is throttled by its
    predecessor's rate, a wider one by its own), so only the height is planned."""
    best, tb = 0, None
    for R in (4, 8, 12, 16, 24, 32):
        step = {4: 97, 8: 138, 12: 181, 16: 222, 24: 291, 32: 368}[R]
        strips = -(-m // (64 * R))
        rounds = -(-strips // waves)
        t = step * ((rounds + world - 1) * band_cols + 280.0 * min(strips, waves))
        if tb is None or t < tb:
            best, tb = R, t
    return best


def canonical_best(cands):
    """BestScoreList order (M/common/BestScoreList.hpp:30-38): score desc, i asc, j asc."""
    best = (-1, -1, -INF)
    for (i, j, s) in cands:
        if j < 0 and i < 0:
            continue
        if s > best[2] or (s == best[2] and (i < best[0] or (i == best[0] and j < best[1]))):
            best = (int(i), int(j), int(s))
    return best


SEED_MIN_EXTENT = 8 << 20            # rows and columns from which the diagonal seed pays (as in mi355sw_stream_begin)
NO_BOUND = -(1 << 40)                # "no bound" in the start token that travels along the chain


def chain_seed_bound(engine, m, n, recurrence, first_row_init_type, first_col_init_type):
    """What the pruning bound of every band of a chain starts from: the diagonal seed pass over the WHOLE m x n matrix
    (mi355sw_seed_bound), run once on `engine`'s GPU before the first band starts.  A single partition gets it from
    mi355sw_stream_begin by itself; a band cannot (it sees its own columns, and its borders are ports), and without it
    each band prunes only against what the chain has found so far -- at 62 M x 57 M, 41 % of the cells instead of 75 %.
    None where the pass does not apply: small matrices, borders other than the recurrence's own (zeroes for a local
    alignment, gap penalties from the origin for a global one), an engine without the entry point, MI355SW_NO_DIAGONAL_SEED."""
    if m < SEED_MIN_EXTENT or n < SEED_MIN_EXTENT or not hasattr(engine, "seedBound") or os.environ.get("MI355SW_NO_DIAGONAL_SEED"):
        return None
    own = INIT_WITH_ZEROES if recurrence == SMITH_WATERMAN else 1          # 1 = INIT_WITH_GAPS
    if first_row_init_type != own or first_col_init_type != own:
        return None
    return engine.seedBound(Partition(0, 0, m, n), recurrence)


def check_chain_bound(score, bound):
    """A pruning chain must end at or above the bound it started from (the diagonal seed's, a caller's): that bound is the score of
    an alignment that exists / a lower bound of H[m][n], and slabs that can still reach it are computed.  A chain that ends below
    was given a bound no alignment reaches -- its optimum may have been pruned away, the result is void.  One stream checks this
    itself (MI355SW_EBOUND, mi355sw_stream_end); a band of a chain cannot (the best may be another band's)."""
    from .engine import AlignerError
    if bound is not None and score is not None and int(score) < int(bound):
        raise AlignerError("EBOUND: the chain ended at %d, below the bound %d its pruning started from -- no alignment reaches that "
                           "bound, the result is void" % (int(score), int(bound)))


class BandRunner:
    """Runs one band of the chain on this rank.  `dist` is torch.distributed (already initialised) or an object
    with send/recv/all_gather, or None for a single band.  All ranks must call run() with the same m.

    transport = "p2p"  : the boundary column goes GPU to GPU (column ports, include/mi355sw.h): band g+1 owns a
                         port in its own HBM, band g's strip kernel stores its last column straight into it over
                         xGMI and publishes the row count with a system-scope release store; the hosts only
                         exchange the 80-byte port handle before the kernels start.  Default on GPUs.
                "host" : pinned zero-copy columns + send/recv of row segments between the rank processes (the
                         reference's socket chain, libmasa.cpp:540-642); used by the CPU tests' stand-in engine
                         and as a fallback.
    `device` is only the device of the tensors used for collectives (reduce_best)."""

    def __init__(self, engine, dist=None, rank=0, world=1, device=None, segment_rows=1 << 16, prune_blocks=False,
                 transport="host", seed_bound=True):
        self.engine, self.dist, self.rank, self.world = engine, dist, rank, world
        self.prune_blocks = prune_blocks
        self.seed_bound = seed_bound   # pruning chains: band 0 runs the diagonal seed of the whole matrix first (chain_seed_bound)
        self.initial_bound = None      # ... and this is what the last run's bound started from
        self.device = device
        self.segment_rows = segment_rows
        self.transport = transport if world > 1 else "none"
        self._in_port = None         # PortHandle of this band's inbound port, and the rows it was made for
        self._in_rows = 0
        self._out_token = None       # handle bytes of the neighbour's port currently mapped
        self.restarts = 0            # packed-kernel overflow restarts (int32 rerun) of the last run
        self.p2p_error = None        # why probe_p2p() / verify_p2p() failed on this rank
        self.inbound_crc = None      # crc32 of the inbound boundary column of the last run(digest_inbound=True)
        self.special_rows = []       # DP rows of the special rows the last run handed to its sink
        self.hints = 0               # best-score hints this band took from the others in the last run (host transport)
        self.stall_abort_s = None    # seconds without progress after which run() gives up (None: MI355SW_BAND_STALL_S, default 900)
        # Block pruning over a chain of bands needs the best of the WHOLE matrix (the reference switches pruning off
        # when it forks, libmasa.cpp:1318-1321).  Between GPUs the kernels share it through the column ports; on the
        # host transport a side thread does, with one small all_reduce(MAX) every few milliseconds on its own group.
        self._best_group = None
        if prune_blocks and world > 1 and dist is not None and hasattr(dist, "new_group"):
            self._best_group = dist.new_group(list(range(world)), backend="gloo")

    def _tensor(self, rows):
        import torch
        return torch.empty((rows, 2), dtype=torch.int32, device="cpu")

    # -- p2p: port hand-shake, once per run ------------------------------------------------------------
    def _exchange_ports(self, m):
        """band g (> 0) makes its inbound port ready (created, or its counter reset) and THEN tells band g-1, which
        only then starts its kernel: nothing can be published into a port before its owner has reset it."""
        import torch
        from .engine import PortHandle
        eng, dist = self.engine, self.dist
        first, last = self.rank == 0, self.rank == self.world - 1
        if not first:
            if self._in_port is None or self._in_rows < m:     # a port made for a taller matrix serves a shorter one
                self._in_port, self._in_rows = eng.portCreate(m), m
            else:
                eng.portReset()
            tok = torch.frombuffer(bytearray(self._in_port.tobytes()), dtype=torch.uint8).clone()
            dist.send(tok, dst=self.rank - 1)
        if not last:
            tok = torch.empty(len(PortHandle().tobytes()), dtype=torch.uint8)
            dist.recv(tok, src=self.rank + 1)
            b = bytes(tok.numpy().tobytes())
            if b != self._out_token:
                eng.portOpen(PortHandle.frombytes(b))
                self._out_token = b

    def probe_p2p(self, m):
        """One port hand-shake without a run, errors caught: True if this rank could create its inbound port and map
        its neighbour's.  The caller makes the ranks agree (all_reduce MIN) and falls back to transport="host" for
        everybody if any of them could not -- a rank that finds out in the middle of run() would leave its
        neighbours' kernels waiting."""
        if self.transport != "p2p":
            return True
        import torch
        from .engine import PortHandle
        ok = True
        first, last = self.rank == 0, self.rank == self.world - 1
        size = len(PortHandle().tobytes())
        try:
            if not first:
                try:
                    self._in_port, self._in_rows = self.engine.portCreate(m), m
                    tok = torch.frombuffer(bytearray(self._in_port.tobytes()), dtype=torch.uint8).clone()
                except Exception as e:            # whatever it was: the neighbour must still get its (empty) token
                    self.p2p_error = str(e)
                    ok = False
                    tok = torch.zeros(size, dtype=torch.uint8)
                self.dist.send(tok, dst=self.rank - 1)
            if not last:
                tok = torch.empty(size, dtype=torch.uint8)
                self.dist.recv(tok, src=self.rank + 1)
                b = bytes(tok.numpy().tobytes())
                if not any(b):
                    ok = False
                else:
                    try:
                        self.engine.portOpen(PortHandle.frombytes(b))
                        self._out_token = b
                    except Exception as e:
                        self.p2p_error = str(e)
                        ok = False
        except Exception as e:            # transport of the tokens itself failed
            self.p2p_error = str(e)
            ok = False
        return ok

    def verify_p2p(self, m, j0, j1, all_min, budget_s=60.0):
        """Functional check of the port transport before anything is timed: the same short chain (m rows) once
        through the ports and once through the host.  True only if, on EVERY rank, the port run came through within
        `budget_s` seconds of waiting and gave the band the same inbound column (crc32) and the same best cell as the
        host run -- a peer mapping that opens but does not deliver (a kernel that never sees the neighbour's counter
        move, or sees it move before the cells) shows up here, bounded in time, instead of in the measurement.
        `all_min(int) -> int` is a collective MIN over the ranks (every rank calls it twice).  On False the caller
        switches all ranks to transport="host"."""
        if self.transport != "p2p":
            return True
        from .engine import AlignerError
        # the check's time budget: the engine's in-kernel waits (mi355sw_config.wait_seconds) and this driver's stall limit
        saved_wait, saved_stall, saved_prune = self.engine._opts["wait_seconds"], self.stall_abort_s, self.prune_blocks
        self.engine.configure(wait_seconds=budget_s)
        self.stall_abort_s = budget_s
        # the check compares boundary COLUMNS: with block pruning a skipped slab leaves a lower bound in them and which slabs are
        # skipped depends on when the chain's best score arrives -- two runs of the same chain may then differ cell by cell and
        # still both be right (seen at N = 8: bands 5-7, same best cell, different crc).  The check runs unpruned.
        self.prune_blocks = False
        got = {}
        try:
            for transport in ("p2p", "host"):
                self.transport = transport
                ok = 1
                try:
                    best = self.run(m, j0, j1, digest_inbound=True)
                    got[transport] = (tuple(int(x) for x in best), self.inbound_crc)
                except Exception as e:          # a wait that ran out, a stalled band, a refused mapping
                    self.p2p_error = "%s run of the transport check: %s" % (transport, e)
                    ok = 0
                    for fn in (self.engine.streamAbort, self.engine.streamEnd):
                        try:
                            fn()
                        except (AlignerError, AttributeError):
                            pass
                if all_min(ok) == 0:
                    return False
            same = 1 if got["p2p"] == got["host"] else 0
            if not same:
                self.p2p_error = "port run and host run of the transport check differ: %r vs %r" % (got["p2p"], got["host"])
            return all_min(same) == 1
        finally:
            self.transport = "p2p"
            self.stall_abort_s = saved_stall
            self.prune_blocks = saved_prune
            try:
                self.engine.configure(wait_seconds=saved_wait)
            except AlignerError:            # (a stream the check left active: the caller closes the engine)
                self.engine._opts["wait_seconds"] = saved_wait

    def run(self, m, j0, j1, recurrence=SMITH_WATERMAN, track_best=True, first_row_init_type=INIT_WITH_ZEROES,
            first_col_init_type=INIT_WITH_ZEROES, poll_sleep=0.0005, want_last_row=False, before_end=None,
            force_int32=False, digest_inbound=False, special_row_interval=0, special_row_sink=None, n_total=None,
            keep_inbound=False):
        """seq1 of the engine must already hold the whole horizontal sequence (or at least [j0,j1)).
        digest_inbound: leave the crc32 of the boundary column this band received in self.inbound_crc (host transport:
        the segments as they arrive; p2p: the port's memory, read back once the band is through).
        special_row_interval > 0: the band keeps its [j0, j1) slice of every special row (the reference: one Special
        Rows Area per forked node, work.tmp/FORK.NN, Job.cpp:123-128); `special_row_sink(dp_row, first_cell, cells)`
        gets each row as soon as its strip is complete -- cells = (H,F) of columns j0..j1-1, first_cell = the boundary
        column's cell of that row with f = -INF (AbstractDiagonalAligner.cpp:290-298).
        n_total: width of the whole matrix (default j1 of the last band = this band's j1): block pruning bounds what an
        alignment can still gain by the rows and columns left in the SUPER-partition (M3).
        keep_inbound: leave the whole boundary column this band received, corner cell first, in self.inbound_column
        ((m+1, 2) cells (H,E)) -- what the reference tees into C00000000.INIT_WITH_CUSTOM_DATA (sw_stage1.cpp:186-191);
        8 bytes of host memory per row (2 GB at C5's 249 M rows), like the file it becomes."""
        import zlib
        from .engine import AlignerError
        self.inbound_crc = 0 if digest_inbound else None
        eng, dist = self.engine, self.dist
        first, last = self.rank == 0, self.rank == self.world - 1
        self.inbound_column = np.empty((m + 1, 2), dtype=np.int32) if (keep_inbound and not first) else None
        p2p = self.transport == "p2p"
        part = Partition(0, j0, m, j1)
        seg = self.segment_rows
        nseg = (m + seg - 1) // seg
        # local alignments prune against the chain's running best score; global ones (NEEDLEMAN_WUNSCH with nothing tracked:
        # the answer is the last cell of the last band) against a running lower bound of that cell, AbstractBlockPruning.cpp:92-109
        prune = bool(self.prune_blocks and (track_best if recurrence == SMITH_WATERMAN else not track_best))
        kw = dict(recurrence_type=recurrence, track_best=track_best,
                  first_row_init_type=first_row_init_type, first_row_start_offset=j0,
                  want_last_column=(not last) and not p2p, last_column_port=(not last) and p2p,
                  want_last_row=want_last_row, force_int32=force_int32)
        if special_row_interval > 0:
            kw.update(special_row_interval=int(special_row_interval))
        if prune:
            # every band prunes against the best of the whole chain (share_best) and bounds the gain still possible
            # by the extents of the whole matrix, not of its own band
            kw.update(prune_blocks=True, prune_rows=m, prune_cols=(n_total if n_total is not None else j1) - j0,
                      share_best=self.world > 1)
        if first:
            kw.update(first_column_init_type=first_col_init_type)
        else:
            # corner cell = first-row cell at column j0-1+1 ... (H of DP cell (0, j0)): zero-initialised
            # rows give (0,-INF); custom rows are handled by the caller feeding row 0 of the stream.
            corner = np.array([[0, -INF]], dtype=np.int32)
            if first_row_init_type != INIT_WITH_ZEROES:
                open_ = 3 if first_row_init_type == 1 else 0
                corner[0, 0] = -2 * j0 - open_ if j0 > 0 else 0
            kw.update(first_column_init_type=INIT_WITH_CUSTOM_DATA, first_column=corner,
                      stream_first_column=not p2p, first_column_port=p2p)
            if self.inbound_column is not None:
                self.inbound_column[0] = corner[0]
        if p2p:
            self._exchange_ports(m)
        # Ordered start: band g+1 launches its kernel only after band g has launched its own.  It could do nothing
        # before band g's first strip is through anyway, and where bands SHARE a GPU (tests and rehearsals on a
        # one-GPU box) a kernel that sits polling for its neighbour has been seen to keep that neighbour's set-up
        # work (fills, the seed pass) off the GPU for tens of seconds.
        # The token always travels, whatever happens to this band's own start: 1 = started, 0 = failed (no memory for the
        # special rows, a port that is not there) -- a band that never hears from its left neighbour would sit in recv
        # with no kernel running and nothing for the stall watchdog to see, and so would every band to its right.
        # With the token travels what the pruning bound starts from: band 0 runs the diagonal seed pass of the WHOLE matrix
        # before it starts (the bands to its right could not start earlier anyway) and every band begins with that value.
        def token(ok, bound):
            import torch
            return torch.tensor([1 if ok else 0, NO_BOUND if bound is None else int(bound)], dtype=torch.int64)

        bound = None
        if not first and dist is not None:
            go = token(False, None)
            dist.recv(go, src=self.rank - 1)
            if int(go[0]) != 1:
                if not last:
                    dist.send(token(False, None), dst=self.rank + 1)
                raise RuntimeError("band %d/%d: a band to the left failed to start its kernel" % (self.rank, self.world))
            bound = None if int(go[1]) == NO_BOUND else int(go[1])
        try:
            if first and prune and self.seed_bound and self.world > 1 and n_total is not None:
                bound = chain_seed_bound(eng, m, n_total, recurrence, first_row_init_type, first_col_init_type)
            if prune and bound is not None:
                kw.update(initial_bound=bound)
            self.initial_bound = bound
            eng.streamBegin(part, **kw)
        except BaseException:
            if not last and dist is not None:
                dist.send(token(False, None), dst=self.rank + 1)
            raise
        if not last and dist is not None:
            dist.send(token(True, bound), dst=self.rank + 1)
        self.restarts = 0

        lock = threading.Lock()          # the engine handle is driven by one thread at a time
        errors = []
        fed = [0]                        # rows of the inbound column handed to the engine so far (host transport)

        def receiver():
            # inbound boundary column: blocking recv of row segments from the left neighbour
            try:
                for q in range(nseg):
                    r0 = q * seg
                    ln = min(seg, m - r0)
                    buf = self._tensor(ln)
                    dist.recv(buf, src=self.rank - 1)
                    if digest_inbound:
                        self.inbound_crc = zlib.crc32(buf.numpy().tobytes(), self.inbound_crc)
                    if self.inbound_column is not None:
                        self.inbound_column[r0 + 1:r0 + ln + 1] = buf.numpy()
                    if inbound_h is not None:          # H of the boundary column at every possible special row
                        q0 = r0 // 256 + 1
                        q1 = (r0 + ln) // 256
                        if q1 >= q0:
                            inbound_h[q0:q1 + 1] = buf.numpy()[q0 * 256 - 1 - r0:q1 * 256 - r0:256, 0]
                    with lock:
                        eng.streamFeedColumn(r0, buf.numpy())
                        fed[0] = r0 + ln
            except BaseException as e:     # surfaced by the main loop
                errors.append(e)

        def restart_int32():
            """the packed kernel left its exact range: everything handed out so far is exact (the engine only
            reports rows below the failing strip), so the band starts again on the int32 kernel and REPLAYS --
            the inbound rows already received are still in the engine's column (pinned, or the port), outbound
            segments already sent are not sent again (a port is simply written again with the same cells)."""
            try:
                eng.streamAbort()
            except AlignerError:
                pass
            try:
                eng.streamEnd()
            except AlignerError:
                pass
            kw2 = dict(kw, force_int32=True)
            if not first and not p2p:
                kw2["first_column_resume_rows"] = fed[0]
            eng.streamBegin(part, **kw2)
            self.restarts += 1
            reprobe_special()

        # special rows of this band: DP rows (multiples of the strip height, AbstractDiagonalAligner::isSpecialRow)
        # (special rows sit on multiples of the strip height, every strip height is a multiple of 256: the boundary
        #  column's H at every 256th row covers whatever geometry a restart on the int32 kernels picks)
        special_dp, special_next = [], [0]
        inbound_h = np.zeros(m // 256 + 2, dtype=np.int32) if (special_row_interval > 0 and not first and not p2p) else None
        if special_row_interval > 0:
            while True:
                try:
                    dp, _ = eng.streamReadSpecialRow(len(special_dp), 0, 0)
                except (AlignerError, IndexError):
                    break
                special_dp.append(int(dp))
        self.special_rows = []

        def reprobe_special():
            """after a restart on the int32 kernels (other strip heights): the rows already handed over stay, the rest
            are where the new geometry puts them"""
            if special_row_interval <= 0:
                return
            done_to = self.special_rows[-1] if self.special_rows else 0
            del special_dp[:]
            while True:
                try:
                    dp, _ = eng.streamReadSpecialRow(len(special_dp), 0, 0)
                except (AlignerError, IndexError):
                    break
                special_dp.append(int(dp))
            special_next[0] = sum(1 for dp in special_dp if dp <= done_to)

        def flush_special(rows_ok):
            """hands over every special row whose strip is complete; called with the lock held"""
            while special_next[0] < len(special_dp) and special_dp[special_next[0]] <= rows_ok:
                k, dp = special_next[0], special_dp[special_next[0]]
                if not first and not p2p and fed[0] < dp:
                    return                                  # its boundary cell is still on the way
                _, cells = eng.streamReadSpecialRow(k)
                if first:
                    c0 = np.array([0 if first_col_init_type == INIT_WITH_ZEROES
                                   else -2 * dp - (3 if first_col_init_type == 1 else 0), -INF], dtype=np.int32)
                elif p2p:
                    c0 = np.array([int(eng.portRead(dp - 1, 1)[0, 0]), -INF], dtype=np.int32)
                else:
                    c0 = np.array([int(inbound_h[dp // 256]), -INF], dtype=np.int32)
                if special_row_sink is not None:
                    special_row_sink(dp, c0, cells)
                self.special_rows.append(dp)
                special_next[0] += 1

        # host transport: the chain's running best travels between the rank processes (p2p: between the kernels)
        self.hints = 0
        bx_stop, bx = threading.Event(), None
        if prune and self.world > 1 and not p2p and self._best_group is not None:
            def best_exchange():
                import torch
                known = -INF
                try:
                    while True:
                        mine = eng.streamRunningBest() if not bx_stop.is_set() else -INF
                        t = torch.tensor([max(mine, known), -1 if bx_stop.is_set() else 0], dtype=torch.int64)
                        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._best_group)
                        if int(t[0]) > known:
                            known = int(t[0])
                            if known > mine and not bx_stop.is_set():
                                try:
                                    eng.streamBestHint(known)
                                    self.hints += 1
                                except AlignerError:
                                    pass
                        if int(t[1]) == -1:                  # every band is through
                            return
                        time.sleep(0.002)
                except BaseException as e:
                    errors.append(e)
            bx = threading.Thread(target=best_exchange, daemon=True)
            bx.start()

        rx = None
        if not first and not p2p:
            rx = threading.Thread(target=receiver, daemon=True)
            rx.start()
        send_q = 0
        # a chain that stands still says where: one line per 10 s without progress on this band (rows completed,
        # segments sent), one per second with MI355SW_BAND_DEBUG=1; after MI355SW_BAND_STALL_S seconds (default
        # 900) without progress the band gives up instead of hanging its neighbours for ever
        debug = os.environ.get("MI355SW_BAND_DEBUG") == "1"
        stall_abort = self.stall_abort_s if self.stall_abort_s is not None else float(os.environ.get("MI355SW_BAND_STALL_S", "900"))
        t_dbg = t_prog = time.time()
        seen = (-1, -1)
        rows_done, fin = 0, False
        while True:
            now = time.time()
            if now - t_dbg >= (1.0 if debug else 10.0):
                t_dbg = now
                if (rows_done, send_q) != seen:
                    seen, t_prog = (rows_done, send_q), now
                if debug or now - t_prog >= 10.0:
                    sys.stderr.write("[band %d/%d] rows_done=%d/%d finished=%d segments_sent=%d/%d no progress for %.0f s\n"
                                     % (self.rank, self.world, rows_done, m, int(fin), send_q, nseg, now - t_prog))
                    sys.stderr.flush()
                if now - t_prog >= stall_abort:
                    errors.append(RuntimeError("band %d/%d: no progress for %.0f s (rows_done=%d/%d, segments_sent=%d/%d)"
                                               % (self.rank, self.world, now - t_prog, rows_done, m, send_q, nseg)))
            if errors:
                bx_stop.set()
                with lock:
                    eng.streamAbort()
                    try:
                        eng.streamEnd()
                    except AlignerError:
                        pass
                if bx is not None:           # (it leaves its loop at the next all_reduce in which every band has said stop)
                    bx.join(timeout=5.0)
                raise errors[0]
            with lock:
                try:
                    rows_done, fin = eng.streamPoll()
                except AlignerError as e:
                    if "EOVERFLOW16" not in str(e) or kw.get("force_int32") or self.restarts > 0:
                        raise
                    if os.environ.get("MI355SW_VERBOSE"):
                        print("[bands] band %d/%d restarts on the int32 kernels: %s" % (self.rank, self.world, e), file=sys.stderr)
                    restart_int32()
                    continue
                # (device-resident rows are read through the copy stream: only once nothing more has to be fed, so that
                #  a copy held up behind the running kernel cannot starve that kernel of its first column)
                if special_dp and (first or p2p or fed[0] >= m or fin):
                    flush_special(rows_done)
            progressed = False
            # outbound boundary column (host transport): every complete segment goes to the right neighbour
            while not last and not p2p and send_q < nseg:
                r0 = send_q * seg
                ln = min(seg, m - r0)
                if rows_done < r0 + ln:
                    break
                buf = self._tensor(ln)
                with lock:
                    import torch
                    buf.copy_(torch.from_numpy(eng.streamReadColumn(r0, ln)))
                dist.send(buf, dst=self.rank + 1)
                send_q += 1
                progressed = True
            if fin and (last or p2p or send_q >= nseg):
                break
            if not progressed:
                time.sleep(poll_sleep)
        if rx is not None:
            rx.join()
        if special_dp:
            with lock:
                flush_special(m)
        if digest_inbound and p2p and not first:
            self.inbound_crc = zlib.crc32(np.ascontiguousarray(eng.portRead(0, m), dtype=np.int32).tobytes())
        if self.inbound_column is not None and p2p:
            self.inbound_column[1:] = eng.portRead(0, m)
        if before_end is not None:       # e.g. read this band's slice of the last row while the stream is open
            before_end(eng)
        bx_stop.set()                   # the exchange thread keeps answering the other bands until every band is through
        try:
            best, _ = eng.streamEnd()
        except AlignerError as e:        # e.g. reported by the exact-position pass of a very tall band
            if "EOVERFLOW16" not in str(e) or kw.get("force_int32") or self.restarts > 0:
                raise
            with lock:
                kw2 = dict(kw, force_int32=True)
                if not first and not p2p:
                    kw2["first_column_resume_rows"] = fed[0]
                eng.streamBegin(part, **kw2)
                self.restarts += 1
            while True:
                rows_done, fin = eng.streamPoll()
                if fin:
                    break
                time.sleep(poll_sleep)
            if before_end is not None:
                before_end(eng)
            best, _ = eng.streamEnd()
        if bx is not None:
            bx.join()
        if errors:
            raise errors[0]
        return best

    def reduce_best(self, best):
        """global best = canonical max over bands (reference: relay through AlignerPool files)."""
        if self.dist is None or self.world == 1:
            return best
        import torch
        t = torch.tensor(list(best), dtype=torch.int64, device=self.device if self.device is not None else "cpu")
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        best = canonical_best([tuple(int(x) for x in o.tolist()) for o in out])
        if self.prune_blocks:
            check_chain_bound(best[2], self.initial_bound)
        return best


class InProcessChain:
    """The whole band chain driven by ONE host process: handle k runs band k on GPU `devices[k]`, the boundary columns
    travel through column ports attached with mi355sw_port_attach -- hipDeviceEnablePeerAccess between the devices of one
    process, no hipIpc handle and no second process involved.  The streaming calls never block (begin launches the
    persistent kernel, poll reads pinned words), so one thread serves all bands.

    It is the SECOND device-side transport of bench.py: when the ranks' hipIpc mappings fail their check (probe_p2p /
    verify_p2p) rank 0 runs the chain this way ("comm": "p2p-attach") before anybody falls back to pinned host columns +
    gloo; and it is how a single process uses several GPUs (tools, tests).  The reference has no such mode: its bands are
    forked processes chained by sockets (M/libmasa/libmasa.cpp:540-642).
    Score passes only (best cell, or H[m][n] of a global alignment): special rows per band are BandRunner's job."""

    def __init__(self, aligners, prune_blocks=False, seed_bound=True):
        self.aligners = list(aligners)
        self.prune_blocks = prune_blocks
        self.seed_bound = seed_bound       # pruning: the diagonal seed of the whole matrix first (chain_seed_bound)
        self.initial_bound = None
        self.restarts = 0
        self._rows = 0

    def attach(self, m):
        """ports of m rows between consecutive bands (made once; a later run of at most m rows resets them)"""
        als = self.aligners
        if self._rows >= m:
            for k in range(1, len(als)):
                als[k].portReset()
            return
        for k in range(1, len(als)):
            als[k].portCreate(m)
            als[k - 1].portAttach(als[k])
        self._rows = m

    def run(self, m, limits, recurrence=SMITH_WATERMAN, first_row_init_type=INIT_WITH_ZEROES, first_col_init_type=INIT_WITH_ZEROES,
            force_int32=False, poll_sleep=0.0005):
        """limits = band_limits(n, weights): band k = columns [limits[k], limits[k+1]).  Returns (best, per-band statistics):
        best = the canonical best cell of the chain (0-based), or for NEEDLEMAN_WUNSCH (m-1, n-1, H[m][n])."""
        from .engine import AlignerError
        als, N = self.aligners, len(self.aligners)
        n = limits[-1]
        sw = recurrence == SMITH_WATERMAN
        self.attach(m)
        self.initial_bound = None
        if self.prune_blocks and self.seed_bound and N > 1:
            self.initial_bound = chain_seed_bound(als[0], m, n, recurrence, first_row_init_type, first_col_init_type)
        for attempt in (0, 1):
            begun = []
            try:
                for k in range(N):                         # left to right: band k+1's kernel starts after band k's
                    kw = dict(recurrence_type=recurrence, track_best=sw, first_row_init_type=first_row_init_type,
                              first_row_start_offset=limits[k], last_column_port=k < N - 1, want_last_row=(not sw and k == N - 1),
                              force_int32=bool(force_int32 or attempt))
                    if self.prune_blocks:
                        kw.update(prune_blocks=True, prune_rows=m, prune_cols=n - limits[k], share_best=N > 1)
                        if self.initial_bound is not None:
                            kw.update(initial_bound=self.initial_bound)
                    if k == 0:
                        kw.update(first_column_init_type=first_col_init_type)
                    else:
                        corner = np.array([[0, -INF]], dtype=np.int32)
                        if first_row_init_type != INIT_WITH_ZEROES:
                            corner[0, 0] = -2 * limits[k] - (3 if first_row_init_type == 1 else 0)
                        kw.update(first_column_init_type=INIT_WITH_CUSTOM_DATA, first_column=corner, first_column_port=True)
                    als[k].streamBegin(Partition(0, limits[k], m, limits[k + 1]), **kw)
                    begun.append(k)
                open_ = set(range(N))
                while open_:
                    for k in sorted(open_):
                        _rows, fin = als[k].streamPoll()
                        if fin:
                            open_.discard(k)
                    if open_:
                        time.sleep(poll_sleep)
                bests, stats, h_last = [], [], None
                for k in range(N):
                    if not sw and k == N - 1:
                        h_last = int(als[k].streamReadLastRow(col=limits[k + 1] - limits[k] - 1, length=1)[0, 0])
                    b, _ = als[k].streamEnd()
                    bests.append(b)
                    stats.append(als[k].getStatistics())
                best = canonical_best(bests) if sw else (m - 1, n - 1, h_last)
                if self.prune_blocks:
                    check_chain_bound(best[2], self.initial_bound)
                return best, stats
            except AlignerError as e:
                for k in begun:                            # stop whatever is running, then the whole chain again on the int32 kernels
                    for fn in (als[k].streamAbort, als[k].streamEnd):
                        try:
                            fn()
                        except AlignerError:
                            pass
                if "EOVERFLOW16" not in str(e) or force_int32 or attempt == 1:
                    raise
                self.restarts += 1
                self.attach(m)


def band_stage1(runner, m, j0, j1, work, sra_limit, n_total=None, recurrence=SMITH_WATERMAN,
                first_row_init_type=INIT_WITH_ZEROES, first_col_init_type=INIT_WITH_ZEROES, **run_kw):
    """Stage 1 of one band with its Special Rows Area: what a forked MASA-Core node leaves in its own work directory
    (`work`/FORK.NN, Job::initializeWorkPath, M/common/Job.cpp:118-128) -- partition directory
    `00000000.<j0>.<m>.<j1>` in absolute coordinates, one file per special row with (j1-j0+1) cells whose leading cell
    is the boundary column's, the last row as the completion marker, the border markers (row marker at offset j0), and
    for every band but the first the boundary column it received as `C00000000.INIT_WITH_CUSTOM_DATA` (the tee of
    sw_stage1.cpp:186-191), so that stage 2 can walk back through this band and hand over to the band on its left at
    that column.  The spacing of the rows follows `sra_limit` (the node's --disk-size) and the size of the WHOLE
    matrix, m x n_total -- a split is a trim, and Job.cpp:62-67 computes the interval from the untrimmed sizes --
    rounded up to the engine's strip height.
    Collective: every rank of the chain calls it.  Returns {"best": the chain's best (i, j, score) in 1-based DP
    coordinates as stage1.py's, "band_best": this band's own, as the engine reports it (0-based cell), "work":
    this band's work directory, "special_rows": DP rows written}; `crosspoints/crosspoint_01.00` of every band holds
    the chain's best (the reference: the running best relayed along the chain, sw_stage1.cpp:421-464 -- the last
    node's file is the one that counts there)."""
    from . import sra as sra_mod
    first = runner.rank == 0
    bwork = os.path.join(work, "FORK.%02d" % runner.rank) if runner.world > 1 else work
    os.makedirs(bwork, exist_ok=True)
    interval = sra_mod.flush_interval(m, n_total if n_total is not None else j1, sra_limit) if sra_limit > 0 else 0
    area = sra_mod.SpecialRowsArea(sra_mod.special_rows_path(bwork, 1, 0), disk_limit=max(sra_limit, 1))
    part = area.create_partition(0, j0, m, j1)
    part.set_border_markers(first_row_init_type, j0, first_col_init_type if first else INIT_WITH_CUSTOM_DATA, 0)

    def sink(dp, c0, cells):
        part.write(dp, np.concatenate([np.asarray(c0, dtype=np.int32).reshape(1, 2), cells]))

    got = {}
    prev_end = run_kw.pop("before_end", None)

    def before_end(eng):
        got["row"] = eng.streamReadLastRow()
        if prev_end is not None:
            prev_end(eng)
    band_best = runner.run(m, j0, j1, recurrence=recurrence, first_row_init_type=first_row_init_type,
                           first_col_init_type=first_col_init_type, special_row_interval=interval,
                           special_row_sink=sink if interval > 0 else None, n_total=n_total, want_last_row=True,
                           before_end=before_end, keep_inbound=not first, **run_kw)
    if first:
        lead = 0 if first_col_init_type == INIT_WITH_ZEROES else -2 * m - (3 if first_col_init_type == 1 else 0)
    else:
        lead = int(runner.inbound_column[m, 0])
        runner.inbound_column.tofile(os.path.join(part.path, "C00000000.INIT_WITH_CUSTOM_DATA"))
    if m not in runner.special_rows:              # the partition's last row: sw_stage1.cpp:203-240's completion marker
        part.write(m, np.concatenate([np.array([[lead, -INF]], dtype=np.int32), got["row"]]))
    part.close()
    best = runner.reduce_best(band_best)
    if best[0] >= 0 or best[1] >= 0:
        best = (best[0] + 1, best[1] + 1, best[2])       # the engine reports the 0-based cell, the files DP coordinates
        sra_mod.write_crosspoint(sra_mod.crosspoint_path(bwork, 1, 0), best)
    return {"best": tuple(best), "band_best": tuple(band_best), "work": bwork,
            "special_rows": list(runner.special_rows) + ([m] if m not in runner.special_rows else [])}

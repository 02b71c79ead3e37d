"""Native Stage-2 driver: from the end point of the optimal alignment back to its start, one crosspoint per special
row of stage 1 -- what MASA-Core's stage2() does around the aligner (M/stage2/sw_stage2.cpp:49-129 find_next_crosspoint,
:237-524 stage2).

The stage works in the REVERSED, TRANSPOSED matrix: its vertical sequence is S1 reversed, its horizontal one S0
reversed, so that a special ROW of stage 1 is a COLUMN here -- the last column of a partition that starts at the
current crosspoint.  The aligner sweeps that partition with the global (NW) recurrence and gap-initialised borders;
the manager adds, cell by cell of the dispatched last column, the reverse value to the forward value stage 1 stored in
the special row, and the first cell where the sum is the goal score (the forward score of the current crosspoint) is
the next crosspoint (AlignerManager::findGoalCell).  The aligner is told to stop there (mustContinue), the special
rows this stage saved for stage 3 are cut back to the crosspoint, and the walk goes on from it.  With a local
alignment the walk ends inside a partition, at the cell whose reverse value alone is the goal.

Everything is written in the reference's formats (crosspoints.py, sra.py): `crosspoint_02.NN` (in this stage's
reversed coordinates, like MASA-Core's) and `special_rows/stage.02.NN/`."""
import os
import time

import numpy as np

from .engine import NEEDLEMAN_WUNSCH, Partition
from .manager import (AlignerManager, InitialCellsReader, ReversedCellsReader, BacktraceLost, AT_ANYWHERE,
                      AT_SEQUENCE_1_OR_2, GAP_OPEN, GAP_EXT, INIT_WITH_GAPS, INIT_WITH_GAPS_OPENED)
from .crosspoints import Crosspoint, CrosspointsFile, crosspoint_file, TYPE_MATCH, TYPE_GAP_1, TYPE_GAP_2
from . import sra as sra_mod

MATCH_SCORE = 1
MIN_ROW_DISTANCE = 128          # sw_stage2.cpp:420: special rows nearer than this to the crosspoint are skipped
# a sweep from a GUESSED crosspoint is given up this many rows + twice its width down its last column, and its partition ends
# GUESS_PARTITION_SLACK rows below that (the engine hands the column over in chunks of up to 16 strips)
GUESS_CAP_SLACK = 4096
GUESS_PARTITION_SLACK = 16384


def speculation_default():
    """stage 2 starts its sweeps side by side from guessed crosspoints (_Speculation) unless MI355SW_STAGE2_SPECULATE=0"""
    return os.environ.get("MI355SW_STAGE2_SPECULATE", "1") not in ("", "0")


def _as_u8(seq):
    return np.frombuffer(bytes(seq), dtype=np.uint8) if isinstance(seq, (bytes, bytearray)) else np.asarray(seq, dtype=np.uint8)


def prepare_next_crosspoint(mgr, area, c0, c1, alignment_start, must_find=True, goal_location=None, scratch=False):
    """the set-up of find_next_crosspoint: goal, borders and special-rows partition of the sweep c0 -> c1.  Returns
    (special-rows partition, partition for the aligner or None if the manager met the goal without it).
    scratch: the special-rows partition is made under a name no reader looks for (a sweep from a guessed crosspoint: it
    moves into place when the walk accepts it, SpecialRowsArea.truncate_partition)."""
    first_row = InitialCellsReader(0 if c0.type == TYPE_GAP_1 else GAP_OPEN, GAP_EXT)
    first_col = InitialCellsReader(0 if c0.type == TYPE_GAP_2 else GAP_OPEN, GAP_EXT)
    if not must_find:
        mgr.unsetGoalScore()
    elif goal_location is not None:
        mgr.setGoalScore(c0.score, goal_location)
    elif alignment_start == AT_ANYWHERE and c0.score <= (c1.j - c0.j + 1) * MATCH_SCORE:
        mgr.setGoalScore(c0.score, AT_ANYWHERE)          # short enough to hold the alignment's start
    else:
        mgr.setGoalScore(c0.score, AT_SEQUENCE_1_OR_2)
    part = area.create_partition(c0.i, c0.j, c1.i, c1.j, scratch=scratch)
    part.set_first_column_reader(first_col)
    part.set_first_row_reader(first_row)
    mgr.setSpecialRowsPartition(part)
    try:
        adj = mgr.prepareAlign(Partition(c0.i, c0.j, c1.i, c1.j), c0.type)
    except BaseException:
        part.close()
        raise
    return part, adj


def conclude_next_crosspoint(mgr, area, part, c0, c1, must_find=True):
    """what find_next_crosspoint does once the aligner is through with the sweep"""
    if not must_find:
        part.close()
        return c1.copy()
    if not mgr.isFoundCrosspoint():
        part.close()
        raise BacktraceLost("backtrace lost: partition (%d,%d)-(%d,%d) swept without meeting the goal %d"
                            % (c0.i, c0.j, c1.i, c1.j, c0.score))
    i, j, score, typ = mgr.getNextCrosspoint()
    area.truncate_partition(part, i, j)
    return Crosspoint(i, j, score, typ)


def find_next_crosspoint(mgr, area, c0, c1, alignment_start, must_find=True, goal_location=None):
    """sw_stage2.cpp:49-129 / sw_stage3.cpp:49-122: aligns partition c0 -> c1 (both in the coordinates of the running
    orientation), returns the crosspoint where the goal c0.score is met on the last column / last row / inside"""
    part, adj = prepare_next_crosspoint(mgr, area, c0, c1, alignment_start, must_find, goal_location)
    if adj is not None:
        try:
            mgr.aligner.alignPartition(adj, mgr)
        except BaseException:
            part.close()
            raise
    return conclude_next_crosspoint(mgr, area, part, c0, c1, must_find)


class _GuessManager(AlignerManager):
    """The manager of a sweep that starts from a GUESSED crosspoint (_Speculation).  A wrong guess shows as a border sum above
    its goal -- "backtrace lost" for a real walk, just a wrong guess here -- or as a goal that never turns up: the sweep is
    stopped (mustContinue) instead of raising, at the latest `row_cap` rows down its last column, and says so in `gave_up`."""

    @classmethod
    def of(cls, mgr, row_cap):
        c = cls(mgr.aligner)
        c.recurrence, c.block_pruning, c.special_row_interval = mgr.recurrence, mgr.block_pruning, mgr.special_row_interval
        c.seq0_offset, c.seq1_offset = mgr.seq0_offset, mgr.seq1_offset
        c.row_cap, c.gave_up = int(row_cap), None
        return c

    def _find_goal_cell(self, buf, length, reader):
        try:
            return AlignerManager._find_goal_cell(self, buf, length, reader)
        except BacktraceLost as e:
            self.gave_up = str(e)
            self.active = False
            return None

    def dispatchColumn(self, j, buf, length):
        AlignerManager.dispatchColumn(self, j, buf, length)
        if self.active and not self.found and self.last_column_pos > self.row_cap:
            self.gave_up = "no goal within %d rows" % self.row_cap
            self.active = False


class _Speculation:
    """Stage 2's walk is a chain: the sweep towards special row k+1 starts at the crosspoint the sweep towards row k found,
    so the sweeps run one after the other, each a few hundred thousand rows of latency on a GPU that could take dozens
    side by side (C3: 68 sweeps, 0.2 s each).  But where the crosspoints WILL be can be guessed before any of them ran:
    the optimal alignment crosses a special row of stage 1 where that row's H is largest (the prefix of an optimal
    alignment is the best alignment that ends there), as an aligned pair, with that H as its score.  So: one sweep from the
    real crosspoint and one from the guess on every row further up, all handed to the aligner together
    (MI355Aligner.alignPartitions: one kernel launch); then the walk takes them in order -- a sweep counts only if it started
    at exactly the crosspoint (cell, score AND type) the sweep before it found.  From such a start it is the sweep the
    plain walk would have made: same borders, same goal, same special rows saved, same hit.  A guess that is off (the path
    crosses the row inside a gap, or one of several co-optimal paths is met first) costs its sweep, which is thrown away
    (SpecialRowsArea.discard_partition), and the walk makes that one step the plain way.  The files of the stage do not
    change.  The special rows an accepted sweep saves are the rows the chain's sweep would have saved (both are cut back to the
    crosspoint), so the area holds what the budget was planned for plus the rows of the sweeps still to be thrown away.  No counterpart in the reference (its stage 2 is the plain chain, sw_stage2.cpp:387-441).

    The default since round 6 (BASELINE config 3 on one MI355X: stage 2 15.8 s as a chain, 7.8 s this way, the same
    alignment.00.txt and crosspoint_04.00 -- profiles/r06_native_pipeline_c3_48Mx46M_stage2_guessed.json);
    stage2(speculate=False) or MI355SW_STAGE2_SPECULATE=0 walks the plain chain.  The row maxima are recorded by stage 1 while
    it writes the rows (SpecialRowsPartition.peaks), read back from the rows where they were not."""

    def __init__(self, mgr, area, part1, cp, col_reader, len_v, len_h, alignment_start):
        self.area = area
        self.sweeps = {}
        self.made = self.accepted = self.discarded = 0
        ids = [0] + part1.rows

        def target(i_abs, idx):                         # SpecialRowsPartition.next_special_row's choice, without its state
            while idx >= 0:
                dist = (i_abs - part1.i0) - ids[idx]
                if idx == 0:
                    return 0 if dist > 0 else None
                if dist > MIN_ROW_DISTANCE:
                    return idx
                idx -= 1
            return None

        chain = []                                       # (crosspoint the sweep starts from, index of its row, guessed?)
        c0, idx, guess = cp.copy(), part1._reading_idx, False
        while True:
            c0_r = c0.reverse(len_v, len_h)
            if (alignment_start == AT_ANYWHERE and c0.score <= 0) or c0_r.i <= part1.i0 or c0_r.j <= part1.j0:
                break
            t = target(c0_r.i, idx)
            if t is None:
                break
            chain.append((c0, t, guess))
            peak = part1.row_peak(ids[t], c0_r.j - part1.j0)
            if peak is None:
                break
            c0 = Crosspoint(part1.i0 + ids[t], part1.j0 + peak[1], peak[0], TYPE_MATCH).reverse(len_h, len_v)
            idx, guess = t, True
        if len(chain) < 2:
            return
        jobs = []
        try:
            for c0, t, guess in chain:
                c0_r = c0.reverse(len_v, len_h)
                c1 = Crosspoint(len_v - part1.j0, len_h - (part1.i0 + ids[t]))
                # A guessed sweep is given up `cap` rows down its last column at the latest (_GuessManager), so its partition
                # need not be taller than that: the engine sizes its buffers -- first column, special rows of its own -- by
                # the partition's height, and dozens of sweeps as tall as what is left of the sequence (C3: 22 M rows, 15 GB of
                # special-row room each) do not fit side by side.  The rows above the cut do not depend on what lies below.
                cap = 2 * (c1.j - c0.j) + GUESS_CAP_SLACK
                c1p = Crosspoint(min(c1.i, c0.i + cap + GUESS_PARTITION_SLACK), c1.j) if guess else c1
                m = _GuessManager.of(mgr, cap) if guess else mgr.clone()
                row = sra_mod.SpecialRowReader(part1, ids[t])
                row.seek(abs(c0_r.j - part1.j0) + 1)
                m.setLastColumnReader(row)
                if col_reader is not None and c1p.i == c1.i:      # (a cut partition's last row is not the matrix's border)
                    col = ReversedCellsReader(col_reader)
                    col.seek(c0_r.i - part1.i0 + 1)
                    m.setLastRowReader(col)
                part, adj = prepare_next_crosspoint(m, area, c0, c1p, alignment_start, scratch=guess)
                sw = dict(mgr=m, part=part, c0=c0, c1=c1, guess=guess)
                self.sweeps[(c0.astuple(), c1.astuple())] = sw
                if adj is not None:
                    jobs.append((m, adj))
            self.made = len(self.sweeps)
            if len(jobs) > 1 and hasattr(mgr.aligner, "alignPartitions"):
                # tall strips: a batch of sweeps that run to their goals hundreds of thousands of rows down is throughput-bound
                # (1024-row strips: 4.6 TCUPS against the 2.7 of the 256-row strips stage 3's small partitions get), and 1024
                # keeps the rows this stage stores for stage 3 on CUDAlign's 8192-row grid like the chain's 512
                kw = {"rows_per_lane": 16} if 16 in getattr(mgr.aligner, "batch_rows_per_lane_choices", ()) else {}
                mgr.aligner.alignPartitions([a for _, a in jobs], [m for m, _ in jobs], **kw)
            else:
                for m, a in jobs:
                    mgr.aligner.alignPartition(a, m)
        except BaseException:
            self.discard_rest()
            raise

    def take(self, cp, c1):
        """the finished sweep cp -> c1, or None when the walk has to make this step itself"""
        sw = self.sweeps.pop((cp.astuple(), c1.astuple()), None)
        if sw is not None and sw["guess"] and (sw["mgr"].gave_up or not sw["mgr"].isFoundCrosspoint()):
            self._discard(sw)                        # (a right start, but the goal lay beyond the cap: the plain way)
            sw = None
        if sw is None:
            # a sweep from the same CELL with another score or type would leave its directory where the walk's own goes
            for key in [k for k, o in self.sweeps.items() if (o["c0"].i, o["c0"].j, o["c1"].i, o["c1"].j) == (cp.i, cp.j, c1.i, c1.j)]:
                self._discard(self.sweeps.pop(key))
            return None
        self.accepted += 1
        return sw

    def _discard(self, sw):
        self.area.discard_partition(sw["part"])
        self.discarded += 1

    def discard_rest(self):
        for sw in self.sweeps.values():
            self._discard(sw)
        self.sweeps = {}


@sra_mod.with_async_files
def stage2(aligner, seq0, seq1, work, alignment_start=AT_ANYWHERE, sra_limit=0, ident=0, bounds=None, ram_limit=0,
           areas=None, speculate=None):
    """Runs stage 2 for alignment `ident` of work directory `work` (stage 1 must have left crosspoint_01.NN and,
    with sra_limit > 0, its special rows there).  seq0 / seq1: the whole sequences; `bounds` = (i0, j0, i1, j1) the
    part --trim selected for stage 1 (only its origin matters here: where a global alignment must begin).  Returns {"crosspoints": [(type, i, j, score), ...] as written to
    crosspoint_02.NN, "end": the last crosspoint in ORIGINAL coordinates, "partitions", "seconds"}.
    speculate: sweeps from guessed crosspoints side by side (_Speculation); None = on unless MI355SW_STAGE2_SPECULATE=0."""
    t_start = time.time()
    if speculate is None:
        speculate = speculation_default()
    guessed = {"sweeps": 0, "accepted": 0, "discarded": 0}
    s0, s1 = _as_u8(seq0), _as_u8(seq1)
    m, n = len(s0), len(s1)
    seq_v = np.ascontiguousarray(s1[::-1])               # sw_stage2.cpp:258-276: reverse = 1
    seq_h = np.ascontiguousarray(s0[::-1])
    len_v, len_h = n, m
    bi0, bj0, bi1, bj1 = bounds if bounds is not None else (0, 0, m, n)
    budget = max(sra_limit, 0) + max(ram_limit, 0)
    area1 = sra_mod.get_area(areas, work, 1, 0, ram_limit=ram_limit, disk_limit=sra_limit)
    area2 = sra_mod.get_area(areas, work, 2, ident, ram_limit=ram_limit, disk_limit=sra_limit)
    cps1 = CrosspointsFile(crosspoint_file(work, 1, ident)).load()
    if not cps1:
        raise RuntimeError("stage 2: no crosspoint_01.%02d in %s" % (ident, work))
    cps1.reverse_all(len_h, len_v)                       # :294
    cp = cps1[0].copy()
    cp_r = cp.reverse(len_v, len_h)                      # back in stage 1's coordinates
    if (cp_r.j <= bj0 or cp_r.j > bj1) and cp_r.j != bj0:     # :299-305: not this process's columns
        return {"crosspoints": [], "end": cp_r.astuple(), "partitions": 0, "seconds": 0.0}
    part1 = area1.open_partition_at(cp_r.i, cp_r.j)
    mgr = AlignerManager(aligner)
    mgr.setRecurrenceType(NEEDLEMAN_WUNSCH)
    mgr.setBlockPruning(False)
    if part1 is not None and budget > 0:
        mgr.setSpecialRowInterval(sra_mod.flush_intervals(m, n, budget)[1])
    else:
        mgr.setSpecialRowInterval(0)
    if speculate:
        area2.remove_scratch_partitions()                # (what a run that died between its batch and its walk left behind)
    out = CrosspointsFile(crosspoint_file(work, 2, ident)).open()
    out.write(cp)
    if cp.type != TYPE_MATCH:
        cp.score += GAP_OPEN
    col_reader = row_reader = None
    partitions = 0
    crossing = True
    # Stage 2's partitions are tall and stopped by their goal a few hundred thousand rows down: what they cost is the
    # first strip's sweep plus one hop per strip down to the goal row, and 512-row strips halve that against the 2048-row
    # strips the engine's cost model picks for a full sweep of such a shape (mi355sw_set_rows_per_lane).  Only when the
    # caller left the height to the engine.
    # (Round 4 tried heights that follow the partition's width -- the goal lies about as many rows down as the partition is
    #  wide, and the strips of a second round of wavefronts come a whole sweep late.  C3, 700 k-column partitions: 768-row
    #  strips 13.1 s against 14.8 s, but the special rows this stage stores for stage 3 then leave CUDAlign's 8192-row grid
    #  and stage 3 picks another, equally optimal alignment; 1024-row strips, which keep the grid, were slower: 16.8 s.)
    short_strips = hasattr(aligner, "setRowsPerLane") and aligner.getRowsPerLane() == 0
    if short_strips:
        aligner.setRowsPerLane(8)
    try:
        while crossing and part1 is not None:
            col_reader, row_reader = part1.first_column_reader, part1.first_row_reader
            corner = Crosspoint(part1.i0, part1.j0).reverse(len_h, len_v)
            mgr.setSequences(seq_v, seq_h, cp.i, cp.j, corner.i, corner.j)
            spec = None
            try:
                if speculate:
                    spec = _Speculation(mgr, area2, part1, cp, col_reader, len_v, len_h, alignment_start)
                while True:
                    cp_r = cp.reverse(len_v, len_h)
                    if alignment_start == AT_ANYWHERE and cp.score <= 0:
                        crossing = False
                        break
                    if cp_r.i <= part1.i0 or cp_r.j <= part1.j0:
                        break
                    row = part1.next_special_row(cp_r.i, cp_r.j, MIN_ROW_DISTANCE)
                    mgr.setLastColumnReader(row)
                    if col_reader is not None:
                        col = ReversedCellsReader(col_reader)
                        col.seek(cp_r.i - part1.i0 + 1)
                        mgr.setLastRowReader(col)
                    else:
                        mgr.setLastRowReader(None)
                    c1 = Crosspoint(len_v - part1.j0, len_h - part1.get_reading_row())
                    done = spec.take(cp, c1) if spec is not None else None
                    if done is not None:
                        cp = conclude_next_crosspoint(done["mgr"], area2, done["part"], cp, c1)
                    else:
                        cp = find_next_crosspoint(mgr, area2, cp, c1, alignment_start)
                    partitions += 1
                    out.write(cp)
                    if cp.type != TYPE_MATCH:
                        cp.score += GAP_OPEN
            finally:
                if spec is not None:
                    spec.discard_rest()
                    for k, v in (("sweeps", spec.made), ("accepted", spec.accepted), ("discarded", spec.discarded)):
                        guessed[k] += v
                mgr.unsetSequences()
            cp_r = cp.reverse(len_v, len_h)
            part1 = area1.open_partition_at(cp_r.i, cp_r.j)
        if cp.score != 0:
            # :461-481: a global / semi-global alignment that reached a gap-initialised border away from the origin
            # runs along that border to the origin
            origin_r = Crosspoint(bi0, bj0, 0, TYPE_MATCH)
            gapped = (INIT_WITH_GAPS, INIT_WITH_GAPS_OPENED)
            if ((col_reader is not None and col_reader.getType() in gapped and cp_r.j == origin_r.j and cp_r.i != origin_r.i) or
                    (row_reader is not None and row_reader.getType() in gapped and cp_r.i == origin_r.i and cp_r.j != origin_r.j)):
                cp_r = origin_r
                cp = origin_r.reverse(len_h, len_v)
                out.write(cp)
    finally:
        out.close()
        if short_strips:
            aligner.setRowsPerLane(0)
    return {"crosspoints": out.tuples(), "end": cp_r.astuple(), "partitions": partitions, "seconds": time.time() - t_start,
            "speculation": guessed if speculate else None}

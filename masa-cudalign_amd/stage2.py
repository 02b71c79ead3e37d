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
import time

import numpy as np

from .engine import NEEDLEMAN_WUNSCH, Partition
from .manager import (AlignerManager, InitialCellsReader, ReversedCellsReader, BacktraceLost, AT_ANYWHERE,
                      AT_SEQUENCE_1_OR_2, GAP_OPEN, GAP_EXT, INIT_WITH_GAPS, INIT_WITH_GAPS_OPENED)
from .crosspoints import Crosspoint, CrosspointsFile, crosspoint_file, TYPE_MATCH, TYPE_GAP_1, TYPE_GAP_2
from . import sra as sra_mod

MATCH_SCORE = 1
MIN_ROW_DISTANCE = 128          # sw_stage2.cpp:420: special rows nearer than this to the crosspoint are skipped


def _as_u8(seq):
    return np.frombuffer(bytes(seq), dtype=np.uint8) if isinstance(seq, (bytes, bytearray)) else np.asarray(seq, dtype=np.uint8)


def prepare_next_crosspoint(mgr, area, c0, c1, alignment_start, must_find=True, goal_location=None):
    """the set-up of find_next_crosspoint: goal, borders and special-rows partition of the sweep c0 -> c1.  Returns
    (special-rows partition, partition for the aligner or None if the manager met the goal without it)."""
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
    part = area.create_partition(c0.i, c0.j, c1.i, c1.j)
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


@sra_mod.with_async_files
def stage2(aligner, seq0, seq1, work, alignment_start=AT_ANYWHERE, sra_limit=0, ident=0, bounds=None, ram_limit=0,
           areas=None):
    """Runs stage 2 for alignment `ident` of work directory `work` (stage 1 must have left crosspoint_01.NN and,
    with sra_limit > 0, its special rows there).  seq0 / seq1: the whole sequences; `bounds` = (i0, j0, i1, j1) the
    part --trim selected for stage 1 (only its origin matters here: where a global alignment must begin).  Returns {"crosspoints": [(type, i, j, score), ...] as written to
    crosspoint_02.NN, "end": the last crosspoint in ORIGINAL coordinates, "partitions", "seconds"}."""
    t_start = time.time()
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
            try:
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
                    cp = find_next_crosspoint(mgr, area2, cp, c1, alignment_start)
                    partitions += 1
                    out.write(cp)
                    if cp.type != TYPE_MATCH:
                        cp.score += GAP_OPEN
            finally:
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
    return {"crosspoints": out.tuples(), "end": cp_r.astuple(), "partitions": partitions, "seconds": time.time() - t_start}

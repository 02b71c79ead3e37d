"""Native Stage-3 driver: splits every partition stage 2 left (one per special row of stage 1) at the special rows
stage 2 saved inside it, then -- orientation flipped -- the pieces at the rows saved by this pass, and so on until the
rows would be closer than 1024: what MASA-Core's stage3() does around the aligner (M/stage3/sw_stage3.cpp:49-122
find_next_crosspoint, :125-208 processPartition, :210-262 reduce_partitions, :264-453 stage3).

Round r reads `crosspoint_03.NN.r<r-1>` (round 1: `crosspoint_02.NN`) and the special rows of the round before, and
writes `crosspoint_03.NN.r<r>` plus `special_rows/stage.03.NN.r<r>/`; odd rounds run in the original orientation, even
rounds in the reversed, transposed one (the same flip stage 2 makes, so that the rows a round saves are the columns the
next round matches against).  The final list, in original coordinates from the alignment's start to its end, goes to
`crosspoint_03.NN` -- stage 4's input (mi355sw_stage4).

Inside a partition the walk is stage 2's: sweep from the running crosspoint to the nearest saved row, match the
dispatched last column against that row, continue from the hit; the goal is the score still to be covered between the
two crosspoints of the partition."""
import time

import numpy as np

from .engine import NEEDLEMAN_WUNSCH
from .manager import AlignerManager, ReversedCellsReader, AT_SEQUENCE_1_OR_2, GAP_OPEN
from .crosspoints import Crosspoint, CrosspointsFile, crosspoint_file, TYPE_MATCH
from .stage2 import prepare_next_crosspoint, conclude_next_crosspoint, MIN_ROW_DISTANCE, _as_u8
from . import sra as sra_mod

MIN_INTERVAL = 1024             # sw_stage3.cpp:327: closer rows are not worth saving, stage 4 takes over
MAX_DEEP = 15


def _walk_partition(mgr, area, prev_part, c0, c1, len_v, len_h, reverse, out):
    """processPartition (:125-208) as a generator: walks partition c0 -> c1 along the rows `prev_part` holds, appending
    one crosspoint per row to `out`.  Every sweep is yielded as (manager, partition for the aligner) -- the caller runs
    it, alone or together with the sweeps of other walks, and resumes the walk.  Scores: forward rounds carry the score
    from the alignment's start, reversed rounds the score still missing to its end; the goal of each sweep is what lies
    between the running crosspoint and c1."""
    c1 = c1.copy()
    cp = c0.copy()
    if reverse:
        cp.score = c0.score - c1.score
    else:
        c1.score += 0 if c1.type == TYPE_MATCH else GAP_OPEN
        cp.score = c1.score - c0.score
    while True:
        if cp.i == c1.i:
            break
        if reverse:
            cp.score += 0 if cp.type == TYPE_MATCH else GAP_OPEN
        cp_r = cp.reverse(len_v, len_h)
        row = prev_part.next_special_row(cp_r.i, cp_r.j, MIN_ROW_DISTANCE)
        if row is None:
            break
        mgr.setLastColumnReader(row)
        col = ReversedCellsReader(prev_part.first_column_reader)
        col.seek(cp_r.i - prev_part.i0 + 1)
        mgr.setLastRowReader(col)
        c_row = Crosspoint(c1.i, len_h - prev_part.get_reading_row(), 0)
        if c_row.j >= c1.j:
            # the rest of the partition: swept without a goal, only for the rows it saves for the next round
            part, adj = prepare_next_crosspoint(mgr, area, cp, c_row, None, must_find=False)
            if adj is not None:
                yield mgr, adj, part
            conclude_next_crosspoint(mgr, area, part, cp, c_row, must_find=False)
            break
        part, adj = prepare_next_crosspoint(mgr, area, cp, c_row, None, goal_location=AT_SEQUENCE_1_OR_2)
        if adj is not None:
            yield mgr, adj, part
        cp = conclude_next_crosspoint(mgr, area, part, cp, c_row)
        if reverse:
            goal_adj = c1.score + cp.score
        else:
            cp.score += 0 if cp.type == TYPE_MATCH else GAP_OPEN
            goal_adj = c1.score - cp.score
        out.append(Crosspoint(cp.i, cp.j, goal_adj, cp.type))


def _run_walks(aligner, walks):
    """drives the walks of a round side by side: the pending sweep of every walk goes to the aligner in ONE call
    (MI355Aligner.alignPartitions: one kernel launch, the partitions' strip chains next to each other), then every walk
    takes its hit and prepares its next sweep.  An aligner without alignPartitions (the CPU double of the tests) gets
    the sweeps one by one -- same calls per walk, same results.  Returns the number of sweeps."""
    pending = {}
    sweeps = 0

    def advance(k):
        try:
            pending[k] = next(walks[k])
        except StopIteration:
            pending.pop(k, None)

    for k in range(len(walks)):
        advance(k)
    batched = hasattr(aligner, "alignPartitions")
    while pending:
        ids = sorted(pending)
        sweeps += len(ids)
        try:
            if batched and len(ids) > 1:
                aligner.alignPartitions([pending[k][1] for k in ids], [pending[k][0] for k in ids])
            else:
                for k in ids:
                    aligner.alignPartition(pending[k][1], pending[k][0])
        except BaseException:
            for k in ids:
                pending[k][2].close()
            raise
        for k in ids:
            advance(k)
    return sweeps


def _reduce_partitions(mgr, prev, out, seq_v, seq_h, area_prev, area, reverse):
    """reduce_partitions (:210-262): `prev` runs from the alignment's start to its end in this round's orientation.
    The partitions between consecutive crosspoints are independent of each other: their walks run side by side
    (_run_walks), the crosspoints are written in the reference's order afterwards."""
    len_v, len_h = len(seq_v), len(seq_h)
    # (an alignment of one crosspoint, or one that runs along a border, spans no cells: nothing to hand to the aligner)
    spans = prev[-1].i > prev[0].i and prev[-1].j > prev[0].j
    if spans:
        mgr.setSequences(seq_v, seq_h, prev[0].i, prev[0].j, prev[-1].i, prev[-1].j)
    found = []                       # per partition: the crosspoints its walk found
    walks = []
    try:
        for c0, c1 in zip(prev, prev[1:]):
            found.append([])
            c0r, c1r = c1.reverse(len_v, len_h), c0.reverse(len_v, len_h)      # the partition's name in the round before
            if c0r.i != c1r.i and c0r.j != c1r.j:
                prev_part = area_prev.open_partition(c0r.i, c0r.j, c1r.i, c1r.j)
                if prev_part.rows_count() > 1:
                    walks.append(_walk_partition(mgr.clone(), area, prev_part, c0, c1, len_v, len_h, reverse, found[-1]))
                else:
                    area.create_partition(c0.i, c0.j, c1.i, c1.j)              # nothing to split at: an empty partition
            # else: a partition crossed by one gap run, nothing to refine
        sweeps = _run_walks(mgr.aligner, walks)
    finally:
        if spans:
            mgr.unsetSequences()
    for c0, pts in zip(prev, found):
        out.write(c0)
        for c in pts:
            out.write(c)
    out.write(prev[-1])
    out.close()
    return sweeps


@sra_mod.with_async_files
def stage3(aligner, seq0, seq1, work, sra_limit=0, ident=0, ram_limit=0, areas=None):
    """Runs stage 3 for alignment `ident`.  Returns {"crosspoints": the final list [(type, i, j, score)] in original
    coordinates (also written to crosspoint_03.NN), "rounds": [(crosspoints in, crosspoints out, sweeps)], "seconds"}."""
    t_start = time.time()
    seq_v, seq_h = np.ascontiguousarray(_as_u8(seq0)), np.ascontiguousarray(_as_u8(seq1))
    m, n = len(seq_v), len(seq_h)
    intervals = sra_mod.flush_intervals(m, n, max(sra_limit, 0) + max(ram_limit, 0))
    mgr = AlignerManager(aligner)
    mgr.setRecurrenceType(NEEDLEMAN_WUNSCH)
    mgr.setBlockPruning(False)
    area_prev = sra_mod.get_area(areas, work, 2, ident, ram_limit=ram_limit, disk_limit=sra_limit)
    prev = CrosspointsFile(crosspoint_file(work, 2, ident)).load()
    if not prev:
        raise RuntimeError("stage 3: no crosspoint_02.%02d in %s" % (ident, work))
    save = True
    rounds = []
    deep = 0
    cur = None
    while deep < MAX_DEEP:
        reverse = deep % 2
        interval = intervals[min(2 + deep, len(intervals) - 1)]
        deep += 1
        prev.reverse_all(len(seq_h), len(seq_v))          # :352: into this round's orientation, start -> end
        cur = CrosspointsFile(crosspoint_file(work, 3, ident, deep)).open()
        area = sra_mod.get_area(areas, work, 3, ident, deep, ram_limit=ram_limit, disk_limit=sra_limit)
        if interval < MIN_INTERVAL:
            mgr.setSpecialRowInterval(MIN_INTERVAL)
            save = False
        else:
            mgr.setSpecialRowInterval(interval)
        if not save:
            area.set_persistent(False)
        sweeps = _reduce_partitions(mgr, prev, cur, seq_v, seq_h, area_prev, area, reverse)
        rounds.append((len(prev), len(cur), sweeps))
        if len(prev) == len(cur):
            break
        if area.rows_count() <= area.partitions_count():  # nothing but first rows saved: nothing left to split at
            break
        prev, area_prev = cur, area                      # Job::getSpecialRowsArea hands the same object back (:352-353)
        seq_v, seq_h = np.ascontiguousarray(seq_h[::-1]), np.ascontiguousarray(seq_v[::-1])
    if deep % 2 == 0:
        cur.reverse_all(len(seq_v), len(seq_h))
    final = CrosspointsFile(crosspoint_file(work, 3, ident))
    final.extend(c.copy() for c in cur)
    final.save()
    return {"crosspoints": final.tuples(), "rounds": rounds, "seconds": time.time() - t_start}

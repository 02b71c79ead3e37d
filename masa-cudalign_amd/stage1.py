"""Native Stage-1 driver: what MASA-Core's stage1() does around the aligner (M/stage1/sw_stage1.cpp:203-240,
:244-493), for the repo's own engine -- recurrence from the alignment edges, flush interval from the area limit
(Job::calculateFlushIntervals), special rows + last row written to the Special Rows Area on disk in the
reference's layout (sra.py), status file, crosspoint file, and RESUME: a run that was killed continues from the
last complete special row (SpecialRowsPartition::continueFromLastRow) and ends with the same files and the same
best score as an uninterrupted one."""
import os
import time

from .engine import Partition, F_NO_DIAGONAL_SEED
from .manager import (Stage1Manager, ArrayCellsReader, InitialCellsReader, AT_ANYWHERE, AT_SEQUENCE_1, AT_SEQUENCE_2,
                      AT_SEQUENCE_1_OR_2, GAP_OPEN, GAP_EXT)
from . import sra as sra_mod


def border_readers(alignment_start):
    """getBorderCells, sw_stage1.cpp:137-161"""
    if alignment_start in (AT_ANYWHERE, AT_SEQUENCE_1_OR_2):
        return InitialCellsReader(), InitialCellsReader()
    if alignment_start == AT_SEQUENCE_1:
        return InitialCellsReader(), InitialCellsReader(GAP_OPEN, GAP_EXT)
    if alignment_start == AT_SEQUENCE_2:
        return InitialCellsReader(GAP_OPEN, GAP_EXT), InitialCellsReader()
    return InitialCellsReader(GAP_OPEN, GAP_EXT), InitialCellsReader(GAP_OPEN, GAP_EXT)


def progress_line(seconds, best, progress_string):
    """the line MASA-Core prints every two seconds (logStatus, sw_stage1.cpp:120-127)"""
    t = int(seconds)
    return "(%dh%02dm%02ds) best:(%d,%d,%d) %s" % (t // 3600, (t % 3600) // 60, t % 60, best[0], best[1], best[2], progress_string)


def _start_progress_log(aligner, mgr, stream, interval, t0):
    """a timer thread next to the blocking alignPartition call (the reference's RecurrentTimer); the engine's progress
    string is read with one atomic load (mi355sw_progress), the manager's best score list is only looked at"""
    if stream is None or not hasattr(aligner, "getProgressString"):
        return lambda: None
    import threading
    done = threading.Event()

    def loop():
        while not done.wait(interval):
            try:
                stream.write(progress_line(time.time() - t0, mgr.getBestScore(), aligner.getProgressString()) + "\n")
                stream.flush()
            except Exception:          # a logger must never take the run down
                return
    th = threading.Thread(target=loop, daemon=True)
    th.start()

    def stop():
        done.set()
        th.join(timeout=5.0)
    return stop


def _crosspoints_on_disk(work):
    """the end points a finished stage 1 left (crosspoint_01.00, .01, ...)"""
    out, k = [], 0
    while os.path.exists(sra_mod.crosspoint_path(work, 1, k)):
        tok = open(sra_mod.crosspoint_path(work, 1, k)).read().split()
        if len(tok) < 3:
            break
        _t, i, j, sc = (int(x) for x in tok[1].split(","))
        out.append((i, j, sc))
        k += 1
    return out


@sra_mod.with_async_files
def stage1(aligner, seq0, seq1, work, alignment_start=AT_ANYWHERE, alignment_end=AT_ANYWHERE, sra_limit=0,
           block_pruning=True, manager_class=Stage1Manager, bounds=None, progress=None, progress_interval=2.0,
           max_alignments=1, ram_limit=0, areas=None, prune_global=False):
    """Runs (or resumes) stage 1 of seq0 x seq1 in work directory `work`.  `prune_global`: block pruning for a GLOBAL alignment
    too (both ends in the corners) -- beyond the reference, whose stage 1 prunes local alignments only (sw_stage1.cpp:219-225);
    the adapter's --prune-global.  `aligner` is an MI355Aligner; create it
    with a fixed strip height (rows_per_lane) when the area must be resumable or shared with CUDAlign: special rows
    sit on multiples of the strip height (1024 or 2048 rows give CUDAlign's 8192-row spacing).
    `bounds` = (i0, j0, i1, j1): the part of the matrix --trim selects (Sequence::getTrimStart()-1 .. getTrimEnd(),
    sw_stage1.cpp:281-284); seq0 / seq1 are always the WHOLE sequences, every coordinate (special-row directory,
    crosspoints, status) stays absolute, and the flush interval is computed from the whole sizes (Job.cpp:62-67).
    `progress`: a text stream (sys.stderr) that gets MASA-Core's progress line every `progress_interval` seconds while
    the aligner runs -- "(0h00m02s) best:(i,j,score) PROGRESS: d/D strips" (logStatus, sw_stage1.cpp:112-128; the status
    file itself is saved with every completed special row, not by this timer).
    `sra_limit` is the DISK budget of the special rows (--disk-size), `ram_limit` the part kept in memory instead
    (--ram-size): rows alternate between the two in proportion, the spacing follows their sum; rows in memory are only
    there for later stages that get the same `areas` dict (pipeline.py does), and a resumed run continues from the last
    row on disk.
    `max_alignments` > 1 (--max-alignments): the best-score list keeps that many end points of different alignments
    (BestScoreList), one crosspoint_01.NN each; candidates are what the aligner dispatches (one best cell per strip).
    Returns {"best": (i, j, score) in 1-based DP coordinates, "bests": the whole list, "resumed_from": row or None, "seconds", "gcups", ...}."""
    m, n = len(seq0), len(seq1)
    bi0, bj0, bi1, bj1 = bounds if bounds is not None else (0, 0, m, n)
    if not (0 <= bi0 < bi1 <= m and 0 <= bj0 < bj1 <= n):
        raise ValueError("stage1: bounds %r outside the %d x %d matrix" % (bounds, m, n))
    os.makedirs(work, exist_ok=True)
    status = sra_mod.Status(work)
    budget = max(sra_limit, 0) + max(ram_limit, 0)                      # Job::getSRALimit (Job.cpp:354-364)
    interval = sra_mod.flush_interval(m, n, budget) if budget > 0 else 0
    fr, fc = border_readers(alignment_start)
    i0, resumed_from, part_sra = bi0, None, None
    area = sra_mod.get_area(areas, work, 1, 0, ram_limit=ram_limit, disk_limit=sra_limit)
    if budget > 0:
        part_sra = area.create_partition(bi0, bj0, bi1, bj1)
        last = part_sra.last_disk_row_id()
        if last == bi1 and status.loaded:
            # "Stage 1 was already executed" (sw_stage1.cpp:212-214)
            return {"best": status.best, "bests": _crosspoints_on_disk(work), "resumed_from": bi1, "seconds": 0.0, "gcups": 0.0, "already_done": True,
                    "special_rows": [r for r in part_sra.rows]}
        if last != bi0:
            # The status file is saved AFTER the row it belongs to has been renamed into place (two file writes): a kill
            # in between leaves a row on disk whose best-score list was never saved.  The run continues from the row the
            # status file knows (the rows below it are computed again and replace the files), and refuses to continue
            # rows that have no status at all -- their best cells would be lost silently.
            if not status.loaded:
                raise RuntimeError("stage1: %s holds special rows but no status file: cannot resume (clean the work directory)" % work)
            i0 = min(last, max(status.last_special_row, bi0))
            if i0 != bi0:
                row = part_sra.read_row(i0)
                fc.read(None, i0 - bi0)                # firstColumnReader->read(NULL, lastRowId): rows since the border
                fr = ArrayCellsReader(row)             # FileCellsReader(lastRowFilename): cell 0 = the corner of the rest
                resumed_from = i0
            else:
                part_sra.set_border_markers(fr.getType(), 0, fc.getType(), 0)
        else:
            part_sra.set_border_markers(fr.getType(), 0, fc.getType(), 0)
    else:
        # no budget for special rows: the partition's directory and border markers still exist, as MASA-Core's
        # (SpecialRowsArea::createSplittedPartitions) -- stage 2 then walks back over ONE partition, to its first row
        area.create_partition(bi0, bj0, bi1, bj1).set_border_markers(fr.getType(), 0, fc.getType(), 0)
    part = Partition(i0, bj0, bi1, bj1)
    sup = Partition(bi0, bj0, bi1, bj1)
    v0, v1 = seq0[bi0:bi1], seq1[bj0:bj1]       # AlignerManager::setSequences (:168-176): the aligner sees the trimmed data
    rel = Partition(i0 - bi0, 0, bi1 - bi0, bj1 - bj0)
    mgr = manager_class(part, alignment_start=alignment_start, alignment_end=alignment_end,
                        special_row_interval=interval, first_row_reader=fr, first_column_reader=fc,
                        super_partition=sup, block_pruning=block_pruning, prune_global=prune_global,
                        sra_partition=part_sra, status=status, seq0_offset=bi0, seq1_offset=bj0,
                        **({"max_alignments": max_alignments} if max_alignments != 1 else {}))
    if status.loaded and status.best is not None and status.best[0] >= 0:
        mgr.best_list.add(*status.best)            # Status::load -> bestScoreList->add (Status.cpp:60-64)
    status.stage = 1
    # the best strip VALUE a two-phase run left (status.mi355) only counts for the run that continues THAT partition
    key = (bi0, bj0, bi1, bj1, int(alignment_start), int(alignment_end))
    if resumed_from is None or status.value_key != key:
        status.value_best = None
    status.value_key = key
    prefix_value = status.value_best            # left by the run(s) this one continues: strips above row i0
    aligner.setSequences(v0, v1)
    t0 = time.time()
    stop_log = _start_progress_log(aligner, mgr, progress, progress_interval, t0)
    # --max-alignments > 1: the weaker alignments are candidates too, and a pruning bound that starts from the score of the
    # best one (the engine's diagonal seed pass) removes them sooner than a bound that grows with the sweep, as the
    # reference's does -- the seed is left out then (include/mi355sw.h: MI355SW_F_NO_DIAGONAL_SEED)
    seed_off = max_alignments != 1 and hasattr(aligner, "setFlag") and not (aligner.getFlags() & F_NO_DIAGONAL_SEED)
    if seed_off:
        aligner.setFlag(F_NO_DIAGONAL_SEED, True)
    try:
        aligner.alignPartition(rel, mgr)
    finally:
        if seed_off:
            aligner.setFlag(F_NO_DIAGONAL_SEED, False, defer=True)      # (handed over with the next call: nothing here may raise)
        stop_log()
        if part_sra is not None:
            part_sra.close()
        aligner.unsetSequences()
    dt = time.time() - t0
    best = mgr.getBestScore()
    # Two-phase tracking + resume: strips above the row this run continued from may only be known by VALUE (the run
    # that computed them died before it could locate its best cell).  If that value beats -- or ties, the smaller row
    # wins -- everything with a known position, its cell is located now: one exact pass from the special row above
    # that strip down to the strip's last row, a 1/(number of special rows) slice of the matrix.
    # (this run's own strips need none of that: the engine located their best cell itself before it returned)
    located = None
    vb = prefix_value
    if vb is not None and part_sra is not None and (vb[0] > best[2] or (vb[0] == best[2] and vb[1] < best[0])):
        above = [part_sra.i0 + r for r in part_sra.rows if part_sra.i0 + r <= vb[1]]
        r0 = max(above) if above else bi0
        fr2, fc2 = border_readers(alignment_start)
        if r0 > bi0:
            fc2.read(None, r0 - bi0)
            fr2 = ArrayCellsReader(part_sra.read_row(r0))
        sub = Partition(r0, bj0, min(vb[2], bi1), bj1)
        mgr2 = manager_class(sub, alignment_start=alignment_start, alignment_end=alignment_end,
                             first_row_reader=fr2, first_column_reader=fc2, super_partition=sup,
                             seq0_offset=bi0, seq1_offset=bj0)
        aligner.setSequences(v0, v1)
        try:
            aligner.alignPartition(Partition(sub.i0 - bi0, 0, sub.i1 - bi0, bj1 - bj0), mgr2)
        finally:
            aligner.unsetSequences()
        located = tuple(mgr2.getBestScore())
        if located[2] != vb[0]:
            raise RuntimeError("stage1 resume: rows [%d,%d) were recorded with best value %d, the exact pass finds %d"
                               % (vb[1], vb[2], vb[0], located[2]))
        mgr.best_list.add(*located)
        best = mgr.getBestScore()
    status.stage = 2
    if part_sra is not None:
        status.last_special_row = part_sra.last_row_id()
    status.drop_value_best()
    status.save(best)
    bests = mgr.best_list.all() if hasattr(mgr.best_list, "all") else ([tuple(best)] if best[0] >= 0 else [])
    for k, b in enumerate(bests):                   # sw_stage1.cpp:481-492: one file per entry of the list
        sra_mod.write_crosspoint(sra_mod.crosspoint_path(work, 1, k), b)
    # else: an empty best-score list -- MASA-Core writes no crosspoint file and runs no traceback (sw_stage1.cpp:481-492: one file per entry of the list)
    st = aligner.getStatistics()
    return {"best": tuple(best), "bests": [tuple(b) for b in bests], "resumed_from": resumed_from, "seconds": dt,
            "gcups": float(bi1 - i0) * (bj1 - bj0) / dt / 1e9 if dt > 0 else 0.0, "strip_rows": st["strip_rows"],
            "kernel_ms": st["kernel_ms"], "pruned_cells": st["pruned_cells"], "located_from_value": located,
            "special_rows": list(part_sra.rows) if part_sra is not None else []}

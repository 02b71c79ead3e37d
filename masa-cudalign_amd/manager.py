"""Python restatement of the caller side of the boundary: what MASA-Core's AlignerManager does
with the rows, columns and scores a Stage-1 aligner dispatches (M/common/AlignerManager.cpp),
BestScoreList (M/common/BestScoreList.cpp) and InitialCellsReader (M/common/io/InitialCellsReader.cpp).
Used by the repo's own driver, tests/ and bench.py; it holds no DP arithmetic.
"""
import numpy as np

from .engine import (INF, NEEDLEMAN_WUNSCH, SMITH_WATERMAN, INIT_WITH_ZEROES, INIT_WITH_GAPS,
                     INIT_WITH_CUSTOM_DATA, INIT_WITH_GAPS_OPENED, Partition)

# M/common/Job.hpp: alignment edge flags
AT_NOWHERE, AT_ANYWHERE, AT_SEQUENCE_1, AT_SEQUENCE_2, AT_SEQUENCE_1_OR_2, AT_SEQUENCE_1_AND_2 = -1, 0, 1, 2, 3, 4

GAP_OPEN, GAP_EXT = 3, 2


class InitialCellsReader:
    """M/common/io/InitialCellsReader.cpp:30-108."""

    def __init__(self, gap_open=0, gap_ext=0, start_offset=0):
        self.gap_open, self.gap_ext = gap_open, gap_ext
        if gap_open == 0 and gap_ext == 0:
            self.type = INIT_WITH_ZEROES
        elif gap_open == 0:
            self.type = INIT_WITH_GAPS_OPENED
        else:
            self.type = INIT_WITH_GAPS
        self.start_offset = start_offset
        self.position = start_offset

    def getType(self):
        return self.type

    def read(self, buf, length):
        if buf is None:                 # skip `length` cells (SpecialRowsPartition::continueFromLastRow reads into NULL)
            self.position += length
            return length
        if self.type == INIT_WITH_ZEROES:
            buf[:length, 0] = 0
            buf[:length, 1] = -INF
        else:
            pos = self.position + np.arange(length, dtype=np.int64)
            h = -self.gap_ext * pos - self.gap_open
            h[pos == 0] = 0
            buf[:length, 0] = h.astype(np.int32)
            buf[:length, 1] = -INF
        self.position += length
        return length


class ArrayCellsReader:
    """Custom data border (INIT_WITH_CUSTOM_DATA): a cell array incl. the corner, read sequentially."""

    def __init__(self, cells):
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.position = 0

    def getType(self):
        return INIT_WITH_CUSTOM_DATA

    def read(self, buf, length):
        if buf is not None:
            buf[:length] = self.cells[self.position:self.position + length]
        self.position += length
        return length


class BestScoreList:
    """limit-1 behaviour of M/common/BestScoreList.cpp:129-195 with the order of BestScoreList.hpp:30-38:
    score desc, then i asc, then j asc; scores below min_score are ignored."""

    def __init__(self, min_score):
        self.min_score = min_score
        self.best = None

    def add(self, i, j, score):
        if score < self.min_score:
            return
        c = (-score, i, j)
        if self.best is None or c < (-self.best[2], self.best[0], self.best[1]):
            self.best = (i, j, score)

    def getBestScore(self):
        return self.best if self.best is not None else (-1, -1, -INF)


def initial_best_score(alignment_start, alignment_end):
    """getInitialBestScore, M/stage1/sw_stage1.cpp:68-107."""
    s, e = alignment_start, alignment_end
    s00 = s in (AT_ANYWHERE, AT_SEQUENCE_1, AT_SEQUENCE_2, AT_SEQUENCE_1_OR_2, AT_SEQUENCE_1_AND_2)
    s01 = s in (AT_ANYWHERE, AT_SEQUENCE_1, AT_SEQUENCE_1_OR_2)
    s10 = s in (AT_ANYWHERE, AT_SEQUENCE_2, AT_SEQUENCE_1_OR_2)
    s11 = s == AT_ANYWHERE
    e00 = e == AT_ANYWHERE
    e01 = e in (AT_ANYWHERE, AT_SEQUENCE_2, AT_SEQUENCE_1_OR_2)
    e10 = e in (AT_ANYWHERE, AT_SEQUENCE_1, AT_SEQUENCE_1_OR_2)
    e11 = e in (AT_ANYWHERE, AT_SEQUENCE_1, AT_SEQUENCE_2, AT_SEQUENCE_1_OR_2, AT_SEQUENCE_1_AND_2)
    return 0 if ((s00 and e00) or (s01 and e01) or (s10 and e10) or (s11 and e11)) else -INF


class Stage1Manager:
    """What stage1() + AlignerManager set up around one partition (M/stage1/sw_stage1.cpp:244-493,
    M/common/AlignerManager.cpp).  Collects special rows / last row / last column in memory."""

    def __init__(self, partition, alignment_start=AT_ANYWHERE, alignment_end=AT_ANYWHERE,
                 special_row_interval=0, keep_last_row=False, keep_last_column=False,
                 first_row_reader=None, first_column_reader=None, seq0_offset=0, seq1_offset=0,
                 super_partition=None, block_pruning=False, sra_partition=None, status=None):
        self.partition = partition
        self.super_partition = super_partition or partition
        self.seq0_offset, self.seq1_offset = seq0_offset, seq1_offset
        # sw_stage1.cpp:318-322
        self.recurrence = SMITH_WATERMAN if alignment_start == AT_ANYWHERE else NEEDLEMAN_WUNSCH
        # sw_stage1.cpp:137-161 getBorderCells
        if first_row_reader is None or first_column_reader is None:
            if alignment_start in (AT_ANYWHERE, AT_SEQUENCE_1_OR_2):
                fr, fc = InitialCellsReader(start_offset=seq1_offset), InitialCellsReader(start_offset=seq0_offset)
            elif alignment_start == AT_SEQUENCE_1:
                fr, fc = InitialCellsReader(start_offset=seq1_offset), InitialCellsReader(GAP_OPEN, GAP_EXT, seq0_offset)
            elif alignment_start == AT_SEQUENCE_2:
                fr, fc = InitialCellsReader(GAP_OPEN, GAP_EXT, seq1_offset), InitialCellsReader(start_offset=seq0_offset)
            else:
                fr = InitialCellsReader(GAP_OPEN, GAP_EXT, seq1_offset)
                fc = InitialCellsReader(GAP_OPEN, GAP_EXT, seq0_offset)
            first_row_reader = first_row_reader or fr
            first_column_reader = first_column_reader or fc
        self.first_row_reader, self.first_column_reader = first_row_reader, first_column_reader
        self.best_location = alignment_end
        self.best_list = BestScoreList(initial_best_score(alignment_start, alignment_end))
        self.special_row_interval = special_row_interval
        self.keep_last_row, self.keep_last_column = keep_last_row, keep_last_column
        # sw_stage1.cpp:219-225: pruning only when the alignment may end anywhere
        self.block_pruning = block_pruning and alignment_end == AT_ANYWHERE
        # special rows / last row go to disk when a SpecialRowsPartition is given (AlignerManager::dispatchRow ->
        # SpecialRowsPartition::write, AlignerManager.cpp:334-356); the status file follows every completed row
        self.sra, self.status = sra_partition, status
        self.value_best = None      # (score, row_lo, row_hi): best strip VALUE of a two-phase run (dispatchStripValue)
        self.active = True
        self.special_rows = {}      # dp row -> list of chunks
        self.last_row_chunks, self.last_column_chunks = [], []
        self.last_row_pos = self.last_column_pos = 0
        self.calls = []

    # --- IManager getters (IManager.hpp:98-140) ---
    def getRecurrenceType(self):
        return self.recurrence

    def getSpecialRowInterval(self):
        return self.special_row_interval

    def getFirstColumnInitType(self):
        return self.first_column_reader.getType()

    def getFirstRowInitType(self):
        return self.first_row_reader.getType()

    def getSuperPartition(self):
        p = self.super_partition
        return Partition(p.i0 - self.seq0_offset, p.j0 - self.seq1_offset, p.i1 - self.seq0_offset, p.j1 - self.seq1_offset)

    # --- streams (AlignerManager.cpp:318-332) ---
    def receiveFirstRow(self, buf, length):
        self.first_row_reader.read(buf, length)

    def receiveFirstColumn(self, buf, length):
        self.first_column_reader.read(buf, length)

    # --- sinks (AlignerManager.cpp:334-450) ---
    def dispatchColumn(self, j, buf, length):
        j += self.seq1_offset
        if j == self.partition.j1:
            if self.keep_last_column:
                self.last_column_chunks.append(np.array(buf[:length], copy=True))
            if self.best_location in (AT_SEQUENCE_2, AT_SEQUENCE_1_OR_2):
                k = int(np.argmax(buf[:length, 0]))
                self.best_list.add(self.partition.i0 + self.last_column_pos + k, self.partition.j1, int(buf[k, 0]))
            self.last_column_pos += length

    def dispatchRow(self, i, buf, length):
        i += self.seq0_offset
        if self.sra is not None:
            if self.sra.write(i, buf[:length]) and self.status is not None:
                self.status.last_special_row = i
                self.status.merge_value_best(self.value_best)
                self.status.save(self.best_list.best)
        elif self.mustDispatchSpecialRows():
            self.special_rows.setdefault(i, []).append(np.array(buf[:length], copy=True))
        if i == self.partition.i1:
            if self.keep_last_row:
                self.last_row_chunks.append(np.array(buf[:length], copy=True))
            if self.best_location in (AT_SEQUENCE_1, AT_SEQUENCE_1_OR_2):
                k = int(np.argmax(buf[:length, 0]))
                self.best_list.add(self.partition.i1, self.partition.j0 + self.last_row_pos + k, int(buf[k, 0]))
            self.last_row_pos += length

    def dispatchScore(self, score, bx=-1, by=-1):
        i, j, s = score
        i += self.seq0_offset + 1      # AlignerManager.cpp:412-415: cell index -> 1-based DP coordinate
        j += self.seq1_offset + 1
        if s > -INF:
            if self.best_location == AT_ANYWHERE:
                self.best_list.add(i, j, s)
            elif self.best_location == AT_SEQUENCE_1_AND_2:
                if i == self.partition.i1 and j == self.partition.j1:
                    self.best_list.add(i, j, s)

    def dispatchStripValue(self, row_lo, row_hi, score):
        """optional engine call (include/mi355sw.h): rows [row_lo, row_hi) hold a cell of `score`, position still
        unknown (two-phase tracking).  Only the first strip reaching the maximum matters (canonical order: min i)."""
        row_lo += self.seq0_offset
        row_hi += self.seq0_offset
        if self.best_location == AT_ANYWHERE and score >= self.best_list.min_score:
            if self.value_best is None or score > self.value_best[0]:
                self.value_best = (int(score), int(row_lo), int(row_hi))

    # --- must* (AlignerManager.cpp:455-503) ---
    def mustContinue(self):
        return self.active

    def mustDispatchLastCell(self):
        return self.best_location == AT_SEQUENCE_1_AND_2

    def mustDispatchLastRow(self):
        # a special-rows partition always takes the last row too (SpecialRowsPartition hands out a last-row writer)
        return self.keep_last_row or self.sra is not None or self.best_location in (AT_SEQUENCE_1, AT_SEQUENCE_1_OR_2)

    def mustDispatchLastColumn(self):
        return self.keep_last_column or self.best_location in (AT_SEQUENCE_2, AT_SEQUENCE_1_OR_2)

    def mustDispatchSpecialRows(self):
        return self.special_row_interval > 0

    def mustDispatchScores(self):
        return self.best_location == AT_ANYWHERE

    def mustPruneBlocks(self):
        return self.block_pruning

    # --- results ---
    def getBestScore(self):
        return self.best_list.getBestScore()

    def specialRow(self, i):
        return np.concatenate(self.special_rows[i], axis=0)

    def lastRow(self):
        return np.concatenate(self.last_row_chunks, axis=0)

    def lastColumn(self):
        return np.concatenate(self.last_column_chunks, axis=0)

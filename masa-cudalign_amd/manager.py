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

    def getStartOffset(self):
        return self.start_offset

    def seek(self, position):           # InitialCellsReader.cpp:57-59: relative to the start offset
        self.position = self.start_offset + position

    def getOffset(self):
        return self.position - self.start_offset

    def clone(self, offset):            # :65-71
        return InitialCellsReader(self.gap_open, self.gap_ext, self.start_offset + offset)

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

    def seek(self, position):
        self.position = position

    def getOffset(self):
        return self.position

    def read(self, buf, length):
        if buf is not None:
            buf[:length] = self.cells[self.position:self.position + length]
        self.position += length
        return length


class FileCellsReader:
    """M/common/io/FileCellsReader.cpp: a border kept in a file of 8-byte cells (C00000000.INIT_WITH_CUSTOM_DATA)"""

    def __init__(self, filename):
        self.filename = filename
        self.position = 0

    def getType(self):
        return INIT_WITH_CUSTOM_DATA

    def seek(self, position):
        self.position = position

    def getOffset(self):
        return self.position

    def read(self, buf, length):
        if buf is not None:
            a = np.fromfile(self.filename, dtype=np.int32, count=2 * length, offset=8 * self.position).reshape(-1, 2)
            if a.shape[0] != length:
                raise RuntimeError("%s: %d cells at %d, file ends after %d" % (self.filename, length, self.position, a.shape[0]))
            buf[:length] = a
        self.position += length
        return length


class ReversedCellsReader:
    """M/common/io/ReversedCellsReader.cpp:46-72: reads a seekable border backwards from the position it was seeked to;
    every read returns the cells in reversed order"""

    def __init__(self, reader):
        self.reader = reader
        self.position = 0

    def getType(self):
        return INIT_WITH_CUSTOM_DATA

    def seek(self, position):
        self.position = position

    def getOffset(self):
        return self.position

    def read(self, buf, length):
        length = min(length, self.position)
        self.position -= length
        self.reader.seek(self.position)
        if buf is not None and length > 0:
            tmp = np.empty((length, 2), dtype=np.int32)
            self.reader.read(tmp, length)
            buf[:length] = tmp[::-1]
        return length


class BestScoreList:
    """M/common/BestScoreList.cpp:45-195 with the order of BestScoreList.hpp:30-38 (score desc, then i asc, then j asc).
    Up to `limit` end points of DIFFERENT alignments (--max-alignments): a candidate that lies in the shadow of a better
    one -- reachable from it by a diagonal run and gaps whose expected cost explains the score difference (isDerived,
    :67-107) -- or that scores under a quarter of the best (isAllowed, :109-117) is not kept, and a new entry evicts the
    entries it shadows.  With limit 1 this is the canonical best cell: (max score, min i, min j).
    seq0_len / seq1_len: the extent of the matrix stage 1 sweeps (sw_stage1.cpp:285-286, :330)."""

    MATCH, MISMATCH, GAP_EXT_ = 1, -3, 2

    def __init__(self, min_score, limit=1, seq0_len=0, seq1_len=0):
        self.min_score, self.limit = min_score, max(int(limit), 1)
        self.seq0_len, self.seq1_len = seq0_len, seq1_len
        self.entries = []               # (i, j, score), best first

    @staticmethod
    def _key(e):
        return (-e[2], e[0], e[1])

    @property
    def best(self):
        return self.entries[0] if self.entries else None

    def _derived(self, best, cand):
        """isDerived(best, cand): is `cand` just another cell of the alignment that ends in `best`?"""
        diff_z, diff_x, diff_y = cand[2] - best[2], cand[1] - best[1], cand[0] - best[0]
        if diff_z >= 0:
            return False
        ax, ay = abs(diff_x), abs(diff_y)
        if self.min_score >= 0 and best[2] > (ax + ay) // self.GAP_EXT_:
            return True
        if self.min_score < 0 and ax + ay > min(self.seq0_len, self.seq1_len):
            return True
        same_side = (diff_x <= 0 and diff_y <= 0) or (diff_x >= 0 and diff_y >= 0)
        gaps = abs(ax - ay) if same_side else ax + ay
        diagonal = min(ax, ay)
        f32 = np.float32                                     # the reference computes these bounds in single precision
        z_min = int((f32(self.MISMATCH) * (f32(1) - f32(0.25)) + f32(self.MATCH) * f32(0.25)) * f32(diagonal))
        z_max = self.MATCH * diagonal
        g_diff = self.GAP_EXT_ * gaps
        return bool(f32(diff_z) >= f32(z_min) - f32(g_diff) * f32(2.0) and f32(diff_z) <= f32(z_max) - f32(g_diff) / f32(2.0))

    @staticmethod
    def _allowed(best, cand):
        if best[2] > 0 and cand[2] < 0.25 * best[2]:
            return False
        if best[2] < 0 and cand[2] < 1.10 * best[2]:
            return False
        return True

    def add(self, i, j, score):
        """_add (:129-195)"""
        if score < self.min_score:
            return
        reg = (int(i), int(j), int(score))
        e = self.entries
        if len(e) == self.limit and e[-1][2] > reg[2]:
            return
        if e and not self._allowed(e[0], reg):
            return
        if reg in e:
            return
        if any(self._derived(it, reg) for it in e):
            return
        e[:] = [it for it in e if not (self._derived(reg, it) or not self._allowed(reg, it))]
        e.append(reg)
        e.sort(key=self._key)
        if len(e) > self.limit:
            e.pop()

    def getBestScore(self):
        return self.entries[0] if self.entries else (-1, -1, -INF)

    def all(self):
        return list(self.entries)


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
                 super_partition=None, block_pruning=False, sra_partition=None, status=None, max_alignments=1, prune_global=True):
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
        sup = super_partition or partition
        self.best_list = BestScoreList(initial_best_score(alignment_start, alignment_end), max_alignments,
                                       sup.i1 - sup.i0, sup.j1 - sup.j0)
        self.special_row_interval = special_row_interval
        self.keep_last_row, self.keep_last_column = keep_last_row, keep_last_column
        # sw_stage1.cpp:219-225: pruning only when the alignment may end anywhere -- and, beyond the reference's stage 1
        # (which holds the bound, AbstractBlockPruning.cpp:92-109, but never asks for it), for global alignments: both
        # ends in the corners, the goal is the last cell
        # (prune_global=False: the reference's own rule only -- what stage1.py / pipeline.py / tools/align_fasta.py pass unless
        #  asked otherwise, like the adapter's --prune-global: a pruned global stage 1 leaves lower bounds in its special rows,
        #  a work directory that is no longer value-compatible with one the reference wrote)
        self.block_pruning = block_pruning and (alignment_end == AT_ANYWHERE or
                                                (prune_global and alignment_start == AT_SEQUENCE_1_AND_2 and alignment_end == AT_SEQUENCE_1_AND_2))
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
        if bx != -1 and by != -1:      # :418-423: what --dump-blocks stores (BlocksFile::setScore)
            if getattr(self, "block_scores", None) is None:
                self.block_scores = {}
            self.block_scores[(bx, by)] = (i, j, s)
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
        # the native driver always saves the partition's last row next to the special rows: it is what marks stage 1
        # as complete for a later run (sw_stage1.cpp:210-214 tests getLastRowId() == i1; MASA-Core itself only has that
        # row when the last block row happens to be a special one).  Stage 2 never reads it: it lies below every
        # crosspoint (SpecialRowsPartition::nextSpecialRow wants a row ABOVE)
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


START_TYPE_MATCH, START_TYPE_GAP_H, START_TYPE_GAP_V = 0, 1, 2       # IManager.hpp:52-58 (= the crosspoint types)
MATCH_ALIGNED, MATCH_GAPPED = 0, 1                                    # libmasaTypes.hpp:66-72


class BacktraceLost(RuntimeError):
    """the reference prints "Backtrace lost" and exits: a border sum above the goal, or a partition swept to its end
    without meeting the goal"""


class AlignerManager:
    """MASA-Core's AlignerManager as the traceback stages use it (M/common/AlignerManager.cpp): one object that lives
    through a stage, gets sequences, borders, a goal score and the special row / first column to match it against,
    runs partitions on the aligner and reports where the optimal alignment leaves each of them.

    The aligner sees coordinates relative to the sub-sequences handed to setSequences(); everything the stages see is
    absolute (seq offsets added back, :336, :375, :414-415)."""

    def __init__(self, aligner):
        self.aligner = aligner
        self.recurrence = SMITH_WATERMAN
        self.block_pruning = False
        self.special_row_interval = 0
        self.sra = None
        self.seq0_offset = self.seq1_offset = 0
        self.first_row_reader = self.first_column_reader = None
        self.last_column_reader = self.last_row_reader = None
        self.last_column_writer = self.last_row_writer = None
        self.goal_score, self.goal_location = -INF, AT_NOWHERE
        self.best_list, self.best_location = None, AT_NOWHERE
        self.super_partition = None
        self.partition = None
        self.start_type = START_TYPE_MATCH
        self.found = False
        self.next_crosspoint = None          # (i, j, score, type)
        self.active = False
        self.last_column_pos = self.last_row_pos = 0

    # -- configuration (:191-316) ------------------------------------------------------------------------------
    def setSequences(self, seq0, seq1, i0, j0, i1, j1):
        self.seq0_offset, self.seq1_offset = i0, j0
        self.aligner.setSequences(seq0[i0:i1], seq1[j0:j1])

    def unsetSequences(self):
        self.aligner.unsetSequences()

    def setRecurrenceType(self, r):
        self.recurrence = r

    def setBlockPruning(self, b):
        self.block_pruning = b

    def setSpecialRowInterval(self, n):
        self.special_row_interval = int(n)

    def setSpecialRowsPartition(self, p):
        self.sra = p

    def setLastColumnReader(self, r):
        self.last_column_reader = r

    def setLastRowReader(self, r):
        self.last_row_reader = r

    def setGoalScore(self, score, location):
        if score == -INF or location == AT_NOWHERE:
            self.unsetGoalScore()
        else:
            self.goal_score, self.goal_location = score, location

    def unsetGoalScore(self):
        self.goal_score, self.goal_location = -INF, AT_NOWHERE

    # -- one partition (:89-166) -------------------------------------------------------------------------------
    def prepareAlign(self, partition, start_type):
        """everything of alignPartition (:89-166) up to the call of the aligner: returns the partition to hand to the
        aligner (sequence-relative), or None when there is nothing for it to do (no cells, or the goal is met by a
        border-long gap).  alignPartition = prepareAlign + aligner.alignPartition; stage 3 collects the prepared
        partitions of many walks and hands them over together (MI355Aligner.alignPartitions)."""
        self.partition, self.start_type, self.found = partition, start_type, False
        if partition.getWidth() == 0 or partition.getHeight() == 0:
            return None
        self.last_column_pos = self.last_row_pos = 0
        self.active = True
        if self.sra is not None:
            self.first_row_reader, self.first_column_reader = self.sra.first_row_reader, self.sra.first_column_reader
            self.last_column_writer, self.last_row_writer = self.sra.last_column_writer, self.sra.last_row_writer
        if self.goal_location in (AT_SEQUENCE_2, AT_SEQUENCE_1_OR_2):
            r = self._find_full_gap(partition.getWidth(), start_type != START_TYPE_GAP_H, self.last_column_reader)
            if r is not None:
                self.next_crosspoint = (partition.i0, partition.j1, r, 1)
                self.active = False
        if self.goal_location in (AT_SEQUENCE_1, AT_SEQUENCE_1_OR_2):
            r = self._find_full_gap(partition.getHeight(), start_type != START_TYPE_GAP_V, self.last_row_reader)
            if r is not None:
                self.next_crosspoint = (partition.i1, partition.j0, r, 2)
                self.active = False
        if not self.active:
            return None
        return Partition(partition.i0 - self.seq0_offset, partition.j0 - self.seq1_offset,
                         partition.i1 - self.seq0_offset, partition.j1 - self.seq1_offset)

    def alignPartition(self, partition, start_type):
        adj = self.prepareAlign(partition, start_type)
        if adj is not None:
            self.aligner.alignPartition(adj, self)

    def clone(self):
        """another manager on the same aligner and sequences, with state of its own: one per walk that runs side by side"""
        c = AlignerManager(self.aligner)
        c.recurrence, c.block_pruning, c.special_row_interval = self.recurrence, self.block_pruning, self.special_row_interval
        c.seq0_offset, c.seq1_offset = self.seq0_offset, self.seq1_offset
        return c

    def isFoundCrosspoint(self):
        return self.found

    def getNextCrosspoint(self):
        return self.next_crosspoint if self.found else (-1, -1, -INF, 0)

    # -- goal matching (:625-718) ------------------------------------------------------------------------------
    def _find_goal_cell(self, buf, length, reader):
        """first cell k of the dispatched border chunk whose forward value (from `reader`: the special row / first
        column of the stage before, read backwards) and reverse value (buf) add up to the goal"""
        if self.found or reader is None:
            return None
        base = np.empty((length, 2), dtype=np.int32)
        got = reader.read(base, length)
        r = self.aligner.matchLastColumn(buf[:got], base[:got], self.goal_score)
        if r["found"]:
            self.found = True
            return r
        if r["type"] < 0:
            self.active = False          # inside an engine callback the exception only surfaces after the call: stop the sweep
            raise BacktraceLost("backtrace lost (%d): border sum above the goal %d at cell %d of a chunk of %d"
                                % (r["type"], self.goal_score, r["k"], length))
        return None

    def _find_full_gap(self, length, open_gap, reader):
        """(:659-676) the partition crossed by one gap run: the border cell the run comes from, its gap component plus
        the run, equals the goal"""
        if self.found or reader is None:
            return None
        first_f = -length * GAP_EXT - (GAP_OPEN if open_gap else 0)
        cell = np.empty((1, 2), dtype=np.int32)
        off = reader.getOffset()
        reader.read(cell, 1)
        reader.seek(off)
        if int(cell[0, 1]) + first_f + GAP_OPEN == self.goal_score:
            self.found = True
            return int(cell[0, 1])
        return None

    # -- IManager (what the aligner calls) ---------------------------------------------------------------------
    def getRecurrenceType(self):
        return self.recurrence

    def getSpecialRowInterval(self):
        return self.special_row_interval

    def getFirstColumnInitType(self):
        return self.first_column_reader.getType()

    def getFirstRowInitType(self):
        return self.first_row_reader.getType()

    def getSuperPartition(self):
        p = self.super_partition or self.partition
        return Partition(p.i0 - self.seq0_offset, p.j0 - self.seq1_offset, p.i1 - self.seq0_offset, p.j1 - self.seq1_offset)

    def receiveFirstRow(self, buf, length):
        self.first_row_reader.read(buf, length)

    def receiveFirstColumn(self, buf, length):
        self.first_column_reader.read(buf, length)

    def dispatchColumn(self, j, buf, length):
        """(:334-371)"""
        j += self.seq1_offset
        p = self.partition
        if j != p.j1:
            return
        if self.last_column_writer is not None:
            self.last_column_writer.write(buf[:length])
        if self.best_list is not None and self.best_location in (AT_SEQUENCE_2, AT_SEQUENCE_1_OR_2):
            k = int(np.argmax(buf[:length, 0]))
            self.best_list.add(p.i0 + self.last_column_pos + k, p.j1, int(buf[k, 0]))
        if self.goal_location in (AT_ANYWHERE, AT_SEQUENCE_2, AT_SEQUENCE_1_OR_2):
            r = self._find_goal_cell(buf, length, self.last_column_reader)
            if r is not None:
                self.next_crosspoint = (p.i0 + self.last_column_pos + r["k"], p.j1, r["score"],
                                        0 if r["type"] == MATCH_ALIGNED else 1)
                self.active = False
        self.last_column_pos += length

    def dispatchRow(self, i, buf, length):
        """(:376-407)"""
        i += self.seq0_offset
        p = self.partition
        if self.mustDispatchSpecialRows():
            self.sra.write(i, buf[:length])
        if i != p.i1:
            return
        if self.last_row_writer is not None:
            self.last_row_writer.write(buf[:length])
        if self.best_list is not None and self.best_location in (AT_SEQUENCE_1, AT_SEQUENCE_1_OR_2):
            k = int(np.argmax(buf[:length, 0]))
            self.best_list.add(p.i1, p.j0 + self.last_row_pos + k, int(buf[k, 0]))
        if self.goal_location in (AT_ANYWHERE, AT_SEQUENCE_1, AT_SEQUENCE_1_OR_2):
            r = self._find_goal_cell(buf, length, self.last_row_reader)
            if r is not None:
                self.next_crosspoint = (p.i1, p.j0 + self.last_row_pos + r["k"], r["score"],
                                        0 if r["type"] == MATCH_ALIGNED else 2)
                self.active = False
        self.last_row_pos += length

    def dispatchScore(self, score, bx=-1, by=-1):
        """(:412-448): with the goal AT_ANYWHERE the alignment may START inside this partition -- the cell whose
        reverse value is the whole goal.  No test of `found` here: a later block reporting the goal again replaces
        an earlier hit, as in the reference."""
        i, j, s = score
        i += self.seq0_offset + 1
        j += self.seq1_offset + 1
        if s <= -INF:
            return
        if self.best_list is not None:
            if self.best_location == AT_ANYWHERE:
                self.best_list.add(i, j, s)
            elif self.best_location == AT_SEQUENCE_1_AND_2 and i == self.partition.i1 and j == self.partition.j1:
                self.best_list.add(i, j, s)
        if self.goal_location == AT_ANYWHERE and s == self.goal_score:
            self.next_crosspoint = (i, j, 0, 0)
            self.found = True
            self.active = False

    def mustContinue(self):
        return self.active

    def mustDispatchLastCell(self):
        return self.best_location == AT_SEQUENCE_1_AND_2

    def mustDispatchLastRow(self):
        return (self.last_row_writer is not None or self.best_location in (AT_SEQUENCE_1, AT_SEQUENCE_1_OR_2) or
                (self.goal_location in (AT_ANYWHERE, AT_SEQUENCE_1, AT_SEQUENCE_1_OR_2) and self.last_row_reader is not None))

    def mustDispatchLastColumn(self):
        return (self.last_column_writer is not None or self.best_location in (AT_SEQUENCE_2, AT_SEQUENCE_1_OR_2) or
                (self.goal_location in (AT_ANYWHERE, AT_SEQUENCE_2, AT_SEQUENCE_1_OR_2) and self.last_column_reader is not None))

    def mustDispatchSpecialRows(self):
        return self.sra is not None and self.sra.persistent and self.special_row_interval > 0

    def mustDispatchScores(self):
        return self.best_location == AT_ANYWHERE or self.goal_location == AT_ANYWHERE

    def mustPruneBlocks(self):
        return self.block_pruning

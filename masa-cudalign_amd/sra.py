"""Special Rows Area on disk + status file: the native driver's counterpart of MASA-Core's
M/common/sra/SpecialRowsPartition.cpp, SpecialRowFile.cpp, SpecialRowsArea.cpp and M/common/Status.cpp.

The on-disk layout is the reference's, so that an area written here can be read by MASA-Core's
stages 2-6 and vice versa:

  <work>/special_rows/stage.SS.II/<i0>.<j0>.<i1>.<j1>/<rowid>        all numbers %08X; rowid = i - i0
      (j1-j0+1) cell_t of 8 bytes: cell 0 = first-column cell of the row with f = -INF, cells 1.. = columns j0+1..j1
  .../C<offset>.<INIT_TYPE>, R<offset>.<INIT_TYPE>                    zero-length markers: how the borders were made
  <work>/status                                                        "stage\\nlast special row\\ni j score\\n"
  <work>/crosspoints/crosspoint_01.NN                                  "START\\n0,i,j,score\\nEND\\n"

A row is written as <rowid>.tmp and renamed when its last cell has arrived (SpecialRowFile.cpp:93-102, :146-152): a
killed run leaves only complete rows (plus .tmp files that the next run deletes), and the last complete row is the
checkpoint stage 1 continues from (SpecialRowsPartition::continueFromLastRow, :454-462; sw_stage1.cpp:210-217).
"""
import os

import numpy as np

from .engine import INF, INIT_WITH_ZEROES, INIT_WITH_GAPS, INIT_WITH_CUSTOM_DATA, INIT_WITH_GAPS_OPENED

INIT_NAMES = {INIT_WITH_ZEROES: "INIT_WITH_ZEROES", INIT_WITH_GAPS: "INIT_WITH_GAPS",
              INIT_WITH_CUSTOM_DATA: "INIT_WITH_CUSTOM_DATA", INIT_WITH_GAPS_OPENED: "INIT_WITH_GAPS_OPENED"}
CELL_BYTES = 8


def special_rows_path(work, stage=1, ident=0):
    """Job::getSpecialRowsPath, M/common/Job.cpp:202-211"""
    p = os.path.join(work, "special_rows", "stage.%02d.%02d" % (stage, ident))
    os.makedirs(p, exist_ok=True)
    return p


def crosspoint_path(work, stage=1, ident=0):
    p = os.path.join(work, "crosspoints")
    os.makedirs(p, exist_ok=True)
    return os.path.join(p, "crosspoint_%02d.%02d" % (stage, ident))


def write_crosspoint(path, best, typ=0):
    """CrosspointsFile.cpp:99-150 (stage-1 form: one crosspoint)"""
    with open(path, "w") as f:
        f.write("START\n%d,%d,%d,%d\nEND\n" % (typ, best[0], best[1], best[2]))


class _OpenRow:
    def __init__(self, path, rowid, width_cells):
        self.final = os.path.join(path, "%08X" % rowid)
        self.tmp = self.final + ".tmp"
        self.f = open(self.tmp, "wb")
        self.f.truncate(width_cells * CELL_BYTES)      # SpecialRowFile::initialize: the file has its final size at once
        self.offset = 0

    def write(self, cells):
        a = np.ascontiguousarray(cells, dtype=np.int32)
        self.f.write(a.tobytes())
        self.offset += a.shape[0]

    def close(self):
        self.f.close()
        os.replace(self.tmp, self.final)


class SpecialRowsPartition:
    """One partition directory of the area (SpecialRowsPartition.cpp).  Coordinates as in the reference: rows i0..i1,
    columns j0..j1 of the DP matrix, row i0 / column j0 being the border."""

    def __init__(self, area_path, i0, j0, i1, j1):
        self.i0, self.j0, self.i1, self.j1 = i0, j0, i1, j1
        self.path = os.path.join(area_path, "%08X.%08X.%08X.%08X" % (i0, j0, i1, j1))
        os.makedirs(self.path, exist_ok=True)
        self._open = {}
        self.rows = []                 # ids (i - i0) of complete rows, ascending
        self.read_directory()

    # -- directory ---------------------------------------------------------------------------------------------
    def read_directory(self):
        """SpecialRowsPartition::readDirectory (:343-381): complete rows are 8 hex digits; leftovers of a killed run
        (<id>.tmp) are removed (SpecialRowFile.cpp:40-47)"""
        rows = []
        for fn in os.listdir(self.path):
            if len(fn) == 12 and fn.endswith(".tmp"):
                os.remove(os.path.join(self.path, fn))
            elif len(fn) == 8 and all(c in "0123456789ABCDEF" for c in fn):
                rows.append(int(fn, 16))
        self.rows = sorted(rows)

    @property
    def width_cells(self):
        return self.j1 - self.j0 + 1

    def last_row_id(self):
        """absolute DP row of the last complete row, i0 when there is none (getLastRowId, :186-188)"""
        return self.i0 + (self.rows[-1] if self.rows else 0)

    def row_filename(self, i):
        return os.path.join(self.path, "%08X" % (i - self.i0))

    def read_row(self, i):
        return np.fromfile(self.row_filename(i), dtype=np.int32).reshape(-1, 2)

    # -- border markers (setBorderReader, :125-175) ------------------------------------------------------------
    def set_border_markers(self, first_row_type, first_row_offset, first_col_type, first_col_offset):
        for prefix, typ, off in (("C", first_col_type, first_col_offset), ("R", first_row_type, first_row_offset)):
            if typ == INIT_WITH_CUSTOM_DATA:
                continue               # custom borders are tee'd into C00000000.INIT_WITH_CUSTOM_DATA by their reader
            open(os.path.join(self.path, "%s%08X.%s" % (prefix, off, INIT_NAMES[typ])), "wb").close()

    # -- writing (write, :335-353) -----------------------------------------------------------------------------
    def write(self, i, cells):
        """append cells to row i; returns True when the row became complete (closed + renamed)"""
        rid = i - self.i0
        row = self._open.get(rid)
        if row is None:
            row = self._open[rid] = _OpenRow(self.path, rid, self.width_cells)
        row.write(cells)
        if row.offset >= self.width_cells:
            row.close()
            del self._open[rid]
            if rid not in self.rows:
                self.rows.append(rid)
                self.rows.sort()
            return True
        return False

    def close(self):
        for row in self._open.values():
            row.f.close()              # incomplete rows stay .tmp: the next read_directory() removes them
        self._open = {}

    # -- resume (continueFromLastRow, :454-462) ----------------------------------------------------------------
    def continue_from_last_row(self):
        """(row to continue from, its cells): the last complete row becomes the first row of the rest of the
        partition; the caller advances its first-column reader by (row - i0) cells"""
        i = self.last_row_id()
        return i, self.read_row(i)


class Status:
    """M/common/Status.cpp:40-89: stage, last special row, best score; saved through a temporary file + rename.
    Next to it (own file `status.mi355`, MASA-Core never looks at it): the best strip VALUE of a two-phase run whose
    cell has not been located yet -- "score row_lo row_hi" -- see stage1.py."""

    def __init__(self, work):
        self.file = os.path.join(work, "status")
        self.tmp = self.file + ".tmp"
        self.side = os.path.join(work, "status.mi355")
        self.stage, self.last_special_row, self.best = 1, 0, None
        self.value_best = None
        self.loaded = False
        if os.path.exists(self.side):
            tok = open(self.side).read().split()
            if len(tok) == 3:
                self.value_best = (int(tok[0]), int(tok[1]), int(tok[2]))
        if os.path.exists(self.file):
            tok = open(self.file).read().split()
            if len(tok) >= 2:
                self.stage, self.last_special_row = int(tok[0]), int(tok[1])
                if len(tok) >= 5:
                    self.best = (int(tok[2]), int(tok[3]), int(tok[4]))
                self.loaded = True

    def merge_value_best(self, cand):
        """keep the FIRST strip that reaches the largest value (canonical order: smallest i wins ties)"""
        if cand is not None and (self.value_best is None or cand[0] > self.value_best[0] or
                                 (cand[0] == self.value_best[0] and cand[1] < self.value_best[1])):
            self.value_best = tuple(int(x) for x in cand)

    def save(self, best=None):
        if self.value_best is not None:
            with open(self.side + ".tmp", "w") as f:
                f.write("%d %d %d\n" % self.value_best)
            os.replace(self.side + ".tmp", self.side)
        if best is not None:
            self.best = tuple(int(x) for x in best)
        b = self.best if self.best is not None else (-1, -1, -INF)
        with open(self.tmp, "w") as f:
            f.write("%d\n%d\n%d %d %d\n" % (self.stage, self.last_special_row, b[0], b[1], b[2]))
        os.replace(self.tmp, self.file)


def flush_interval(m, n, limit):
    """Job::calculateFlushIntervals, M/common/Job.cpp:231-241 (first interval): rows between two special rows for an
    area of `limit` bytes; at least two rows fit"""
    if limit <= 0:
        return 0
    if limit < n * CELL_BYTES * 2:
        limit = n * CELL_BYTES * 2
    return int(m * n * CELL_BYTES // limit + 1)

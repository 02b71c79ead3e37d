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
import atexit
import collections
import contextlib
import functools
import os
import threading

import numpy as np

from .engine import INF, INIT_WITH_ZEROES, INIT_WITH_GAPS, INIT_WITH_CUSTOM_DATA, INIT_WITH_GAPS_OPENED

INIT_NAMES = {INIT_WITH_ZEROES: "INIT_WITH_ZEROES", INIT_WITH_GAPS: "INIT_WITH_GAPS",
              INIT_WITH_CUSTOM_DATA: "INIT_WITH_CUSTOM_DATA", INIT_WITH_GAPS_OPENED: "INIT_WITH_GAPS_OPENED"}
CELL_BYTES = 8


class _FileQueue:
    """The file operations of an area -- create, write, close + rename, remove, truncate, rename of a directory, status
    file -- carried out by ONE thread in the order they were asked for, while a stage runs (`async_files()`).  A special
    row is tens to hundreds of megabytes handed over inside an engine callback; written inline, the page-cache copy is time
    the aligner's thread does not spend on the running kernel (C3, stage 2: 29 GB, 3.9 of its 15 s).  Order is what the
    on-disk protocol needs -- a row is renamed into place before the status file that names it -- and one thread keeps
    it.  Readers wait: for one partition's operations (`wait(owner)`) or for everything (`drain()`).  Outside
    `async_files()` every operation runs at once, in the caller's thread."""

    MAX_BYTES = 1 << 30             # cells waiting to be written (the submitter waits above that)

    def __init__(self):
        self.cv = threading.Condition()
        self.q = collections.deque()
        self.depth = 0              # nesting of async_files()
        self.bytes = 0
        self.running = None         # owner of the operation being carried out
        self.pending = {}           # id(owner) -> operations not finished
        self.error = None
        self.thread = None
        self.pid = None

    def _enabled(self):
        return self.depth > 0 and not os.environ.get("MI355SW_SRA_SYNC")

    def _after_fork(self):
        """a child process inherits a snapshot of the parent's queue and a condition variable in whatever state it was: the
        parent carries its operations out, the child starts empty"""
        if self.pid is not None and self.pid != os.getpid():
            self.cv = threading.Condition()
            self.q = collections.deque()
            self.bytes = 0
            self.running = None
            self.pending = {}
            self.error = None
            self.thread = None
            self.pid = None

    def _check_alive(self):
        """operations are pending and nobody is left to carry them out (the thread died: an exception outside an operation,
        interpreter shutdown): say so instead of returning as if the files were in place"""
        if self.pending and (self.thread is None or not self.thread.is_alive()):
            n = sum(self.pending.values())
            self.q.clear()
            self.pending.clear()
            self.bytes = 0
            raise RuntimeError("special rows area: %d queued file operations were never carried out (the file thread is gone)" % n)

    def submit(self, owner, fn, *args, nbytes=0):
        self._after_fork()
        if not self._enabled():
            self.drain()
            fn(*args)
            return
        with self.cv:
            if self.error is not None:
                self._raise()
            if self.thread is None or self.pid != os.getpid() or not self.thread.is_alive():
                self.pid = os.getpid()
                self.thread = threading.Thread(target=self._run, name="mi355sw-sra-files", daemon=True)
                self.thread.start()
            while self.bytes > self.MAX_BYTES and self.error is None:
                self.cv.wait()
            self.q.append((owner, fn, args, nbytes))
            self.bytes += nbytes
            self.pending[id(owner)] = self.pending.get(id(owner), 0) + 1
            self.cv.notify_all()

    def _run(self):
        while True:
            with self.cv:
                while not self.q:
                    self.cv.wait()
                owner, fn, args, nbytes = self.q.popleft()
            try:
                if self.error is None:          # after a failure nothing more touches the files
                    fn(*args)
            except BaseException as e:          # noqa: BLE001 -- handed to the thread that asked
                with self.cv:
                    self.error = e
            with self.cv:
                self.bytes -= nbytes
                k = self.pending.get(id(owner), 1) - 1
                if k > 0:
                    self.pending[id(owner)] = k
                else:
                    self.pending.pop(id(owner), None)
                self.cv.notify_all()

    def _raise(self):
        e, self.error = self.error, None
        self.q.clear()
        self.pending.clear()
        self.bytes = 0
        raise RuntimeError("special rows area: a queued file operation failed: %r" % (e,)) from e

    def wait(self, owner):
        """until every operation asked for on behalf of `owner` has been carried out"""
        self._after_fork()
        if self.pid != os.getpid():
            return
        with self.cv:
            while self.pending.get(id(owner), 0) > 0 and self.error is None and self.thread is not None and self.thread.is_alive():
                self.cv.wait(1.0)
            if self.error is not None:
                self._raise()
            if self.pending.get(id(owner), 0) > 0:
                self._check_alive()

    def drain(self):
        self._after_fork()
        if self.pid != os.getpid():
            return
        with self.cv:
            while self.pending and self.error is None and self.thread is not None and self.thread.is_alive():
                self.cv.wait(1.0)
            if self.error is not None:
                self._raise()
            self._check_alive()

    def has_pending(self):
        return bool(self.pending)


_files = _FileQueue()
atexit.register(lambda: _files.drain())


def drain():
    """every queued file operation has been carried out when this returns"""
    _files.drain()


def queue_file_operation(owner, fn, *args, nbytes=0):
    """fn(*args) as one more file operation of the running stages: carried out by the file thread, after everything asked for
    before it, inside async_files(); at once otherwise.  For files that nothing in the same run reads back (the pipeline's
    crosspoint_04: 130 MB of text at C3's size, formatted while stages 5 and 6 run in the library)."""
    _files.submit(owner, fn, *args, nbytes=nbytes)


@contextlib.contextmanager
def async_files():
    """while the block runs, the areas' file operations are queued (see _FileQueue); all carried out when it ends"""
    _files.depth += 1
    try:
        yield
    finally:
        _files.depth -= 1
        if _files.depth == 0:
            _files.drain()


def with_async_files(fn):
    """decorator: the function's body runs inside async_files()"""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        with async_files():
            return fn(*args, **kwargs)
    return wrapper


def special_rows_path(work, stage=1, ident=0, deep=-1):
    """Job::getSpecialRowsPath, M/common/Job.cpp:202-211"""
    name = "stage.%02d.%02d" % (stage, ident) if deep <= -1 else "stage.%02d.%02d.r%02d" % (stage, ident, deep)
    p = os.path.join(work, "special_rows", name)
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
    def __init__(self, path, rowid, width_cells, owner=None):
        self.final = os.path.join(path, "%08X" % rowid)
        self.tmp = self.final + ".tmp"
        self.owner = owner
        self.f = None
        self.offset = 0
        _files.submit(owner, self._create, width_cells)

    def _create(self, width_cells):
        self.f = open(self.tmp, "wb", buffering=0)
        self.f.truncate(width_cells * CELL_BYTES)      # SpecialRowFile::initialize: the file has its final size at once

    def _write(self, data):
        view = memoryview(data)
        while len(view):                               # a raw file may take less than it was given
            view = view[self.f.write(view):]

    def write(self, cells):
        a = np.ascontiguousarray(cells, dtype=np.int32)
        if _files._enabled():
            data = a.tobytes()                         # the caller's buffer is only lent for the call
            _files.submit(self.owner, self._write, data, nbytes=len(data))
        else:
            _files.drain()
            self._write(a.reshape(-1).view(np.uint8))
        self.offset += a.shape[0]

    def _close(self, rename):
        self.f.close()
        if rename:
            os.replace(self.tmp, self.final)

    def close(self, rename=True):
        _files.submit(self.owner, self._close, rename)


class _OpenRamRow:
    """SpecialRowRAM.cpp: a row kept in memory for the life of the area object (the same process's later stages)"""

    def __init__(self, rowid, width_cells):
        self.id = rowid
        self.cells = np.empty((width_cells, 2), dtype=np.int32)
        self.offset = 0

    def write(self, cells):
        a = np.asarray(cells, dtype=np.int32).reshape(-1, 2)
        self.cells[self.offset:self.offset + a.shape[0]] = a
        self.offset += a.shape[0]


class SpecialRowReader:
    """One special row opened for the traceback (SpecialRow.cpp:90-147): seeked to a cell count, it is read BACKWARDS
    from there, every read returning its cells in reversed order -- the order in which the next stage, which sweeps
    the reversed matrix, meets them.  Row id 0 is the partition's first row and comes from its border reader
    (FirstRow.cpp:84-102)."""

    def __init__(self, partition, rid):
        self.partition, self.id = partition, rid
        self.offset = 0

    def getType(self):
        return INIT_WITH_CUSTOM_DATA

    def seek(self, offset):
        self.offset = offset

    def getOffset(self):
        return self.offset

    def _cells(self, offset, length):
        if self.id == 0:
            reader = self.partition.first_row_reader
            tmp = np.empty((length, 2), dtype=np.int32)
            reader.seek(offset)
            reader.read(tmp, length)
            return tmp
        ram = self.partition._ram.get(self.id)
        if ram is not None:
            return ram[offset:offset + length]
        _files.wait(self.partition)
        fn = os.path.join(self.partition.path, "%08X" % self.id)
        a = np.fromfile(fn, dtype=np.int32, count=2 * length, offset=CELL_BYTES * offset).reshape(-1, 2)
        if a.shape[0] != length:
            raise RuntimeError("end of special row %s: %d cells at %d, file ends after %d" % (fn, length, offset, a.shape[0]))
        return a

    def read(self, buf, length):
        if self.offset == 0:
            raise RuntimeError("special row overflow: %d cells asked from row %08X at its start" % (length, self.id))
        length = min(length, self.offset)
        self.offset -= length
        if buf is not None:
            buf[:length] = self._cells(self.offset, length)[::-1]
        return length


def _shorten(fn, size):
    if size < os.path.getsize(fn):
        os.truncate(fn, size)


SCRATCH_PREFIX = "guess."             # partitions of sweeps nobody has accepted yet (SpecialRowsArea.create_partition(scratch=True))


def _remove_tree(path):
    """a discarded partition's directory; what cannot be removed is said (a stale directory under a partition's own name would
    be opened by a later stage: scratch partitions therefore never carry such a name, see SCRATCH_PREFIX)"""
    import shutil
    import sys

    def complain(fn, p, exc):
        sys.stderr.write("[sra] could not remove %s: %s\n" % (p, exc[1] if isinstance(exc, tuple) else exc))
    if os.path.isdir(path):
        shutil.rmtree(path, onerror=complain)


def _move_directory(old, new):
    if os.path.isdir(new):             # left by an earlier run of the same stage in this work directory
        import shutil
        shutil.rmtree(new)
    os.rename(old, new)


class SpecialRowsPartition:
    """One partition directory of the area (SpecialRowsPartition.cpp).  Coordinates as in the reference: rows i0..i1,
    columns j0..j1 of the DP matrix, row i0 / column j0 being the border.  `persistent=False` is the reference's
    partition without a path (SpecialRowsArea::getPartitionPath returns "" for non-persistent areas): it takes the
    border readers and drops every row."""

    def __init__(self, area_path, i0, j0, i1, j1, read_only=False, persistent=True, scratch=False):
        self.i0, self.j0, self.i1, self.j1 = i0, j0, i1, j1
        self.read_only, self.persistent = read_only, persistent
        self.first_row_reader = self.first_column_reader = None
        self.last_row_writer = self.last_column_writer = None
        self._open = {}
        self._ram = {}                 # id -> cells of the complete rows kept in memory
        # RAM / disk proportion of the rows written (setRamProportion, :268-271): SpecialRowsArea.create_partition sets
        # the area's budgets; a partition made on its own keeps its rows on disk
        self.ram_proportion, self.disk_proportion = 0, 1
        self.ram_count = self.disk_count = 0
        self.rows = []                 # ids (i - i0) of complete rows, ascending (the first row, id 0, is implicit)
        # id -> (largest H of the row, its cell index): kept while the rows are written when stage 2 is going to guess its
        # crosspoints from them (stage2.py, MI355SW_STAGE2_SPECULATE); lives with the object, i.e. for the stages of one process
        self.track_peaks = os.environ.get("MI355SW_STAGE2_SPECULATE", "1") not in ("", "0")
        self.peaks = {}
        self.reading = None            # SpecialRowReader handed out last
        self._reading_idx = 0
        self.largest_interval = 0
        if not persistent:
            self.path = ""
            return
        self.path = os.path.join(area_path, (SCRATCH_PREFIX if scratch else "") + "%08X.%08X.%08X.%08X" % (i0, j0, i1, j1))
        if read_only:
            _files.drain()
            if not os.path.isdir(self.path):
                raise RuntimeError("special rows partition %s does not exist" % self.path)
        else:
            # (a queued rename of a directory may target this very path -- the same stage run again in one work directory:
            #  _move_directory would find the directory made here and remove it before renaming over it)
            if _files.has_pending():
                _files.drain()
            os.makedirs(os.path.dirname(self.path), exist_ok=True)
            try:
                os.mkdir(self.path)
                return                 # a directory made just now holds nothing to read (and nothing queued can touch it)
            except FileExistsError:
                pass
        self.read_directory()

    # -- directory ---------------------------------------------------------------------------------------------
    def read_directory(self):
        """SpecialRowsPartition::readDirectory (:343-381): complete rows are 8 hex digits; leftovers of a killed run
        (<id>.tmp) are removed (SpecialRowFile.cpp:40-47)"""
        _files.drain()
        rows = []
        for fn in os.listdir(self.path):
            if len(fn) == 12 and fn.endswith(".tmp"):
                os.remove(os.path.join(self.path, fn))
            elif len(fn) == 8 and all(c in "0123456789ABCDEF" for c in fn):
                rows.append(int(fn, 16))
            elif fn[:1] in ("C", "R") and len(fn) > 10 and fn[9] == ".":
                self._load_border_reader(fn)
        self.rows = sorted(set(rows) | set(self._ram))
        self.reload()

    def set_ram_proportion(self, ram, disk):
        self.ram_proportion, self.disk_proportion = ram, disk

    def _next_row_on_disk(self):
        """getSpecialRow (:316-333): rows alternate between disk and memory in the proportion of the two budgets"""
        if (self.disk_proportion != 0 and self.ram_proportion == 0) or \
                self.ram_count * self.disk_proportion > self.ram_proportion * self.disk_count:
            self.disk_count += 1
            return True
        self.ram_count += 1
        return False

    def _load_border_reader(self, fn):
        """loadBorderReader (:490-515): how a border was made is in the marker's name"""
        from .manager import InitialCellsReader, FileCellsReader, GAP_OPEN, GAP_EXT
        offset, typ = int(fn[1:9], 16), fn[10:]
        if typ == "INIT_WITH_CUSTOM_DATA":
            reader = FileCellsReader(os.path.join(self.path, fn))
        elif typ == "INIT_WITH_ZEROES":
            reader = InitialCellsReader(start_offset=offset)
        elif typ == "INIT_WITH_GAPS":
            reader = InitialCellsReader(GAP_OPEN + offset * GAP_EXT, GAP_EXT)
        elif typ == "INIT_WITH_GAPS_OPENED":
            reader = InitialCellsReader(offset * GAP_EXT, GAP_EXT)
        else:
            return
        if fn[0] == "C":
            self.first_column_reader = reader
        else:
            self.first_row_reader = reader

    def reload(self):
        """(:177-184): the traceback reads the rows from the last one upwards"""
        self.rows.sort()
        self._reading_idx = len(self.rows)          # index into [first row] + rows
        self.reading = None
        ids = [0] + self.rows
        self.largest_interval = max([b - a for a, b in zip(ids, ids[1:])] or [0])

    def rows_count(self):
        """getRowsCount (:272-274): the first row counts"""
        return len(self.rows) + 1

    def get_reading_row(self):
        """(:198-200) absolute DP row of the special row handed out last"""
        return self.i0 + self.reading.id

    def next_special_row(self, i, j, min_dist):
        """nextSpecialRow (:383-429): the nearest row more than `min_dist` rows above DP row i (the first row when none
        is), opened and seeked so that its first read starts at column j and walks towards j0.  None when (i, j)
        already sits on the first row."""
        ids = [0] + self.rows
        while self._reading_idx >= 0:
            dist = (i - self.i0) - ids[self._reading_idx]
            if self._reading_idx == 0:
                if dist > 0:
                    break
                return None
            if dist > min_dist:
                break
            self._reading_idx -= 1
        self.reading = SpecialRowReader(self, ids[self._reading_idx])
        self.reading.seek(abs(j - self.j0) + 1)
        return self.reading

    # -- border readers as the stage that creates the partition sets them (:111-175) ---------------------------
    def set_first_row_reader(self, reader):
        self.first_row_reader = reader
        self._border_marker("R", reader)

    def set_first_column_reader(self, reader):
        self.first_column_reader = reader
        self._border_marker("C", reader)

    def _border_marker(self, prefix, reader):
        if reader is None or not self.persistent:
            return
        typ = reader.getType()
        if typ == INIT_WITH_CUSTOM_DATA:
            raise NotImplementedError("custom-data borders of a traceback partition (TeeCellsReader) are not used by "
                                      "stages 2-3 and not built")
        open(os.path.join(self.path, "%s%08X.%s" % (prefix, reader.getStartOffset(), INIT_NAMES[typ])), "wb").close()

    # -- truncation once the crosspoint is known (:202-236) ----------------------------------------------------
    def truncate(self, max_i, max_j):
        """rows at or below DP row max_i go away, the others keep columns j0..max_j"""
        if self.persistent:
            for rid, row in list(self._open.items()):
                if rid + self.i0 < max_i and row.offset < max_j - self.j0 + 1:
                    raise RuntimeError("special row %08X of %s kept by the crosspoint (%d,%d) holds %d of %d cells"
                                       % (rid, self.path, max_i, max_j, row.offset, max_j - self.j0 + 1))
                if isinstance(row, _OpenRamRow):
                    self._ram[rid] = row.cells
                else:
                    row.close()                         # SpecialRowFile::close renames whatever was written
                if rid not in self.rows:
                    self.rows.append(rid)
            self._open = {}
            keep, cells = [], max_j - self.j0 + 1
            for rid in sorted(self.rows):
                if rid in self._ram:                    # SpecialRowRAM::truncateRow does nothing; dropped rows are freed
                    if rid + self.i0 >= max_i:
                        del self._ram[rid]
                    else:
                        keep.append(rid)
                    continue
                fn = os.path.join(self.path, "%08X" % rid)
                if rid + self.i0 >= max_i:
                    _files.submit(self, os.remove, fn)
                else:
                    _files.submit(self, _shorten, fn, cells * CELL_BYTES)
                    keep.append(rid)
            self.rows = keep
        self.i1, self.j1 = max_i, max_j
        self.reload()

    def change_path(self, new_path):
        if self.persistent and new_path != self.path:
            _files.submit(self, _move_directory, self.path, new_path)
            self.path = new_path

    @property
    def width_cells(self):
        return self.j1 - self.j0 + 1

    def last_row_id(self):
        """absolute DP row of the last complete row, i0 when there is none (getLastRowId, :186-188)"""
        return self.i0 + (self.rows[-1] if self.rows else 0)

    def row_filename(self, i):
        return os.path.join(self.path, "%08X" % (i - self.i0))

    def read_row(self, i):
        if (i - self.i0) in self._ram:
            return self._ram[i - self.i0]
        _files.wait(self)
        return np.fromfile(self.row_filename(i), dtype=np.int32).reshape(-1, 2)

    def row_peak(self, rid, max_index):
        """(H, cell index) of the largest H among cells 1..max_index of complete row `rid` (cell c = DP column j0 + c; the first
        of several equal ones) -- from the record kept while the row was written when that lies in range, from the row itself
        otherwise; None for the first row and for an empty range"""
        if rid == 0 or max_index < 1:
            return None
        p = self.peaks.get(rid)
        if p is not None and 1 <= p[1] <= max_index:
            return p
        cells = self.read_row(self.i0 + rid)[1:max_index + 1]
        if len(cells) == 0:
            return None
        k = int(np.argmax(cells[:, 0]))
        return int(cells[k, 0]), k + 1

    def last_disk_row_id(self):
        """absolute DP row of the last complete row ON DISK (what a later process can continue from)"""
        disk = [r for r in self.rows if r not in self._ram]
        return self.i0 + (disk[-1] if disk else 0)

    # -- border markers (setBorderReader, :125-175) ------------------------------------------------------------
    def set_border_markers(self, first_row_type, first_row_offset, first_col_type, first_col_offset):
        for prefix, typ, off in (("C", first_col_type, first_col_offset), ("R", first_row_type, first_row_offset)):
            if typ == INIT_WITH_CUSTOM_DATA:
                continue               # custom borders are tee'd into C00000000.INIT_WITH_CUSTOM_DATA by their reader
            open(os.path.join(self.path, "%s%08X.%s" % (prefix, off, INIT_NAMES[typ])), "wb").close()

    # -- writing (write, :335-353) -----------------------------------------------------------------------------
    def write(self, i, cells):
        """append cells to row i; returns True when the row became complete (closed + renamed)"""
        if self.read_only:
            raise RuntimeError("writing into a read-only special rows partition")
        if not self.persistent:
            return False
        rid = i - self.i0
        row = self._open.get(rid)
        if row is None:
            row = self._open[rid] = (_OpenRow(self.path, rid, self.width_cells, self) if self._next_row_on_disk()
                                     else _OpenRamRow(rid, self.width_cells))
        if self.track_peaks and len(cells):
            h = np.asarray(cells)[:, 0]
            k = int(np.argmax(h))
            if rid not in self.peaks or int(h[k]) > self.peaks[rid][0]:
                self.peaks[rid] = (int(h[k]), row.offset + k)
        row.write(cells)
        if row.offset >= self.width_cells:
            if isinstance(row, _OpenRamRow):
                self._ram[rid] = row.cells
            else:
                row.close()
            del self._open[rid]
            if rid not in self.rows:
                self.rows.append(rid)
                self.rows.sort()
            return True
        return False

    def close(self):
        for row in self._open.values():
            if not isinstance(row, _OpenRamRow):
                row.close(rename=False)          # incomplete rows stay .tmp: the next read_directory() removes them
        self._open = {}

    # -- resume (continueFromLastRow, :454-462) ----------------------------------------------------------------
    def continue_from_last_row(self):
        """(row to continue from, its cells): the last complete row becomes the first row of the rest of the
        partition; the caller advances its first-column reader by (row - i0) cells"""
        i = self.last_disk_row_id()
        return i, self.read_row(i)


class SpecialRowsArea:
    """M/common/sra/SpecialRowsArea.cpp: the partitions of one stage (one directory)"""

    def __init__(self, directory, persistent=True, ram_limit=0, disk_limit=1):
        """ram_limit / disk_limit: the two budgets (--ram-size, --disk-size); only their proportion matters here --
        rows alternate between memory and disk accordingly.  The default keeps every row on disk."""
        self.directory = directory
        self.persistent = persistent
        self.ram_limit, self.disk_limit = max(int(ram_limit), 0), max(int(disk_limit), 0)
        self.partitions = {}
        self._anonymous = []           # non-persistent partitions have no path to be keyed by
        self.rows = 0

    def set_persistent(self, persistent):
        self.persistent = persistent

    def create_partition(self, i0, j0, i1, j1, scratch=False):
        """scratch: under a name open_partition_at does not look for ("guess.<rectangle>": stage 2's sweeps from guessed
        crosspoints) -- such a partition only gets its place among the stage's partitions when truncate_partition moves it
        there; what a killed run leaves of them is removed by remove_scratch_partitions"""
        p = SpecialRowsPartition(self.directory, i0, j0, i1, j1, persistent=self.persistent, scratch=scratch)
        p.set_ram_proportion(self.ram_limit, self.disk_limit)
        if self.persistent:
            self.partitions[p.path] = p
        else:
            self.partitions[""] = p    # createPartition keys by path: every non-persistent partition lands on ""
        return p

    def open_partition(self, i0, j0, i1, j1):
        """openPartition(i0,j0,i1,j1) (:66-78)"""
        path = os.path.join(self.directory, "%08X.%08X.%08X.%08X" % (i0, j0, i1, j1))
        p = self.partitions.get(path)
        if p is None:
            p = self.partitions[path] = SpecialRowsPartition(self.directory, i0, j0, i1, j1, read_only=True)
        else:
            p.read_directory()         # rows on disk + the rows this object keeps in memory, border readers from the markers
        return p

    def open_partition_at(self, i, j):
        """openPartition(i, j) (:118-146): the partition whose cells (border excluded) hold DP cell (i, j)"""
        _files.drain()
        for name in os.listdir(self.directory):
            tok = name.split(".")
            if len(tok) == 4 and all(len(t) == 8 for t in tok):
                try:
                    i0, j0, i1, j1 = (int(t, 16) for t in tok)
                except ValueError:
                    continue
                if i0 < i <= i1 and j0 < j <= j1:
                    return self.open_partition(i0, j0, i1, j1)
        return None

    def truncate_partition(self, p, max_i, max_j):
        """(:80-95)"""
        old = p.path
        p.truncate(max_i, max_j)
        if self.persistent:
            new = os.path.join(self.directory, "%08X.%08X.%08X.%08X" % (p.i0, p.j0, max_i, max_j))
            p.change_path(new)
            self.partitions.pop(old, None)
            self.partitions[new] = p
        self.rows += p.rows_count()

    def remove_scratch_partitions(self):
        """leftovers of sweeps from guessed crosspoints (a run that died between its batch and its walk)"""
        import shutil
        _files.drain()
        if not os.path.isdir(self.directory):
            return 0
        gone = 0
        for name in os.listdir(self.directory):
            if name.startswith(SCRATCH_PREFIX):
                shutil.rmtree(os.path.join(self.directory, name), ignore_errors=True)
                gone += 1
        return gone

    def discard_partition(self, p):
        """a partition nobody is going to read (stage 2's sweep from a crosspoint guess that turned out wrong): its rows, its
        directory and its entry go away -- no counterpart in the reference, which never sweeps on a guess"""
        p.close()
        p._ram = {}
        if self.persistent and p.path:
            import shutil
            if self.partitions.get(p.path) is p:
                del self.partitions[p.path]
            _files.submit(p, _remove_tree, p.path)

    def rows_count(self):
        """(:97-103)"""
        return self.rows if self.persistent else 0

    def partitions_count(self):
        return len(self.partitions)


def get_area(areas, work, stage, ident=0, deep=-1, ram_limit=0, disk_limit=1):
    """Job::getSpecialRowsArea (M/common/Job.cpp:273-297): one area object per directory for the life of `areas` (a
    dict the caller keeps across stages) -- rows a stage kept in memory are only there for the stages that share it"""
    d = special_rows_path(work, stage, ident, deep)
    if areas is None:
        return SpecialRowsArea(d, ram_limit=ram_limit, disk_limit=disk_limit)
    if d not in areas:
        areas[d] = SpecialRowsArea(d, ram_limit=ram_limit, disk_limit=disk_limit)
    return areas[d]


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
        self.value_key = None          # what the value belongs to: (i0, j0, i1, j1, alignment start, alignment end)
        self.loaded = False
        _files.drain()
        if os.path.exists(self.side):
            tok = open(self.side).read().split()
            if len(tok) >= 3:
                self.value_best = (int(tok[0]), int(tok[1]), int(tok[2]))
            if len(tok) >= 9:
                self.value_key = tuple(int(x) for x in tok[3:9])
        # Status::load (Status.cpp:40-66) falls back to the temporary file when a kill fell between its write and the rename
        src = self.file if os.path.exists(self.file) else (self.tmp if os.path.exists(self.tmp) else None)
        if src is not None:
            tok = open(src).read().split()
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
        """queued behind the rows written so far (see _FileQueue): on disk, a status never names a row that is not"""
        side = None
        if self.value_best is not None:
            side = "%d %d %d" % self.value_best
            if self.value_key is not None:
                side += " %d %d %d %d %d %d" % tuple(self.value_key)
            side += "\n"
        if best is not None:
            self.best = tuple(int(x) for x in best)
        b = self.best if self.best is not None else (-1, -1, -INF)
        _files.submit(self, self._save, side, "%d\n%d\n%d %d %d\n" % (self.stage, self.last_special_row, b[0], b[1], b[2]))

    def _save(self, side, text):
        if side is not None:
            with open(self.side + ".tmp", "w") as f:
                f.write(side)
            os.replace(self.side + ".tmp", self.side)
        with open(self.tmp, "w") as f:
            f.write(text)
        os.replace(self.tmp, self.file)

    def drop_value_best(self):
        """stage 1 is complete: the value-only record of a two-phase run has served its purpose"""
        self.value_best = self.value_key = None
        _files.drain()
        for fn in (self.side, self.side + ".tmp"):
            if os.path.exists(fn):
                os.remove(fn)


def flush_intervals(m, n, limit, max_deep=20):
    """Job::calculateFlushIntervals, M/common/Job.cpp:231-257: the special-row spacing of stage 1, stage 2 and the
    rounds of stage 3.  From the second entry on the reference divides by `limit / SRA_DECAY` with SRA_DECAY = 1.0f:
    single-precision arithmetic, reproduced here."""
    if limit < n * CELL_BYTES * 2:
        limit = n * CELL_BYTES * 2
    f32 = np.float32
    out = [int(m * n * CELL_BYTES // limit + 1)]
    for k in range(1, max_deep):
        length = n if k % 2 == 1 else m
        v = int(f32(out[k - 1] * length * CELL_BYTES) / f32(limit) + f32(1))
        if k >= 2 and v > out[k - 2] // 2:
            v = out[k - 2] // 2                      # each round at least halves the spacing of the round before last
        out.append(v)
    return out


def flush_interval(m, n, limit):
    """Job::calculateFlushIntervals, M/common/Job.cpp:231-241 (first interval): rows between two special rows for an
    area of `limit` bytes; at least two rows fit"""
    if limit <= 0:
        return 0
    if limit < n * CELL_BYTES * 2:
        limit = n * CELL_BYTES * 2
    return int(m * n * CELL_BYTES // limit + 1)

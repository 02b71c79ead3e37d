"""CPU: the native Special Rows Area writer / status file / stage-1 driver (masa-cudalign_amd/sra.py, stage1.py)
against what MASA-Core itself wrote for the same pair (fixture: directory and file names, file sizes, sha256 of every
row file, status and crosspoint text), and the resume logic: a run that is cut off in the middle -- leaving a
half-written <row>.tmp behind -- continues from the last complete special row and ends with the same files and the
same best score as an uninterrupted run.  The aligner here is a stand-in built on the oracle (the product engine
needs a GPU; tests/test_gpu_sra.py repeats the same checks with it, including a SIGKILL)."""
import hashlib
import os

import numpy as np
import pytest

from helpers import load_golden, make_pair

G = load_golden()
CASE = [c for c in G["cases"] if c["name"] == "sw_special_rows_20000x9000"][0]


class OracleAligner:
    """IAligner surface used by stage1(): setSequences / alignPartition(partition, manager) / getStatistics, with the
    oracle doing the arithmetic and the hook order of AbstractDiagonalAligner (special rows in increasing order with
    their leading first-column cell, then the last row, then the scores)."""

    def __init__(self, oracle, strip_rows=1024, chunk=4000):
        self.o, self.strip_rows, self.chunk = oracle, strip_rows, chunk

    def setSequences(self, s0, s1):
        self.s0, self.s1 = np.asarray(s0), np.asarray(s1)

    def unsetSequences(self):
        pass

    def getStatistics(self):
        return {"strip_rows": self.strip_rows, "kernel_ms": 0.0, "pruned_cells": 0}

    def alignPartition(self, part, mg):
        """block rows of K rows (K = the special-row spacing), top to bottom: the scores of a block are dispatched
        before the special row below it, as the engine does"""
        o = self.o
        m, n = part.i1 - part.i0, part.j1 - part.j0
        row = np.zeros((n + 1, 2), dtype=np.int32)
        col = np.zeros((m + 1, 2), dtype=np.int32)
        mg.receiveFirstColumn(col[:1], 1)
        mg.receiveFirstRow(row[:1], 1)
        mg.receiveFirstRow(row[1:], n)
        mg.receiveFirstColumn(col[1:], m)
        interval = mg.getSpecialRowInterval()
        K = m
        if interval > 0 and mg.mustDispatchSpecialRows():
            K = max(-(-interval // self.strip_rows), -(-8192 // self.strip_rows)) * self.strip_rows
        r0 = 0
        while r0 < m and mg.mustContinue():
            r1 = min(r0 + K, m)
            res = o.stage1(self.s0[part.i0 + r0:part.i0 + r1], self.s1[part.j0:part.j1], recurrence=mg.getRecurrenceType(),
                           first_row_type=o.INIT_WITH_CUSTOM_DATA, custom_first_row=row,
                           first_col_type=o.INIT_WITH_CUSTOM_DATA, custom_first_col=col[r0:r1 + 1],
                           block_h=self.strip_rows, block_w=1 << 20, want_last_row=True)
            row = res["last_row"]
            if mg.mustDispatchScores() and res["best"][2] > -o.INF:
                i, j, sc = res["best"]
                mg.dispatchScore((part.i0 + r0 + i - 1, part.j0 + j - 1, sc))
            if r1 < m or mg.mustDispatchLastRow():
                lead = np.array([[col[r1, 0], -o.INF]], dtype=np.int32)
                mg.dispatchRow(part.i0 + r1, lead, 1)
                for j in range(0, n, self.chunk):
                    mg.dispatchRow(part.i0 + r1, row[1 + j:1 + j + self.chunk], min(self.chunk, n - j))
            r0 = r1


def _sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def _listing(work):
    root = os.path.join(work, "special_rows", "stage.01.00")
    return {d: {fn: os.path.getsize(os.path.join(root, d, fn)) for fn in sorted(os.listdir(os.path.join(root, d)))}
            for d in sorted(os.listdir(root))}


def _check_against_reference(work, res):
    assert list(res["best"]) == CASE["best"]
    assert _listing(work) == CASE["sra_listing"]                      # names and sizes as MASA-Core wrote them
    d = list(CASE["sra_listing"])[0]
    for rid, dg in CASE["special_rows"].items():
        assert _sha(os.path.join(work, "special_rows", "stage.01.00", d, "%08X" % int(rid))) == dg["sha256"], rid
    assert open(os.path.join(work, "status")).read() == CASE["status_txt"]
    assert open(os.path.join(work, "crosspoints", "crosspoint_01.00")).read() == CASE["crosspoint_txt"]


def test_area_written_like_masa_core(pkg, oracle, tmp_path):
    s0, s1 = make_pair(pkg, CASE["seq"])
    work = str(tmp_path / "work")
    res = pkg.stage1(OracleAligner(oracle, strip_rows=1024), s0, s1, work, sra_limit=200 * 1024, block_pruning=False)
    assert res["resumed_from"] is None
    _check_against_reference(work, res)
    # running it again finds stage 1 done (sw_stage1.cpp:212-214)
    again = pkg.stage1(OracleAligner(oracle), s0, s1, work, sra_limit=200 * 1024)
    assert again.get("already_done") and list(again["best"]) == CASE["best"]


@pytest.mark.parametrize("stop_after_rows", [1, 2])
def test_cut_off_run_resumes_to_the_same_result(pkg, oracle, tmp_path, stop_after_rows):
    s0, s1 = make_pair(pkg, CASE["seq"])
    work = str(tmp_path / "work")

    class Killed(Exception):
        pass

    class DyingManager(pkg.Stage1Manager):
        """dies in the middle of the row after `stop_after_rows` complete ones"""
        calls = 0

        def dispatchRow(self, i, buf, length):
            pkg.Stage1Manager.dispatchRow(self, i, buf, length)
            if len(self.sra.rows) >= stop_after_rows and length > 1:
                DyingManager.calls += 1
                if DyingManager.calls == 2:             # second chunk of the next row: a .tmp is on disk
                    raise Killed()

    with pytest.raises(Killed):
        pkg.stage1(OracleAligner(oracle), s0, s1, work, sra_limit=200 * 1024, block_pruning=False,
                   manager_class=DyingManager)
    d = os.path.join(work, "special_rows", "stage.01.00", list(CASE["sra_listing"])[0])
    left = sorted(os.listdir(d))
    assert any(fn.endswith(".tmp") for fn in left)
    assert sum(1 for fn in left if len(fn) == 8) == stop_after_rows
    st = pkg.sra.Status(work)
    assert st.stage == 1 and st.last_special_row == 8192 * stop_after_rows
    res = pkg.stage1(OracleAligner(oracle), s0, s1, work, sra_limit=200 * 1024, block_pruning=False)
    assert res["resumed_from"] == 8192 * stop_after_rows
    _check_against_reference(work, res)


def test_partition_bookkeeping(pkg, tmp_path):
    sra = pkg.sra
    area = sra.special_rows_path(str(tmp_path), 1, 0)
    assert area.endswith(os.path.join("special_rows", "stage.01.00"))
    p = sra.SpecialRowsPartition(area, 100, 200, 1100, 205)
    assert os.path.basename(p.path) == "00000064.000000C8.0000044C.000000CD"
    assert p.width_cells == 6 and p.last_row_id() == 100
    cells = np.arange(12, dtype=np.int32).reshape(6, 2)
    assert p.write(600, cells[:1]) is False
    assert os.path.exists(os.path.join(p.path, "000001F4.tmp"))
    assert os.path.getsize(os.path.join(p.path, "000001F4.tmp")) == 48          # final size from the first write on
    assert p.write(600, cells[1:4]) is False
    assert p.write(600, cells[4:]) is True
    assert sorted(os.listdir(p.path)) == ["000001F4"] and p.last_row_id() == 600
    assert np.array_equal(p.read_row(600), cells)
    p.write(900, cells[:3])
    p.close()
    q = sra.SpecialRowsPartition(area, 100, 200, 1100, 205)                       # re-open: the .tmp is swept
    assert sorted(os.listdir(q.path)) == ["000001F4"] and q.rows == [500]
    i, row = q.continue_from_last_row()
    assert i == 600 and np.array_equal(row, cells)
    q.set_border_markers(pkg.INIT_WITH_GAPS, 7, pkg.INIT_WITH_ZEROES, 0)
    assert {"R00000007.INIT_WITH_GAPS", "C00000000.INIT_WITH_ZEROES"} <= set(os.listdir(q.path))
    assert sra.flush_interval(20000, 9000, 200 * 1024) == 7032
    assert sra.flush_interval(20000, 9000, 0) == 0


def test_status_file_round_trip(pkg, tmp_path):
    st = pkg.sra.Status(str(tmp_path))
    assert not st.loaded
    st.stage, st.last_special_row = 1, 16384
    st.save((9004, 9000, 8091))
    assert open(os.path.join(str(tmp_path), "status")).read() == "1\n16384\n9004 9000 8091\n"
    assert not os.path.exists(os.path.join(str(tmp_path), "status.tmp"))
    again = pkg.sra.Status(str(tmp_path))
    assert again.loaded and (again.stage, again.last_special_row, again.best) == (1, 16384, (9004, 9000, 8091))


def test_progress_line_of_stage1(pkg, oracle, tmp_path):
    """MASA-Core's two-second progress line (logStatus, sw_stage1.cpp:112-128) from the native stage 1: a timer thread
    next to the blocking aligner call"""
    import io
    from oracle.aligner_double import SerialBlockAligner
    from masa_cudalign_amd.stage1 import stage1, progress_line
    assert progress_line(3725.9, (1, 2, 3), "PROGRESS: 5/10 strips") == "(1h02m05s) best:(1,2,3) PROGRESS: 5/10 strips"

    class Aligner(SerialBlockAligner):
        def getProgressString(self):
            return "PROGRESS: %d cells" % self.cells
    s0, s1 = pkg.seqgen.related_pair(9000, 9000, cfg=3)
    buf = io.StringIO()
    r = stage1(Aligner(128, 128), s0, s1, str(tmp_path / "w"), progress=buf, progress_interval=0.05, block_pruning=False)
    lines = buf.getvalue().splitlines()
    assert lines and all(ln.startswith("(0h00m0") and " best:(" in ln and "PROGRESS: " in ln for ln in lines)
    assert r["best"][2] == oracle.stage1(s0, s1)["best"][2]


def test_resume_continues_from_the_last_row_on_disk(pkg, oracle, tmp_path):
    """rows kept in memory (--ram-size) die with the process: a resumed stage 1 continues from the last row that reached
    the disk and still ends with the uninterrupted run's best score"""
    s0, s1 = make_pair(pkg, CASE["seq"])
    work = str(tmp_path / "w")

    class Killed(Exception):
        pass

    class DyingManager(pkg.Stage1Manager):
        def dispatchRow(self, i, buf, length):
            pkg.Stage1Manager.dispatchRow(self, i, buf, length)
            if 16384 in self.sra.rows:                   # rows 8192 (memory) and 16384 (disk) are complete
                self.active = False
                raise Killed()
    with pytest.raises(Killed):
        pkg.stage1(OracleAligner(oracle, strip_rows=1024), s0, s1, work, sra_limit=100 * 1024, ram_limit=100 * 1024,
                   block_pruning=False, manager_class=DyingManager)
    d = os.path.join(work, "special_rows", "stage.01.00", list(CASE["sra_listing"])[0])
    on_disk = sorted(int(fn, 16) for fn in os.listdir(d) if len(fn) == 8)
    assert on_disk == [16384]                                 # row 8192 was only in memory
    res = pkg.stage1(OracleAligner(oracle, strip_rows=1024), s0, s1, work, sra_limit=100 * 1024, ram_limit=100 * 1024,
                     block_pruning=False)
    assert res["resumed_from"] == on_disk[-1]
    assert tuple(res["best"]) == tuple(CASE["best"])


def test_queued_file_operations_keep_their_order(pkg, tmp_path):
    """inside sra.async_files() rows, renames, truncations and the status file are carried out by one thread in the order
    they were asked for: what is on disk afterwards is what the inline form leaves, a reader of a partition waits for
    that partition's operations, and a status file never names a row that is not in place"""
    sra = pkg.sra
    cells = (np.arange(2 * 5000, dtype=np.int32).reshape(5000, 2) * 7) % 100003
    seen = []

    def build(area_dir, work):
        area = sra.SpecialRowsArea(area_dir)
        p = area.create_partition(0, 0, 4000, 4999)
        st = sra.Status(work)
        for row in (1000, 2000, 3000):
            for j in range(0, 5000, 1024):
                done = p.write(row, cells[j:j + 1024] + row)
            assert done is True
            st.stage, st.last_special_row = 1, row
            st.save((row, row, row))
        p.write(3500, cells[:100])                      # incomplete: closed by truncate(), renamed as it is
        r = sra.SpecialRowReader(p, 2000)               # reads wait for the partition's queued operations
        r.seek(5000)
        buf = np.empty((5000, 2), dtype=np.int32)
        assert r.read(buf, 5000) == 5000 and np.array_equal(buf[::-1], cells + 2000)
        area.truncate_partition(p, 2500, 3999)          # rows 3000 and 3500 go, the others lose their last 1000 cells
        seen.append(os.path.basename(p.path))
        return p

    os.makedirs(os.path.join(str(tmp_path), "wa"))
    os.makedirs(os.path.join(str(tmp_path), "wb"))
    with sra.async_files():
        p = build(os.path.join(str(tmp_path), "a"), os.path.join(str(tmp_path), "wa"))
        assert np.array_equal(p.read_row(1000), (cells + 1000)[:4000])
    q = build(os.path.join(str(tmp_path), "b"), os.path.join(str(tmp_path), "wb"))   # the same, inline

    def tree(d):
        out = {}
        for root, _, files in os.walk(d):
            for fn in files:
                out[os.path.relpath(os.path.join(root, fn), d)] = hashlib.sha256(open(os.path.join(root, fn), "rb").read()).hexdigest()
        return out
    assert seen[0] == seen[1] == "00000000.00000000.000009C4.00000F9F"
    assert tree(os.path.join(str(tmp_path), "a")) == tree(os.path.join(str(tmp_path), "b")) and len(tree(os.path.join(str(tmp_path), "a"))) == 2
    assert tree(os.path.join(str(tmp_path), "wa")) == tree(os.path.join(str(tmp_path), "wb"))
    assert open(os.path.join(str(tmp_path), "wa", "status")).read() == "1\n3000\n3000 3000 3000\n"
    assert p.rows == q.rows == [1000, 2000]


def test_queued_file_operation_that_fails_is_reported(pkg, tmp_path):
    sra = pkg.sra
    area = sra.SpecialRowsArea(os.path.join(str(tmp_path), "a"))
    os.makedirs(area.directory)
    with pytest.raises(RuntimeError, match="queued file operation failed"):
        with sra.async_files():
            p = area.create_partition(0, 0, 100, 9)
            p.write(50, np.zeros((10, 2), dtype=np.int32))
            sra._files.submit(p, os.remove, os.path.join(p.path, "no such row"))
            sra.drain()
    sra.drain()                                          # the queue is usable again
    with sra.async_files():
        p.write(60, np.ones((10, 2), dtype=np.int32))
    assert np.array_equal(p.read_row(60), np.ones((10, 2), dtype=np.int32))


def test_file_operations_inline_on_request(pkg, tmp_path, monkeypatch):
    """MI355SW_SRA_SYNC=1: no thread, every operation at once"""
    sra = pkg.sra
    monkeypatch.setenv("MI355SW_SRA_SYNC", "1")
    with sra.async_files():
        p = sra.SpecialRowsPartition(os.path.join(str(tmp_path), "a"), 0, 0, 100, 9)
        assert p.write(50, np.zeros((10, 2), dtype=np.int32)) is True
        assert os.listdir(p.path) == ["00000032"] and not sra._files.pending


def test_file_queue_says_when_nobody_is_left_to_carry_out_its_operations(pkg):
    """wait()/drain() used to return silently when the file thread was gone with operations pending: the files a
    caller then reads were never written"""
    from masa_cudalign_amd import sra as sra_mod
    q = sra_mod._FileQueue()
    owner = object()
    q.pid = os.getpid()
    q.pending[id(owner)] = 2                   # two operations queued, the thread that would carry them out never started / died
    with pytest.raises(RuntimeError, match="never carried out"):
        q.drain()
    assert not q.pending
    q.pending[id(owner)] = 1
    with pytest.raises(RuntimeError, match="never carried out"):
        q.wait(owner)


def test_file_queue_starts_empty_in_a_forked_child(pkg):
    """a child process inherits a snapshot of the parent's queue: it must neither repeat the parent's operations nor wait on
    the parent's condition variable"""
    from masa_cudalign_amd import sra as sra_mod
    q = sra_mod._FileQueue()
    q.pid = os.getpid() + 1                    # "another process's" queue, as a fork leaves it
    q.pending[1] = 3
    q.q.append((None, lambda: None, (), 0))
    cv = q.cv
    q.drain()                                  # resets, then has nothing to wait for
    assert not q.pending and not q.q and q.pid is None and q.cv is not cv
    done = []
    with sra_mod.async_files():
        pass
    q.depth = 1
    q.submit(None, done.append, 1)             # and works: a thread of this process is started
    q.drain()
    assert done == [1]


def test_scratch_partitions_are_invisible_until_accepted(pkg, tmp_path):
    """Sweeps from guessed crosspoints (stage2._Speculation) keep their rows under "guess.<rectangle>": open_partition_at never
    finds such a directory, truncate_partition moves an accepted one to its real name, discard_partition removes a rejected one,
    and what a killed run left is cleared by remove_scratch_partitions (a stale guess must not be opened as
    a partition by a later stage)."""
    sra = pkg.sra
    area = sra.SpecialRowsArea(str(tmp_path / "area"))
    os.makedirs(area.directory, exist_ok=True)
    a = area.create_partition(0, 0, 4000, 300, scratch=True)
    b = area.create_partition(0, 0, 5000, 300, scratch=True)
    c = area.create_partition(10, 10, 6000, 300, scratch=True)
    sra._files.drain()
    names = sorted(os.listdir(area.directory))
    assert len(names) == 3 and all(n.startswith("guess.") for n in names)
    assert area.open_partition_at(100, 100) is None              # nothing a reader could pick up
    area.truncate_partition(a, 1000, 200)                        # accepted: cut back to its crosspoint and moved into place
    area.discard_partition(b)                                    # rejected
    sra._files.drain()
    names = sorted(os.listdir(area.directory))
    assert names == sorted(["%08X.%08X.%08X.%08X" % (0, 0, 1000, 200), os.path.basename(c.path)])
    p = area.open_partition_at(100, 100)
    assert p is not None and (p.i1, p.j1) == (1000, 200)
    assert area.remove_scratch_partitions() == 1                 # c: as if the run had died here
    assert sorted(os.listdir(area.directory)) == ["%08X.%08X.%08X.%08X" % (0, 0, 1000, 200)]

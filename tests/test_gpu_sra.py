"""GPU (-m gpu): the native stage-1 driver (masa-cudalign_amd/stage1.py) with the real engine: special rows, status
and crosspoint files byte-identical to what MASA-Core wrote for the same pair; a run killed with SIGKILL in the middle
of writing its special rows resumes to exactly the files and the best score of an uninterrupted run; and the same for
the two-phase tracking used on very tall matrices, where the best cell of the part computed before the kill is only
known by value and has to be located after the resume."""
import hashlib
import os
import signal
import subprocess
import sys
import time

import numpy as np
import pytest

from helpers import load_golden, make_pair
from test_sra import _check_against_reference, _listing, _sha, CASE

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_area_written_by_the_engine_like_masa_core(pkg, tmp_path):
    s0, s1 = make_pair(pkg, CASE["seq"])
    work = str(tmp_path / "work")
    al = pkg.MI355Aligner(device=0, rows_per_lane=16)             # 1024-row strips: CUDAlign's 8192-row spacing
    try:
        res = pkg.stage1(al, s0, s1, work, sra_limit=200 * 1024, block_pruning=False)
    finally:
        al.close()
    assert res["resumed_from"] is None and res["strip_rows"] == 1024
    _check_against_reference(work, res)


CHILD = r"""
import sys, time
sys.path.insert(0, %(root)r)
import __graft_entry__ as g
pkg = g.load_package()
class Slow(pkg.Stage1Manager):
    def dispatchRow(self, i, buf, length):
        pkg.Stage1Manager.dispatchRow(self, i, buf, length)
        if length > 1:
            time.sleep(0.05)
s0, s1 = pkg.seqgen.related_pair(%(m)d, %(n)d, cfg=77)
al = pkg.MI355Aligner(device=0, rows_per_lane=16)
pkg.stage1(al, s0, s1, %(work)r, sra_limit=%(limit)d, manager_class=Slow, block_pruning=%(prune)r)
print("child finished", flush=True)
"""


def _tree(work):
    out = {}
    for root, _, files in os.walk(work):
        for fn in files:
            p = os.path.join(root, fn)
            out[os.path.relpath(p, work)] = _sha(p)
    return out


@pytest.mark.timeout(600)
@pytest.mark.parametrize("prune", [False, True])
def test_sigkill_in_the_middle_then_resume(pkg, tmp_path, prune):
    """pruning off: every file of the resumed run equals the uninterrupted run's, byte for byte.  Pruning on: which
    slabs are skipped depends on when the running best became known, so special rows differ off the optimal path
    (any two runs do); best score, status and crosspoint are the same."""
    m, n, limit = 200000, 30000, 8 << 20
    work = str(tmp_path / "killed")
    child = subprocess.Popen([sys.executable, "-c", CHILD % dict(root=ROOT, m=m, n=n, work=work, limit=limit, prune=prune)],
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    d = os.path.join(work, "special_rows", "stage.01.00", "%08X.%08X.%08X.%08X" % (0, 0, m, n))
    t0 = time.time()
    while time.time() - t0 < 300:
        done = [fn for fn in (os.listdir(d) if os.path.isdir(d) else []) if len(fn) == 8]
        if len(done) >= 4 or child.poll() is not None:
            break
        time.sleep(0.01)
    assert child.poll() is None, child.stdout.read().decode(errors="replace")[-2000:]
    child.send_signal(signal.SIGKILL)
    child.wait(timeout=60)
    rows_before = sorted(int(fn, 16) for fn in os.listdir(d) if len(fn) == 8)
    assert 4 <= len(rows_before) < m // 8192                       # killed in the middle
    st = pkg.sra.Status(work)
    assert st.stage == 1 and st.last_special_row in rows_before
    # resume, and an uninterrupted run of the same pair next to it
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=77)
    al = pkg.MI355Aligner(device=0, rows_per_lane=16)
    try:
        res = pkg.stage1(al, s0, s1, work, sra_limit=limit, block_pruning=prune)
        assert res["resumed_from"] == rows_before[-1]
        ref_work = str(tmp_path / "straight")
        ref = pkg.stage1(al, s0, s1, ref_work, sra_limit=limit, block_pruning=prune)
    finally:
        al.close()
    assert ref["resumed_from"] is None
    assert tuple(res["best"]) == tuple(ref["best"])
    a, b = _tree(work), _tree(ref_work)
    assert sorted(a) == sorted(b)
    if prune:
        assert ref["pruned_cells"] > 0
        a = {k: v for k, v in a.items() if not k.startswith("special_rows")}
        b = {k: v for k, v in b.items() if not k.startswith("special_rows")}
    assert a == b                                                   # every special row, status, crosspoint: same bytes
    assert len(_listing(work)[os.path.basename(d)]) == m // 8192 + 1 + 2   # rows + last row + two border markers


@pytest.mark.parametrize("planted_early", [True, False])
def test_two_phase_run_cut_off_and_resumed(pkg, oracle, tmp_path, monkeypatch, planted_early):
    """two-phase tracking (forced here; automatic from 32 Mi rows): the strips computed before the cut are only known
    by their best VALUE.  When that value wins, the resumed run locates its cell with one exact pass from the special
    row above it; the answer is the oracle's either way."""
    monkeypatch.setenv("MI355SW_TWO_PHASE", "1")
    m, n = 120000, 6000
    s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=78)
    s0, s1 = s0.copy(), s1.copy()
    at = 13000 if planted_early else 100000
    s1[2000:2600] = s0[at:at + 600]                                 # a 600-base exact repeat: the best by far
    ref = oracle.stage1(s0, s1)
    assert 600 <= ref["best"][2] <= 610 and abs(ref["best"][0] - (at + 600)) <= 10
    work = str(tmp_path / "work")

    class Killed(Exception):
        pass

    class DyingManager(pkg.Stage1Manager):
        dead = False

        def dispatchRow(self, i, buf, length):
            if DyingManager.dead:
                return
            pkg.Stage1Manager.dispatchRow(self, i, buf, length)
            if len(self.sra.rows) >= 5 and length > 1:              # dies inside the 6th special row (row 49152)
                DyingManager.dead = True
                self.active = False
                raise Killed()

    al = pkg.MI355Aligner(device=0, rows_per_lane=16)
    try:
        with pytest.raises(Killed):
            pkg.stage1(al, s0, s1, work, sra_limit=4 << 20, manager_class=DyingManager)
        st = pkg.sra.Status(work)
        assert st.last_special_row == 5 * 8192
        if planted_early:
            assert st.value_best is not None and st.value_best[0] == ref["best"][2]
            assert st.value_best[1] < ref["best"][0] <= st.value_best[2]
        res = pkg.stage1(al, s0, s1, work, sra_limit=4 << 20)
    finally:
        al.close()
    assert res["resumed_from"] == 5 * 8192
    assert tuple(res["best"]) == tuple(ref["best"])
    assert (res["located_from_value"] is not None) == planted_early
    assert open(os.path.join(work, "crosspoints", "crosspoint_01.00")).read() == "START\n0,%d,%d,%d\nEND\n" % tuple(ref["best"])

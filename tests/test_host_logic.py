"""CPU: host-side mirror of the caller (manager.py), band splitting, generator determinism."""
import hashlib

import numpy as np
import pytest


def test_initial_cells_reader_matches_oracle(pkg, oracle):
    for (go, ge, typ) in [(0, 0, oracle.INIT_WITH_ZEROES), (3, 2, oracle.INIT_WITH_GAPS), (0, 2, oracle.INIT_WITH_GAPS_OPENED)]:
        for off in (0, 1, 777):
            rd = pkg.InitialCellsReader(go, ge, off)
            assert rd.getType() == typ
            a = np.empty((50, 2), dtype=np.int32)
            rd.read(a, 1)
            rd.read(a[1:], 49)
            assert np.array_equal(a, oracle.initial_cells(typ, off, 50))


def test_best_score_list_order(pkg):
    b = pkg.BestScoreList(0)
    for t in [(5, 9, 10), (4, 20, 10), (4, 3, 10), (1, 1, 9), (9, 9, -1)]:
        b.add(*t)
    assert b.getBestScore() == (4, 3, 10)
    assert pkg.BestScoreList(0).getBestScore() == (-1, -1, -pkg.INF)


def test_band_limits_follow_reference_split(pkg):
    # libmasa.cpp:511-512 : trim_j0 = len*P[s-1]/sum + 1 (1-based), trim_j1 = len*P[s]/sum
    from masa_cudalign_amd.bands import band_limits
    n, w = 1000003, [100, 100, 250, 50]
    lim = band_limits(n, w)
    acc = np.cumsum([0] + w)
    for s in range(1, len(w) + 1):
        assert lim[s - 1] + 1 == n * int(acc[s - 1]) // int(acc[-1]) + 1
        assert lim[s] == n * int(acc[s]) // int(acc[-1])
    assert lim[0] == 0 and lim[-1] == n


def test_generator_is_deterministic(pkg):
    a = pkg.seqgen.random_dna(0xC0FFEE00, 1000)
    assert hashlib.sha256(a.tobytes()).hexdigest() == hashlib.sha256(pkg.seqgen.random_dna(0xC0FFEE00, 1000, chunk=64).tobytes()).hexdigest()
    assert set(np.unique(a)) <= set(b"ACGT")
    s0, s1 = pkg.seqgen.related_pair(5000, 4000, cfg=1)
    t0, t1 = pkg.seqgen.related_pair(5000, 4000, cfg=1)
    assert np.array_equal(s0, t0) and np.array_equal(s1, t1) and len(s1) == 4000


def test_stage1_manager_on_oracle_rows(pkg, oracle):
    """drive the Python manager mirror with rows/columns produced by the oracle and compare its
    bookkeeping (semi-global best on last row/column) with the oracle's own."""
    s0, s1 = pkg.seqgen.related_pair(700, 900, cfg=5)
    r = oracle.stage1(s0, s1, recurrence=oracle.NEEDLEMAN_WUNSCH, first_col_type=oracle.INIT_WITH_GAPS,
                      block_h=128, block_w=200, want_last_row=True, want_last_col=True,
                      best_mode=oracle.BEST_LAST_ROW_OR_COL)
    part = pkg.Partition(0, 0, 700, 900)
    mg = pkg.Stage1Manager(part, alignment_start=pkg.AT_SEQUENCE_1, alignment_end=pkg.AT_SEQUENCE_1_OR_2)
    assert mg.getRecurrenceType() == pkg.NEEDLEMAN_WUNSCH
    assert mg.getFirstColumnInitType() == pkg.INIT_WITH_GAPS and mg.getFirstRowInitType() == pkg.INIT_WITH_ZEROES
    lr, lc = r["last_row"], r["last_col"]
    mg.dispatchColumn(900, lc[:1], 1)
    for i in range(0, 700, 128):
        ch = lc[1 + i:1 + min(i + 128, 700)]
        mg.dispatchColumn(900, ch, len(ch))
    mg.dispatchRow(700, lr[:1], 1)
    for j in range(0, 900, 200):
        ch = lr[1 + j:1 + min(j + 200, 900)]
        mg.dispatchRow(700, ch, len(ch))
    assert tuple(mg.getBestScore()) == tuple(r["best"])


def test_band_loop_gives_up_on_a_stalled_band(pkg, monkeypatch):
    """a band whose kernel never completes a row is reported and then aborted (bands.py stall reporter) instead of
    hanging its neighbours: the engine is told to abort and the error names the band"""
    from masa_cudalign_amd.bands import BandRunner

    class StuckEngine:
        aborted = ended = False

        def streamBegin(self, part, **kw):
            pass

        def streamPoll(self):
            return 0, False

        def streamAbort(self):
            self.aborted = True

        def streamEnd(self):
            self.ended = True
            return (-1, -1, -pkg.INF), 0

    monkeypatch.setenv("MI355SW_BAND_DEBUG", "1")
    monkeypatch.setenv("MI355SW_BAND_STALL_S", "2")
    eng = StuckEngine()
    runner = BandRunner(eng, dist=None, rank=0, world=1)
    with pytest.raises(RuntimeError, match="band 0/1: no progress"):
        runner.run(1000, 0, 500)
    assert eng.aborted and eng.ended


def test_crosspoint_array_writer_equals_the_object_writer(pkg, tmp_path):
    """crosspoints.save_array (millions of points at C3's size: no object per point) writes the bytes CrosspointsFile.save
    writes for the same points (M/common/CrosspointsFile.cpp:152-160: START, type,i,j,score per line, END)"""
    import numpy as np
    from masa_cudalign_amd.crosspoints import Crosspoint, CrosspointsFile, save_array
    rng = np.random.default_rng(5)
    pts = np.stack([rng.integers(0, 3, 1000), np.sort(rng.integers(0, 10 ** 8, 1000)), np.sort(rng.integers(0, 10 ** 8, 1000)),
                    rng.integers(-10 ** 7, 10 ** 8, 1000)], axis=1).astype(np.int32)
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    f = CrosspointsFile(a)
    f.extend(Crosspoint(int(i), int(j), int(s), int(t)) for (t, i, j, s) in pts)
    f.save()
    save_array(b, pts)
    assert open(a, "rb").read() == open(b, "rb").read()
    save_array(b, [tuple(int(x) for x in r) for r in pts[:3]])            # a list of tuples is taken as well
    assert open(b).read().split()[1] == "%d,%d,%d,%d" % tuple(pts[0])


def test_chain_seed_bound_only_where_the_seed_applies(pkg, monkeypatch):
    """bands.chain_seed_bound asks the engine for the diagonal seed of the whole matrix only for large matrices whose borders
    are the recurrence's own (zeroes for a local alignment, gap penalties from the origin for a global one)"""
    from masa_cudalign_amd import bands

    class Eng:
        asked = []

        def seedBound(self, part, rec):
            self.asked.append(((part.i0, part.j0, part.i1, part.j1), rec))
            return 4242

    e, big = Eng(), 9 << 20
    Z, G, C = pkg.INIT_WITH_ZEROES, pkg.INIT_WITH_GAPS, pkg.INIT_WITH_CUSTOM_DATA
    assert bands.chain_seed_bound(e, big, big, pkg.SMITH_WATERMAN, Z, Z) == 4242
    assert bands.chain_seed_bound(e, big, big, pkg.NEEDLEMAN_WUNSCH, G, G) == 4242
    assert e.asked == [((0, 0, big, big), pkg.SMITH_WATERMAN), ((0, 0, big, big), pkg.NEEDLEMAN_WUNSCH)]
    for args in ((1 << 20, big, pkg.SMITH_WATERMAN, Z, Z), (big, 1 << 20, pkg.SMITH_WATERMAN, Z, Z),       # too small
                 (big, big, pkg.SMITH_WATERMAN, G, Z), (big, big, pkg.NEEDLEMAN_WUNSCH, Z, G),               # other borders
                 (big, big, pkg.NEEDLEMAN_WUNSCH, G, C)):
        assert bands.chain_seed_bound(e, *args) is None, args
    assert bands.chain_seed_bound(object(), big, big, pkg.SMITH_WATERMAN, Z, Z) is None                       # an engine without the entry point
    monkeypatch.setenv("MI355SW_NO_DIAGONAL_SEED", "1")
    assert bands.chain_seed_bound(e, big, big, pkg.SMITH_WATERMAN, Z, Z) is None
    assert len(e.asked) == 2


def test_gpu_call_scripts_parse_and_python_tools_compile():
    """the scripts a GPU call runs are not exercised by any CPU test: a syntax error in one of them costs a box (minutes of the
    round's GPU budget) to find.  bash -n for the shell ones, py_compile for the Python ones."""
    import glob
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for sh in sorted(glob.glob(os.path.join(root, "tools", "*.sh")) + glob.glob(os.path.join(root, "oracle", "*.sh"))):
        r = subprocess.run(["bash", "-n", sh], capture_output=True, text=True)
        assert r.returncode == 0, (sh, r.stderr)
    for py in sorted(glob.glob(os.path.join(root, "tools", "*.py")) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]):
        compile(open(py, "rb").read(), py, "exec")

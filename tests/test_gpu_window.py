"""GPU (-m gpu): the pruning WINDOW -- strips of a pruning run leave the skipped cells outside their band unwritten, jump over
what the strip above jumped over and retire behind the end of the alignment (csrc/sw_kernel_pk16.inc, WIN_RETIRED).

Reference: BlockPruningDiagonal::updatePruningWindow (M/libmasa/pruning/BlockPruningDiagonal.cpp:109-152) hands
[windowStart, windowEnd] to processDiagonal (AbstractDiagonalAligner.cpp:491-501); blocks outside the window cost a branch
(X/CUDAligner.cu:950-960).  Whatever a window leaves out must read exactly like a pruned block: the oracle for a windowed run is
the same run without the window (MI355SW_F_NO_WINDOW) and, for every value that matters, the unpruned oracle."""
import numpy as np
import pytest

from test_gpu_bound import _stream

pytestmark = pytest.mark.gpu
INF = 999999999


def _shapes(k):
    rng = np.random.default_rng(7000 + k)
    m = int(rng.integers(20000, 140000))
    n = int(np.clip(m * rng.uniform(0.3, 1.6), 16000, 150000))
    R = int(rng.choice([4, 4, 8, 16]))
    kind = int(rng.integers(0, 4))      # 0: related from the corner  1: seq1 = a piece from the middle of seq0  2: global  3: short homology, long tail
    if kind == 2:                       # a global alignment of sequences of like length (the packed window holds its scores)
        n = int(np.clip(m * rng.uniform(0.92, 1.08), 16000, 150000))
    return m, n, R, kind


def _pair(pkg, k):
    sg = pkg.seqgen
    m, n, R, kind = _shapes(k)
    if kind in (0, 2):
        s0, s1 = sg.related_pair(m, n, cfg=700 + k, inversion=0.05 if k % 2 else 0.0)
    elif kind == 1:
        s0 = sg.random_dna(sg.SEED0 + 700 + k, m)
        i0 = m // 4
        s1 = sg.mutate_dna(s0[i0:i0 + n + n // 8 + 64], sg.SEED1 + 700 + k, inversion=0.0)
        s1 = np.ascontiguousarray(np.concatenate([s1, sg.random_dna(5 + k, max(0, n - len(s1)))])[:n])
    else:
        s0 = sg.random_dna(sg.SEED0 + 700 + k, m)
        L = min(m, n) // 3
        s1 = np.ascontiguousarray(np.concatenate([sg.mutate_dna(s0[:L + L // 8 + 64], sg.SEED1 + 700 + k, inversion=0.0)[:L], sg.random_dna(9 + k, n - L)]))
    return s0, s1, m, n, R, kind


@pytest.mark.parametrize("k", range(12))
def test_window_against_the_oracle_and_against_the_run_without_it(pkg, oracle, k):
    """12 seeded shapes (20 000 ... 150 000, alignments from the corner, from the middle of seq0, ending early, global): with
    the window the best cell / H[m][n] is the oracle's; last row, last column and every special row are lower bounds of the
    oracle's rows; and the run without the window -- same kernels, every skipped cell written -- agrees on all of it"""
    from masa_cudalign_amd.engine import SMITH_WATERMAN, NEEDLEMAN_WUNSCH, F_NO_WINDOW
    from helpers import oracle_full
    s0, s1, m, n, R, kind = _pair(pkg, k)
    rec = NEEDLEMAN_WUNSCH if kind == 2 else SMITH_WATERMAN
    edge = 4 if kind == 2 else 0
    ref = oracle_full(oracle, s0, s1, edge=edge)
    want_rows = dict(zip(ref["special_row_ids"], ref["special_rows"]))
    final = int(ref["last_row"][-1, 0])
    res = {}
    for flags in (0, F_NO_WINDOW):
        al = pkg.MI355Aligner(device=0, rows_per_lane=R, flags=flags)
        try:
            al.setSequences(s0, s1)
            # (the bound starts from the answer, as behind a diagonal seed: the band is then as narrow as it gets)
            bound = final if kind == 2 else ref["best"][2]
            res[flags] = _stream(pkg, al, m, n, rec, bound)
        finally:
            al.close()
    for flags, got in res.items():
        st = got["stats"]
        assert st["profile_kernel"] == 2 and st["restarts"] == 0
        if kind == 2:
            assert int(got["last_row"][-1, 0]) == final
        else:
            assert got["best"] == tuple(ref["best"]), (flags, got["best"], ref["best"])
        assert st["pruned_cells"] > 0 and st["pruned_cells"] + st["processed_cells"] == m * n
        assert np.all(got["last_row"] <= ref["last_row"][1:]) and np.all(got["last_col"] <= ref["last_col"][1:])
        assert sorted(got["rows"]) == [i for i in sorted(want_rows) if i % 8192 == 0 and i < m] and len(got["rows"]) >= 2
        for i, cells in got["rows"].items():
            assert np.all(cells <= want_rows[i][1:]), (flags, i)
            if kind != 2:
                assert np.all(cells[:, 0] >= 0), (flags, i)                     # a skipped cell of a local alignment reads 0, never less
                # a row the optimal alignment crosses (its maximum is far above the noise of unrelated cells, and it lies
                # above the best cell) carries that maximum exactly, at the same column
                w = want_rows[i][1:, 0]
                if i <= ref["best"][0] and int(w.max()) > 200:
                    assert int(cells[:, 0].max()) == int(w.max()) and int(cells[:, 0].argmax()) == int(w.argmax()), (flags, i)
    a, b = res[0], res[F_NO_WINDOW]
    # the window changes what is WRITTEN, not what is skipped (up to the timing of the running bound, which may differ by a few slabs)
    assert abs(a["stats"]["pruned_cells"] - b["stats"]["pruned_cells"]) <= 0.05 * m * n, (a["stats"]["pruned_cells"], b["stats"]["pruned_cells"])


def test_strips_below_the_end_of_the_alignment_retire(pkg, oracle):
    """a tall matrix whose alignment ends after a sixth of the rows: every strip below holds nothing but skipped cells and
    retires after its first chunks instead of walking 100 000 columns; special rows down there read as zeroes in full; with and
    without the window, one-pass and two-phase tracking give the oracle's canonical cell"""
    from masa_cudalign_amd.engine import SMITH_WATERMAN, F_NO_WINDOW, F_TWO_PHASE
    sg = pkg.seqgen
    m, n = 600000, 100000
    s0 = sg.random_dna(sg.SEED0 + 720, m)
    s1 = np.ascontiguousarray(np.concatenate([sg.mutate_dna(s0[:n + n // 8 + 64], sg.SEED1 + 720, inversion=0.0)[:n - 3000], sg.random_dna(77, 3000)]))
    from helpers import oracle_full
    ref = oracle_full(oracle, s0[:110000], s1)               # everything that aligns lies in the first 110 000 rows
    opt = ref["best"][2]
    out = {}
    for flags in (0, F_NO_WINDOW, F_TWO_PHASE, F_TWO_PHASE | F_NO_WINDOW):
        al = pkg.MI355Aligner(device=0, rows_per_lane=8, flags=flags)
        try:
            al.setSequences(s0, s1)
            out[flags] = _stream(pkg, al, m, n, SMITH_WATERMAN, opt, interval=65536)
        finally:
            al.close()
    for flags, got in out.items():
        assert got["best"] == tuple(ref["best"]), (flags, got["best"])
        deep = [i for i in got["rows"] if i > 200000]
        assert len(deep) >= 5
        for i in deep:
            # (left of column n - optimum a cell still has more columns ahead of it than the best score is worth: nothing there is
            #  skipped, unrelated sequences score their usual handful; right of it the strips retire and the row reads as the
            #  constant the host filled in)
            #  (... but for the ragged last chunk of the row, which every strip that walks that far computes)
            row = got["rows"][i]
            edge = n - opt + 2048
            assert int(row[:, 0].max()) <= 40 and not row[edge:n - 256, 0].any() and np.all(row[edge:n - 256, 1] == -INF), (flags, i)
        assert int(got["last_row"][:, 0].max()) <= 40 and not got["last_row"][n - opt + 2048:n - 256, 0].any()
        assert got["stats"]["pruned_cells"] > 0.8 * m * n
    print("kernel ms with / without the window: %.1f / %.1f (two-phase: %.1f / %.1f)" % (
        out[0]["stats"]["kernel_ms"], out[F_NO_WINDOW]["stats"]["kernel_ms"], out[F_TWO_PHASE]["stats"]["kernel_ms"],
        out[F_TWO_PHASE | F_NO_WINDOW]["stats"]["kernel_ms"]))
    # not a benchmark, but the point of the exercise: 480 strips that do nothing must not cost more than 480 walks of the width
    # (round 5: 43 against 50 ms; round 6, where a walk skips the chunks at the end of the row too: 50 against 49 -- within the
    #  noise of a 50 ms launch; the window's gain is at sizes where a walk is long: 16 M x 14.65 M 7.84 -> 7.18 s)
    assert out[0]["stats"]["kernel_ms"] < 1.25 * out[F_NO_WINDOW]["stats"]["kernel_ms"]


def test_window_through_the_manager_interface_with_pruning_on(pkg, oracle):
    """mi355sw_align_partition (the IAligner seam) on a pruning run with special rows, last row and last column wanted: the
    window is on by default and the manager sees lower bounds of the oracle's cells with the oracle's best cell"""
    m, n = 90000, 80000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=731)
    from helpers import oracle_full
    ref = oracle_full(oracle, s0, s1)
    want_rows = dict(zip(ref["special_row_ids"], ref["special_rows"]))
    al = pkg.MI355Aligner(device=0)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        mg = pkg.Stage1Manager(part, special_row_interval=8192, keep_last_row=True, keep_last_column=True, block_pruning=True)
        al.alignPartition(part, mg)
        st = al.getStatistics()
        assert tuple(mg.getBestScore()) == tuple(ref["best"]) and st["pruned_cells"] > 0.15 * m * n
        assert np.all(mg.lastRow() <= ref["last_row"]) and np.all(mg.lastColumn() <= ref["last_col"])
        crossed = 0
        for i in sorted(mg.special_rows):
            got, want = mg.specialRow(i), want_rows[i]
            assert np.all(got <= want), i
            if i <= ref["best"][0]:                 # rows the optimal alignment crosses: its score, exactly, where the oracle has it
                assert got[:, 0].max() == want[:, 0].max() and got[:, 0].argmax() == want[:, 0].argmax(), i
                crossed += 1
        assert crossed >= 5
    finally:
        al.close()

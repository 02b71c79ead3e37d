"""GPU (-m gpu): BASELINE config C1 -- 1 000 000 x 1 000 000 unrelated, local SW -- against the result the REFERENCE itself
computed (tests/golden/c1_reference.json, made by oracle/make_golden_c1.py: MASA-Core's CPU path as a chain of eight column
bands, sw_stage1.cpp + libmasa.cpp:497-535): best score and canonical position, every boundary column of the chain (sha256 of
1 000 001 cells each), ten special rows.  "The single large pinned result" of SURVEY.md 8(c)."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
INF = 999999999
FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_reference.json")


@pytest.fixture(scope="module")
def c1(pkg):
    with open(FIXTURE) as f:
        fx = json.load(f)
    s0, s1 = pkg.seqgen.unrelated_pair(fx["m"], fx["n"], cfg=fx["seq"]["cfg"])
    assert hashlib.sha256(s0.tobytes()).hexdigest() == fx["seq0_sha256"] and hashlib.sha256(s1.tobytes()).hexdigest() == fx["seq1_sha256"]
    return fx, s0, s1


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.int32).tobytes()).hexdigest()


def test_c1_best_cell_on_the_default_kernel_family(pkg, c1):
    """the engine as bench.py and a MASA run use it (default configuration, no forced strip height): the reference's best
    score and its canonical position; and the same with block pruning left on, MASA-Core's default"""
    fx, s0, s1 = c1
    m, n = fx["m"], fx["n"]
    al = pkg.MI355Aligner(device=0)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        for prune in (False, True):
            mg = pkg.Stage1Manager(part, block_pruning=prune)
            al.alignPartition(part, mg)
            st = al.getStatistics()
            assert st["profile_kernel"] == 2 and st["restarts"] == 0 and st["kernel"].startswith("sw_strip_kernel_pk16")
            assert list(mg.getBestScore()) == fx["best"], (prune, st["kernel"])
            assert st["pruned_cells"] == 0            # an unrelated pair: nothing to prune, whoever asks
    finally:
        al.close()


def test_c1_boundary_columns_are_the_references(pkg, c1):
    """column lim[k] of the matrix = the last column of the partition [0, lim[k]): seven partitions of growing width, each
    column (H,E) of rows 0..m byte for byte what MASA-Core's band k handed to band k+1"""
    fx, s0, s1 = c1
    m = fx["m"]
    al = pkg.MI355Aligner(device=0)
    try:
        al.setSequences(s0, s1)
        for k in range(1, fx["parts"]):
            j = fx["band_limits"][k]
            part = pkg.Partition(0, 0, m, j)
            mg = pkg.Stage1Manager(part, keep_last_column=True)
            al.alignPartition(part, mg)
            col = mg.lastColumn()
            want = fx["boundary_columns"][str(j)]
            assert col.shape == (want["len"], 2) and col[:4].tolist() == want["head"] and col[-4:].tolist() == want["tail"], j
            assert _sha(col) == want["sha256"], j
            assert int(col[:, 0].max()) == want["max_h"]
    finally:
        al.close()


def test_c1_special_rows_are_the_references(pkg, c1):
    """special rows every 96 256 rows (the reference's flush interval for --disk-size=80M per band), 1024-row strips: each
    row, cut at the chain's band limits the way the eight nodes stored their slices (the cell left of the band first, its F
    void: AbstractDiagonalAligner.cpp:290-298), has the reference's digests"""
    fx, s0, s1 = c1
    m, n, lim = fx["m"], fx["n"], fx["band_limits"]
    ids = sorted(int(i) for i in fx["special_rows"])
    interval = ids[0]
    assert all(i == interval * (k + 1) for k, i in enumerate(ids)) and interval % 1024 == 0
    al = pkg.MI355Aligner(device=0, rows_per_lane=16)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        mg = pkg.Stage1Manager(part, special_row_interval=interval)
        al.alignPartition(part, mg)
        assert list(mg.getBestScore()) == fx["best"]
        assert sorted(mg.special_rows) == ids
        for i in ids:
            row = mg.specialRow(i)                       # n + 1 cells: column 0 first
            assert row.shape == (n + 1, 2)
            for k in range(fx["parts"]):
                want = fx["special_rows"][str(i)][k]
                piece = row[lim[k]:lim[k + 1] + 1].copy()
                piece[0, 1] = -INF
                assert piece.shape[0] == want["len"] and piece[:4].tolist() == want["head"], (i, k)
                assert _sha(piece) == want["sha256"], (i, k)
    finally:
        al.close()

"""CPU: the C restatement (oracle/sw_oracle.c) against the fixtures generated from the reference's own
MASA-Core CPU path (tests/golden/stage1_cases.json, made by oracle/make_golden.py)."""
import hashlib

import numpy as np
import pytest

from helpers import load_golden, make_pair, digest, parse_args, oracle_kwargs

G = load_golden()


@pytest.mark.parametrize("case", G["cases"], ids=[c["name"] for c in G["cases"]])
def test_oracle_matches_reference_fixture(case, pkg, oracle):
    s0, s1 = make_pair(pkg, case["seq"])
    assert hashlib.sha256(s0.tobytes()).hexdigest() == case["seq0_sha256"]
    assert hashlib.sha256(s1.tobytes()).hexdigest() == case["seq1_sha256"]
    p = parse_args(case["args"])
    kw = oracle_kwargs(oracle, p, len(s0), len(s1))
    r = oracle.stage1(s0, s1, **kw)
    assert list(r["best"]) == case["best"]
    if "special_rows" in case:
        ids = r["special_row_ids"]
        for key, dg in case["special_rows"].items():
            i = int(key)
            row = r["special_rows"][ids.index(i)] if i in ids else r["last_row"]
            assert digest(row) == dg, "special row %d" % i


def test_oracle_geometry_invariance(pkg, oracle):
    """best score/position and every border are independent of the block geometry."""
    s0, s1 = pkg.seqgen.related_pair(2500, 3100, cfg=21)
    base = oracle.stage1(s0, s1, block_h=2500, block_w=3100, want_last_row=True, want_last_col=True)
    for bh, bw in [(1, 3100), (7, 13), (64, 64), (512, 100), (1000, 1)]:
        r = oracle.stage1(s0, s1, block_h=bh, block_w=bw, want_last_row=True, want_last_col=True)
        assert r["best"] == base["best"]
        assert np.array_equal(r["last_row"], base["last_row"])
        assert np.array_equal(r["last_col"], base["last_col"])
    mt = oracle.stage1(s0, s1, block_h=128, block_w=128, threads=4)
    assert mt["best"] == base["best"]


def test_oracle_chain_matches_reference(pkg, oracle):
    """column bands chained through the boundary column reproduce the reference's --split run."""
    ch = G["chain"]
    s0, s1 = make_pair(pkg, ch["seq"])
    n, parts = len(s1), ch["parts"]
    lim = [n * k // parts for k in range(parts + 1)]
    col = None
    cands = []
    for k in range(parts):
        j0, j1 = lim[k], lim[k + 1]
        kw = dict(want_last_col=True, row_start_offset=j0)
        if col is not None:
            kw.update(first_col_type=oracle.INIT_WITH_CUSTOM_DATA, custom_first_col=col)
        r = oracle.stage1(s0, s1[j0:j1], **kw)
        col = r["last_col"]
        if k < parts - 1:
            assert digest(col) == ch["boundary_columns"]["STEP-%d.tmp" % (k + 1)]
        b = r["best"]
        cands.append((b[0], b[1] + j0, b[2]))
        # the reference relays the running best down the chain (sw_stage1.cpp:421-426, 459-464)
        run = max(cands, key=lambda t: (t[2], -t[0], -t[1]))
        assert list(run) == ch["band_bests"][k]
    assert list(run) == ch["single_best"]


@pytest.mark.parametrize("case", [c for c in G["cases"] if "crosspoints_4" in c], ids=lambda c: c["name"])
def test_stage4_restatement_reproduces_the_reference_file(case, pkg, oracle):
    """oracle/stage4_oracle.c (ort_split_2 + the step loop of sw_stage4.cpp) against crosspoint_04.00 as MASA-Core's own
    CPU stage 4 wrote it: same points, same order -- sha256 of the file text"""
    import hashlib
    s0, s1 = make_pair(pkg, case["seq"])
    pts, steps = oracle.stage4(s0, s1, [tuple(p) for p in case["crosspoints_3"]], 16)
    txt = "START\n" + "".join("%d,%d,%d,%d\n" % p for p in pts) + "END\n"
    assert len(pts) == case["crosspoints_4"]["count"]
    assert hashlib.sha256(txt.encode()).hexdigest() == case["crosspoints_4"]["file_sha256"]


def test_c1_reference_fixture_pins_the_restatement(pkg, oracle):
    """BASELINE config C1 (1 000 000 x 1 000 000 unrelated, local SW) went once through MASA-Core's own CPU path as a chain of
    eight column bands (oracle/make_golden_c1.py -> tests/golden/c1_reference.json: best cell, the seven boundary columns,
    ten special rows).  The whole matrix is the GPU's to reproduce (tests/test_gpu_c1.py); here the C restatement recomputes
    what it can in seconds -- the first 16 384 rows of band 1 -- and must give the head of the reference's first boundary column."""
    import hashlib
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_reference.json")
    with open(path) as f:
        fx = json.load(f)
    assert fx["m"] == fx["n"] == 1000000 and fx["parts"] == 8 and len(fx["boundary_columns"]) == 7 and len(fx["special_rows"]) >= 8
    s0, s1 = pkg.seqgen.unrelated_pair(fx["m"], fx["n"], cfg=fx["seq"]["cfg"])
    assert hashlib.sha256(s0.tobytes()).hexdigest() == fx["seq0_sha256"] and hashlib.sha256(s1.tobytes()).hexdigest() == fx["seq1_sha256"]
    assert fx["best"] == fx["band_bests"][-1] and fx["best"][2] == max(b[2] for b in fx["band_bests"])
    j = fx["band_limits"][1]
    col = fx["boundary_columns"][str(j)]
    rows = col["head_cells"] - 1
    ref = oracle.stage1(s0[:rows], s1[:j], want_last_col=True)
    got = np.ascontiguousarray(ref["last_col"], dtype=np.int32)
    assert got.shape == (rows + 1, 2) and got[:4].tolist() == col["head"]
    assert hashlib.sha256(got.tobytes()).hexdigest() == col["head_sha256"]


@pytest.mark.parametrize("m,n,edge", [(30000, 26000, 0), (20000, 31000, 4), (8191, 5000, 0), (16384, 3000, 0)])
def test_threaded_oracle_equals_the_serial_one(pkg, oracle, m, n, edge):
    """oracle_stage1_mt (blocks on an anti-diagonal wavefront of threads) hands out the serial schedule's best cell, special
    rows, last row and last column: the larger GPU parity cases use it (helpers.oracle_full)"""
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=3)
    kw = oracle_kwargs(oracle, dict(start=edge, end=edge, pruning=False, disk=-1, block=(1024, n)), m, n)
    kw.update(want_last_row=True, want_last_col=True, special_row_interval=8192)
    a = oracle.stage1(s0, s1, **kw)
    kw.update(block_w=1000, threads=8)
    b = oracle.stage1(s0, s1, **kw)
    assert a["best"] == b["best"] and a["special_row_ids"] == b["special_row_ids"] and len(a["special_row_ids"]) >= 1
    assert np.array_equal(a["last_row"], b["last_row"]) and np.array_equal(a["last_col"], b["last_col"])
    assert np.array_equal(a["special_rows"], b["special_rows"])

"""GPU (-m gpu): stage 4 on the GPU (mi355sw_stage4: batched Myers-Miller refinement, csrc/stage4.hip) against
  * the crosspoint_04 files MASA-Core's own CPU stage 4 wrote for the full-pipeline fixtures (same points, same order,
    same tie-breaks: sha256 of the file text), and
  * the oracle's restatement of sw_stage4.cpp (oracle/stage4_oracle.c, itself pinned on those files) on seeded pairs:
    tall and wide partitions (both orientations), halves of more than 256 rows (several passes of the systolic sweep),
    start/end crosspoints of every type (inputs that are themselves partly refined lists), coded and raw sequences."""
import hashlib

import numpy as np
import pytest

from helpers import load_golden, make_pair

pytestmark = pytest.mark.gpu
G = load_golden()
FULL = [c for c in G["cases"] if "crosspoints_4" in c]


def _text(points):
    return ("START\n" + "".join("%d,%d,%d,%d\n" % tuple(p) for p in points) + "END\n").encode()


@pytest.mark.parametrize("case", FULL, ids=[c["name"] for c in FULL])
def test_crosspoint_04_file_of_the_reference(case, pkg, aligner):
    s0, s1 = make_pair(pkg, case["seq"])
    aligner.setSequences(s0, s1)
    try:
        pts, st = aligner.stage4([tuple(p) for p in case["crosspoints_3"]], 16)
    finally:
        aligner.unsetSequences()
    want = case["crosspoints_4"]
    assert len(pts) == want["count"]
    assert [list(p) for p in pts[:4]] == [list(p) for p in want["head"]]
    assert [list(p) for p in pts[-4:]] == [list(p) for p in want["tail"]]
    assert hashlib.sha256(_text(pts)).hexdigest() == want["file_sha256"]
    assert st["steps"] >= 5 and st["partitions"] >= want["count"] // 2


def _global_endpoints(oracle, s0, s1):
    """a stage-3-like input: the two ends of the optimal GLOBAL alignment of the pair"""
    ref = oracle.stage1(s0, s1, recurrence=oracle.NEEDLEMAN_WUNSCH, first_row_type=oracle.INIT_WITH_GAPS,
                        first_col_type=oracle.INIT_WITH_GAPS, best_mode=oracle.BEST_LAST_CELL, want_last_row=True)
    return [(0, 0, 0, 0), (0, len(s0), len(s1), int(ref["last_row"][-1][0]))]


@pytest.mark.parametrize("m,n,cfg", [(40, 33, 1), (300, 290, 2), (1000, 2500, 3), (2600, 900, 4), (5000, 5200, 5),
                                     (20000, 18000, 6), (777, 16, 7), (17, 4000, 8)])
def test_against_the_oracle_on_seeded_pairs(pkg, oracle, m, n, cfg):
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=400 + cfg)
    cp = _global_endpoints(oracle, s0, s1)
    want, steps = oracle.stage4(s0, s1, cp, 16)
    al = pkg.MI355Aligner(device=0)
    try:
        al.setSequences(s0, s1)
        got, st = al.stage4(cp, 16)
        assert got == want and st["steps"] == steps
        # every type of start / end crosspoint: a partly refined list (limit 200) refined to the end
        partly, _ = oracle.stage4(s0, s1, cp, 200)
        if len(partly) > 2:
            assert {p[0] for p in want} >= {0} and al.stage4(partly, 16)[0] == oracle.stage4(s0, s1, partly, 16)[0]
        # other limits
        for limit in (1, 5, 64):
            assert al.stage4(cp, limit)[0] == oracle.stage4(s0, s1, cp, limit)[0], limit
    finally:
        al.close()


def test_gap_rich_alignment_has_gapped_crosspoints(pkg, oracle):
    """long insertions make crosspoints of type 1 and 2 (inside a gap) and partitions with a zero side"""
    rng = np.random.default_rng(9)
    a = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=6000)
    s0 = np.concatenate([a[:2000], rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=700), a[2000:]])
    s1 = np.concatenate([a[:4500], rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=450), a[4500:]])
    cp = _global_endpoints(oracle, s0, s1)
    want, _ = oracle.stage4(s0, s1, cp, 16)
    assert {p[0] for p in want} == {0, 1, 2}
    al = pkg.MI355Aligner(device=0)
    try:
        al.setSequences(s0, s1)
        assert al.stage4(cp, 16)[0] == want
    finally:
        al.close()


@pytest.mark.parametrize("letters", [b"ACGTN", b"ACGTNRYKMSWBDHV"])
def test_coded_and_raw_sequences(pkg, oracle, letters):
    """the refinement compares residues like stage 1 does: coded sequences (<= 14 common letters) and raw bytes"""
    rng = np.random.default_rng(len(letters))
    alpha = np.frombuffer(letters, dtype=np.uint8)
    s0 = alpha[rng.integers(0, 4, 3000)].copy()
    s1 = s0.copy()
    for s in (s0, s1):
        idx = rng.integers(0, len(s), 150)
        s[idx] = alpha[rng.integers(0, len(alpha), len(idx))]
    s1 = np.concatenate([s1[:1000], s1[1040:]])
    cp = _global_endpoints(oracle, s0, s1)
    al = pkg.MI355Aligner(device=0)
    try:
        al.setSequences(s0, s1)
        assert al.stage4(cp, 16)[0] == oracle.stage4(s0, s1, cp, 16)[0]
    finally:
        al.close()


def test_inconsistent_crosspoints_are_reported(pkg, oracle, aligner):
    """a score difference no alignment of the partition reaches: the reference prints NOT FOUND and exits"""
    s0, s1 = pkg.seqgen.related_pair(900, 800, cfg=77)
    cp = _global_endpoints(oracle, s0, s1)
    bad = [cp[0], (0, cp[1][1], cp[1][2], cp[1][3] + 7)]
    aligner.setSequences(s0, s1)
    try:
        with pytest.raises(pkg.AlignerError, match="ETRACEBACK"):
            aligner.stage4(bad, 16)
        with pytest.raises(pkg.AlignerError, match="EINVAL"):
            aligner.stage4([cp[1], cp[0]], 16)
    finally:
        aligner.unsetSequences()

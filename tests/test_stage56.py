"""CPU: native stages 5 and 6 (masa-cudalign_amd/stage56.py: per-partition traceback, gap lists, alignment text) and the
FASTA view they print from, against the alignment.00.txt MASA-Core wrote for the full-pipeline fixtures -- byte for byte
(sha256) -- starting from the refined crosspoints (the oracle's stage 4 here; the GPU's in tests/test_gpu_stage4.py)."""
import hashlib

import numpy as np
import pytest

from helpers import load_golden, make_pair

G = load_golden()
FULL = [c for c in G["cases"] if "crosspoints_4" in c]


def _seqs(pkg, case):
    from masa_cudalign_amd import fasta
    s0, s1 = make_pair(pkg, case["seq"])
    return s0, s1, fasta.parse(b">s0\n" + s0.tobytes() + b"\n"), fasta.parse(b">s1\n" + s1.tobytes() + b"\n")


@pytest.mark.parametrize("case", FULL, ids=[c["name"] for c in FULL])
def test_alignment_text_of_the_reference(case, pkg, oracle):
    from masa_cudalign_amd import stage56
    s0, s1, q0, q1 = _seqs(pkg, case)
    cp4, _ = oracle.stage4(s0, s1, [tuple(p) for p in case["crosspoints_3"]], 16)
    al = stage56.stage5(q0, q1, cp4)
    assert al.raw_score == case["best"][2]
    txt = stage56.stage6_text(al, q0, q1)
    assert hashlib.sha256(txt).hexdigest() == case["alignment_txt_sha256"]
    # the text re-scores itself (stage 6 refuses to write an alignment whose columns do not add up to the score)
    assert txt.decode().rstrip().splitlines()[-5].split()[-1] == str(case["best"][2])
    assert al.matches + al.mismatches + al.gap_extensions >= case["best"][2]


def test_gap_rich_alignment_against_the_live_reference(pkg, oracle, tmp_path):
    """long insertions on both sides: gapped crosspoints, partitions with a zero side, gaps crossing partition borders"""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    from masa_cudalign_amd import fasta, stage56
    rng = np.random.default_rng(9)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    a = rng.choice(acgt, size=6000)
    s0 = np.concatenate([rng.choice(acgt, size=300), a[:2000], rng.choice(acgt, size=700), a[2000:], rng.choice(acgt, size=200)])
    s1 = np.concatenate([a[:4500], rng.choice(acgt, size=450), a[4500:]])
    ref = oracle.run_ref(s0, s1, ["--disk-size=500K", "--block=8192,8192"], workdir=str(tmp_path))
    q0, q1 = fasta.parse(b">s0\n" + s0.tobytes() + b"\n"), fasta.parse(b">s1\n" + s1.tobytes() + b"\n")
    cp4, _ = oracle.stage4(s0, s1, ref["crosspoints_3"], 16)
    assert cp4 == ref["crosspoints_4"] and {p[0] for p in cp4} == {0, 1, 2}
    al = stage56.stage5(q0, q1, cp4)
    assert stage56.stage6_text(al, q0, q1) == ref["alignment_txt"]
    assert len(al.gaps[0]) >= 1 and len(al.gaps[1]) >= 1


def test_empty_and_trivial_paths(pkg):
    from masa_cudalign_amd import fasta, stage56
    q0, q1 = fasta.parse(b">a\nACGTACGT\n"), fasta.parse(b">b\nACGTTACGT\n")
    al = stage56.stage5(q0, q1, [(0, 0, 0, 0)])                    # a single crosspoint: no alignment
    txt = stage56.stage6_text(al, q0, q1).decode()
    assert "There was no alignment produced!" in txt and "Total Score:             0" in txt
    # ACGT-ACGT / ACGTTACGT : 8 matches, one gap of length 1 = 8 - 5
    al = stage56.stage5(q0, q1, [(0, 0, 0, 0), (0, 8, 9, 3)])
    assert (al.raw_score, al.matches, al.gap_open, al.gap_extensions) == (3, 8, 1, 1)
    assert len(al.gaps[0]) == 1 and al.gaps[0][0][1] == 1 and al.gaps[1] == []
    txt = stage56.stage6_text(al, q0, q1)
    assert b"ACG-TACGT" in txt and b"[3/3]" in txt and b"ACGTTACGT" in txt

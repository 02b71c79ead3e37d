"""CPU: masa-cudalign_amd/fasta.py against MASA-Core's own loader.  Where oracle/_ref exists the reference driver is run
on files that exercise the normalisation (lower case, CRLF, blanks, IUPAC letters, a header without residues on its
line) and its `info` / statistics are compared; everywhere, the rules of SequenceData.cpp:67-114 are checked directly."""
import os

import numpy as np
import pytest


def test_normalisation_rules(pkg):
    from masa_cudalign_amd import fasta
    raw = b">chr test  with blanks\r\nacgtN nryk\r\n\r\nAC GT\nnn\n"
    s = fasta.parse(raw)
    assert s.raw_description == ">chr test  with blanks\r\n" and s.description == "chr test  with blanks\r"
    assert s.forward.tobytes() == b"ACGTNNRYKACGTNN" and s.original_size == 15 and len(s) == 15
    c = fasta.parse(raw, fasta.SequenceModifiers(complement=True))
    assert c.forward.tobytes() == b"TGCANNRYKTGCANN"
    n = fasta.parse(raw, fasta.SequenceModifiers(clear_n=True))
    assert n.forward.tobytes() == b"ACGTnnRYKACGTnn"
    r = fasta.parse(raw, fasta.SequenceModifiers(reverse=True))
    assert r.data().tobytes() == b"NNTGCAKYRNNTGCA" and r.data(reverse=True).tobytes() == s.forward.tobytes()
    assert r.absolute_pos(1) == 15 and s.absolute_pos(1) == 1
    t = fasta.parse(raw, fasta.SequenceModifiers(trim_start=3, trim_end=7))
    assert t.trimmed().tobytes() == b"GTNNR" and len(t) == 5 and (t.offset0, t.offset1) == (3, 7)
    t = fasta.parse(raw, fasta.SequenceModifiers(trim_start=0, trim_end=0))
    assert t.trimmed().tobytes() == s.forward.tobytes()


def test_stage1_of_the_reference_sees_the_same_residues(pkg, oracle, tmp_path):
    """the reference driver aligns a messy file against its clean twin: a perfect match over the loader's length means
    both loaders produced the same residues"""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    from masa_cudalign_amd import fasta
    from oracle.binding import REF_DRIVER
    import subprocess
    rng = np.random.default_rng(3)
    clean = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, 5000)]
    messy = bytearray(b">messy sequence\r\n")
    for k, c in enumerate(clean.tobytes()):
        ch = bytes([c])
        messy += ch.lower() if k % 3 == 0 else ch
        if k % 61 == 60:
            messy += b"\r\n"
        if k % 97 == 0:
            messy += b" "
    f0, f1 = tmp_path / "messy.fasta", tmp_path / "clean.fasta"
    f0.write_bytes(bytes(messy))
    f1.write_bytes(b">clean\n" + clean.tobytes() + b"\n")
    mine = fasta.load(str(f0))
    assert mine.forward.tobytes() == clean.tobytes()
    p = subprocess.run([REF_DRIVER, "--work-dir=" + str(tmp_path / "work"), "--stage-1", "--no-flush", str(f0), str(f1)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120, cwd=str(tmp_path))
    assert p.returncode == 0
    cp = open(tmp_path / "work" / "crosspoints" / "crosspoint_01.00").read().split()
    assert cp[1] == "0,5000,5000,5000"                  # every residue of the messy file matched its clean twin

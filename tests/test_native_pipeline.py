"""CPU: the native traceback drivers (masa-cudalign_amd/stage2.py, stage3.py, pipeline.py, the AlignerManager of
manager.py, the read side of sra.py, crosspoints.py) against MASA-Core.

The drivers never compute a DP cell themselves: they talk to an aligner.  Here the aligner is
oracle/aligner_double.py -- MASA-Core's serial block aligner restated over the oracle's processBlock, the very aligner
oracle/ref_driver.cpp linked into the real MASA-Core when the fixtures were made.  Same grid, same special rows, same
order of dispatches: so the native drivers must reproduce the reference's crosspoint files of stages 2, 3 (every
round) and 4, its special-rows directories of stages 2 and 3, and alignment.00.txt, BYTE FOR BYTE -- against the
committed fixtures, and against the live reference (where oracle/_ref is built) on inputs chosen to reach gapped
crosspoints, both gap types, several rounds of stage 3, local / global / semi-global edges and the alignment's start
inside a partition.  On the GPU the same drivers run on the engine: tests/test_gpu_native_pipeline.py."""
import filecmp
import hashlib
import os
import subprocess

import numpy as np
import pytest

from helpers import load_golden, make_pair

G = load_golden()
FULL = [c for c in G["cases"] if "crosspoints_4" in c]


def _double(oracle, case_args):
    from oracle.aligner_double import SerialBlockAligner
    bh, bw = [a for a in case_args if a.startswith("--block=")][0][8:].split(",")
    return SerialBlockAligner(int(bh), int(bw))


def _limit(case_args):
    v = [a for a in case_args if a.startswith("--disk-size=")][0][12:]
    mult = {"K": 1024, "M": 1024 * 1024, "G": 1024 ** 3}.get(v[-1])
    return int(float(v[:-1]) * mult) if mult else int(v)


def _fasta(pkg, s0, s1):
    from masa_cudalign_amd import fasta
    return fasta.parse(b">s0\n" + s0.tobytes() + b"\n"), fasta.parse(b">s1\n" + s1.tobytes() + b"\n")


@pytest.mark.parametrize("case", FULL, ids=[c["name"] for c in FULL])
def test_pipeline_reproduces_the_fixture(case, pkg, oracle, tmp_path):
    """stages 1-6 natively on the fixture's pair with the fixture's block geometry: crosspoints of stages 2 and 3 equal
    the reference's lists, crosspoint_04.00 and alignment.00.txt its files (sha256).  The pruned fixture's reference
    run pruned 143 blocks in stage 1; the double does not prune -- the crosspoints do not depend on it."""
    from masa_cudalign_amd import pipeline
    from masa_cudalign_amd.crosspoints import CrosspointsFile, crosspoint_file
    s0, s1 = make_pair(pkg, case["seq"])
    q0, q1 = _fasta(pkg, s0, s1)
    work = str(tmp_path / "work")
    from helpers import parse_args
    edges = parse_args(case["args"])                      # --edges=++ (a global alignment) in one of the fixtures
    out = pipeline.align(_double(oracle, case["args"]), q0, q1, work, sra_limit=_limit(case["args"]), block_pruning=False,
                         alignment_start=edges["start"], alignment_end=edges["end"])
    assert list(out["best"]) == case["best"]
    cp2 = CrosspointsFile(crosspoint_file(work, 2)).load().tuples()
    cp3 = CrosspointsFile(crosspoint_file(work, 3)).load().tuples()
    assert cp2 == [tuple(p) for p in case["crosspoints_2"]]
    assert cp3 == [tuple(p) for p in case["crosspoints_3"]]
    assert hashlib.sha256(open(crosspoint_file(work, 4), "rb").read()).hexdigest() == case["crosspoints_4"]["file_sha256"]
    assert hashlib.sha256(out["text"]).hexdigest() == case["alignment_txt_sha256"]
    assert open(os.path.join(work, "alignment.00.txt"), "rb").read() == out["text"]
    # alignment.00.bin: the same content as MASA-Core's file; the same BYTES unless a gap run was cut in two by a
    # partition border -- the order of such twins in the reference's file is its C++ library's (alignment_file.py)
    from masa_cudalign_amd import alignment_file as af
    mine, theirs = open(os.path.join(work, "alignment.00.bin"), "rb").read(), bytes.fromhex(case["alignment_bin_hex"])
    assert hashlib.sha256(theirs).hexdigest() == case["alignment_bin_sha256"]
    assert af.canonical(af.loads(mine)) == af.canonical(af.loads(theirs))
    twins = any(len({g[0] for g in gaps}) != len(gaps) for gaps in af.loads(theirs)["result"]["gaps"])
    assert twins or mine == theirs
    assert out["alignment"].raw_score == case["best"][2]


def _gap_rich():
    rng = np.random.default_rng(9)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    a = rng.choice(acgt, size=6000)
    s0 = np.concatenate([rng.choice(acgt, size=300), a[:2000], rng.choice(acgt, size=700), a[2000:], rng.choice(acgt, size=200)])
    s1 = np.concatenate([a[:4500], rng.choice(acgt, size=450), a[4500:]])
    return s0, s1


def _contained(pkg):
    a = pkg.seqgen.random_dna(777, 4000)
    s0 = np.concatenate([pkg.seqgen.random_dna(778, 2500), a, pkg.seqgen.random_dna(779, 1500)])
    return s0, pkg.seqgen.mutate_dna(a, 780, inversion=0.0)


LIVE = [
    # name, pair, block h, block w, area limit, edges
    ("gap_rich_types_0_1", lambda pkg: _gap_rich(), 128, 128, 100 * 1024, "**"),
    ("gap_rich_other_grid", lambda pkg: _gap_rich(), 300, 200, 60 * 1024, "**"),
    ("unrelated_start_inside_first_partition", lambda pkg: pkg.seqgen.unrelated_pair(5000, 5000, cfg=3), 128, 128, 100 * 1024, "**"),
    ("many_indels", lambda pkg: pkg.seqgen.related_pair(12000, 12000, cfg=21, p_indel=0.02, indel_mean=6.0), 128, 128, 300 * 1024, "**"),
    ("wide_long_indels", lambda pkg: pkg.seqgen.related_pair(9000, 14000, cfg=22, p_indel=0.01, indel_mean=20.0), 256, 128, 200 * 1024, "**"),
    ("two_rounds_of_stage3", lambda pkg: pkg.seqgen.related_pair(27648, 27648, cfg=7), 256, 256, 663552, "**"),
    ("global", lambda pkg: pkg.seqgen.related_pair(6000, 6100, cfg=23, inversion=0.0), 128, 128, 150 * 1024, "++"),
    ("contained_global_all_types", _contained, 128, 128, 150 * 1024, "++"),
    ("contained_semiglobal_21", _contained, 128, 128, 150 * 1024, "21"),
    ("contained_semiglobal_31", _contained, 128, 128, 150 * 1024, "31"),
    ("contained_start_on_s2", _contained, 128, 128, 150 * 1024, "2*"),
    ("contained_swapped_12", lambda pkg: _contained(pkg)[::-1], 128, 256, 150 * 1024, "12"),
    ("contained_swapped_global", lambda pkg: _contained(pkg)[::-1], 128, 256, 150 * 1024, "22"),
    ("no_traceback_13", _contained, 128, 128, 150 * 1024, "13"),
    ("no_budget_for_special_rows", lambda pkg: pkg.seqgen.related_pair(3000, 2700, cfg=1), 128, 128, 0, "**"),
]


@pytest.mark.parametrize("name,pair,bh,bw,limit,edges", LIVE, ids=[x[0] for x in LIVE])
def test_stages_1_to_3_against_the_live_reference(name, pair, bh, bw, limit, edges, pkg, oracle, tmp_path):
    """every crosspoint file (02, 03.rNN, 03) and every special-rows directory of stages 2-3 equals MASA-Core's"""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    from oracle.aligner_double import SerialBlockAligner
    from masa_cudalign_amd.stage1 import stage1
    from masa_cudalign_amd.stage2 import stage2
    from masa_cudalign_amd.stage3 import stage3
    edge = {"*": pkg.AT_ANYWHERE, "1": pkg.AT_SEQUENCE_1, "2": pkg.AT_SEQUENCE_2, "3": pkg.AT_SEQUENCE_1_OR_2,
            "+": pkg.AT_SEQUENCE_1_AND_2}
    s0, s1 = pair(pkg)
    m = len(s0)
    args = ["--disk-size=%d" % limit, "--block=%d,%d" % (bh, bw), "--no-block-pruning"]
    if edges != "**":
        args.append("--edges=" + edges)
    refdir = tmp_path / "ref"
    refdir.mkdir()
    ref = oracle.run_ref(s0, s1, args, workdir=str(refdir), timeout=600)
    rwork = str(refdir / "work")
    work = str(tmp_path / "native")
    al = SerialBlockAligner(bh, bw)
    r1 = stage1(al, s0, s1, work, alignment_start=edge[edges[0]], alignment_end=edge[edges[1]], sra_limit=limit,
                block_pruning=False)
    assert tuple(r1["best"]) == tuple(ref["best"])
    r2 = stage2(al, s0, s1, work, alignment_start=edge[edges[0]], sra_limit=limit)
    r3 = stage3(al, s0, s1, work, sra_limit=limit)
    files = sorted(f for f in os.listdir(os.path.join(rwork, "crosspoints")) if not f.startswith("crosspoint_04"))
    assert "crosspoint_02.00" in files and "crosspoint_03.00" in files
    for f in files:
        assert filecmp.cmp(os.path.join(rwork, "crosspoints", f), os.path.join(work, "crosspoints", f), shallow=False), f
    for d in sorted(os.listdir(os.path.join(rwork, "special_rows"))):
        # (the native stage 1 also keeps the partition's last row as its completion marker: manager.py)
        p = subprocess.run(["diff", "-rq", "-x", "%08X" % m, os.path.join(rwork, "special_rows", d),
                            os.path.join(work, "special_rows", d)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert p.returncode == 0, p.stdout.decode()[:2000]
    if name == "two_rounds_of_stage3":
        assert len(r3["rounds"]) == 2 and "crosspoint_03.00.r02" in files
    if name.startswith("contained_global"):
        assert {t for (t, _, _, _) in r3["crosspoints"] + r2["crosspoints"]} == {0, 1, 2}
    if name.startswith("no_traceback"):
        assert len(r2["crosspoints"]) == 1 and r2["partitions"] == 0


def _revcomp(a):
    comp = np.arange(256, dtype=np.uint8)
    for x, y in ((65, 84), (84, 65), (67, 71), (71, 67)):
        comp[x] = y
    return np.ascontiguousarray(comp[a][::-1])


def _with_n(pkg):
    s0, s1 = pkg.seqgen.related_pair(3000, 2700, cfg=1)
    s0, s1 = s0.copy(), s1.copy()
    s0[1000:1040] = ord("N")
    s1[1005:1030] = ord("N")
    return s0, s1


MODS = [
    # name, pair, reference flags, modifiers of sequence 0, of sequence 1
    ("trim", lambda pkg: pkg.seqgen.related_pair(3000, 2700, cfg=1), ["--trim=201,2500,101,2600"],
     dict(trim_start=201, trim_end=2500), dict(trim_start=101, trim_end=2600)),
    ("trim_open_ends", lambda pkg: pkg.seqgen.related_pair(3000, 2700, cfg=1), ["--trim=0,2500,300,0"],
     dict(trim_end=2500), dict(trim_start=300)),
    ("reverse_complement_2", lambda pkg: (lambda p: (p[0], _revcomp(p[1])))(pkg.seqgen.related_pair(3000, 2700, cfg=1)),
     ["--reverse-complement=2"], dict(), dict(reverse=True, complement=True)),
    ("reverse_complement_2_trimmed", lambda pkg: (lambda p: (p[0], _revcomp(p[1])))(pkg.seqgen.related_pair(3000, 2700, cfg=1)),
     ["--reverse-complement=2", "--trim=100,2900,50,2650"], dict(trim_start=100, trim_end=2900),
     dict(reverse=True, complement=True, trim_start=50, trim_end=2650)),
    ("reverse_both", lambda pkg: tuple(np.ascontiguousarray(x[::-1]) for x in pkg.seqgen.related_pair(3000, 2700, cfg=1)),
     ["--reverse=both"], dict(reverse=True), dict(reverse=True)),
    ("clear_n", _with_n, ["--clear-n"], dict(clear_n=True), dict(clear_n=True)),
    ("n_kept", _with_n, [], dict(), dict()),
]


@pytest.mark.parametrize("name,pair,flags,mod0,mod1", MODS, ids=[x[0] for x in MODS])
def test_sequence_modifiers_through_all_stages(name, pair, flags, mod0, mod1, pkg, oracle, tmp_path):
    """--trim, --reverse, --complement, --clear-n from the FASTA view (fasta.py) through stages 1-6: trimming only
    selects the part of the matrix stage 1 sweeps, all coordinates stay absolute; every crosspoint file, the
    special-rows directories and alignment.00.txt (header with the trimmed range, positions of the reversed strand)
    equal MASA-Core's"""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    from oracle.aligner_double import SerialBlockAligner
    from masa_cudalign_amd import fasta, pipeline
    s0, s1 = pair(pkg)
    limit = 200 * 1024
    refdir = tmp_path / "ref"
    refdir.mkdir()
    ref = oracle.run_ref(s0, s1, ["--disk-size=%d" % limit, "--block=128,128", "--no-block-pruning"] + flags,
                         workdir=str(refdir), timeout=600)
    rwork = str(refdir / "work")
    q0 = fasta.parse(b">s0\n" + s0.tobytes() + b"\n", fasta.SequenceModifiers(**mod0))
    q1 = fasta.parse(b">s1\n" + s1.tobytes() + b"\n", fasta.SequenceModifiers(**mod1))
    work = str(tmp_path / "native")
    out = pipeline.align(SerialBlockAligner(128, 128), q0, q1, work, sra_limit=limit, block_pruning=False)
    assert tuple(out["best"]) == tuple(ref["best"])
    for f in sorted(os.listdir(os.path.join(rwork, "crosspoints"))):
        assert filecmp.cmp(os.path.join(rwork, "crosspoints", f), os.path.join(work, "crosspoints", f), shallow=False), f
    assert out["text"] == ref["alignment_txt"]
    assert open(os.path.join(work, "alignment.00.bin"), "rb").read() == open(os.path.join(rwork, "alignment.00.bin"), "rb").read()
    last_row = "%08X" % (q0.offset1 - (q0.offset0 - 1))      # the native stage 1's completion marker (manager.py)
    for d in sorted(os.listdir(os.path.join(rwork, "special_rows"))):
        p = subprocess.run(["diff", "-rq", "-x", last_row, os.path.join(rwork, "special_rows", d),
                            os.path.join(work, "special_rows", d)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert p.returncode == 0, p.stdout.decode()[:2000]
    if "trim" in name:
        assert b"[" in out["text"].split(b"\n")[0]           # "Query: s0 [201..2500](2300)"


def test_native_stages_continue_a_work_directory_of_masa_core(pkg, oracle, tmp_path):
    """the formats are shared, so either side can pick up where the other stopped: stage 2 natively on the special
    rows and the crosspoint MASA-Core's stage 1 wrote (block pruning on), stage 3 natively on what MASA-Core's stage 2
    wrote -- each reproduces the file MASA-Core's own next stage produced"""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    import shutil
    from oracle.aligner_double import SerialBlockAligner
    from masa_cudalign_amd.stage2 import stage2
    from masa_cudalign_amd.stage3 import stage3
    s0, s1 = pkg.seqgen.related_pair(12000, 12000, cfg=21, p_indel=0.02, indel_mean=6.0)
    limit = 300 * 1024
    refdir = tmp_path / "ref"
    refdir.mkdir()
    oracle.run_ref(s0, s1, ["--disk-size=%d" % limit, "--block=128,128"], workdir=str(refdir), timeout=600)
    rwork = str(refdir / "work")
    want2 = open(os.path.join(rwork, "crosspoints", "crosspoint_02.00"), "rb").read()
    want3 = open(os.path.join(rwork, "crosspoints", "crosspoint_03.00"), "rb").read()
    al = SerialBlockAligner(128, 128)
    # (a) stage 3 on MASA-Core's stage-2 output
    wa = str(tmp_path / "a")
    shutil.copytree(rwork, wa)
    shutil.rmtree(os.path.join(wa, "special_rows", "stage.03.00.r01"))
    for f in os.listdir(os.path.join(wa, "crosspoints")):
        if f.startswith("crosspoint_03") or f.startswith("crosspoint_04"):
            os.remove(os.path.join(wa, "crosspoints", f))
    stage3(al, s0, s1, wa, sra_limit=limit)
    assert open(os.path.join(wa, "crosspoints", "crosspoint_03.00"), "rb").read() == want3
    # (b) stages 2 and 3 on MASA-Core's stage-1 output
    wb = str(tmp_path / "b")
    shutil.copytree(rwork, wb)
    for d in os.listdir(os.path.join(wb, "special_rows")):
        if not d.startswith("stage.01"):
            shutil.rmtree(os.path.join(wb, "special_rows", d))
    for f in os.listdir(os.path.join(wb, "crosspoints")):
        if not f.startswith("crosspoint_01"):
            os.remove(os.path.join(wb, "crosspoints", f))
    stage2(al, s0, s1, wb, sra_limit=limit)
    assert open(os.path.join(wb, "crosspoints", "crosspoint_02.00"), "rb").read() == want2
    stage3(al, s0, s1, wb, sra_limit=limit)
    assert open(os.path.join(wb, "crosspoints", "crosspoint_03.00"), "rb").read() == want3


def _random_case(pkg, rng):
    """one random configuration of everything the native drivers take: pair kind and divergence, sizes, block geometry,
    special-rows budgets on disk and in memory, alignment edges, --max-alignments, --trim"""
    g = pkg.seqgen
    m, n, cfg = rng.randint(600, 6000), rng.randint(600, 6000), rng.randint(1, 10000)
    kind = rng.choice(["related", "related", "related", "unrelated", "contained"])
    if kind == "related":
        s0, s1 = g.related_pair(m, n, cfg=cfg, p_sub=rng.choice([0.02, 0.1, 0.2]), p_indel=rng.choice([0.002, 0.02, 0.05]),
                                indel_mean=rng.choice([2.0, 8.0, 30.0]), inversion=rng.choice([0.0, 0.05]))
    elif kind == "unrelated":
        s0, s1 = g.unrelated_pair(m, n, cfg=cfg)
    else:
        a = g.random_dna(cfg, min(m, n) // 2)
        s0 = np.concatenate([g.random_dna(cfg + 1, rng.randint(0, m // 2)), a, g.random_dna(cfg + 2, rng.randint(0, m // 3))])
        s1 = np.concatenate([g.random_dna(cfg + 3, rng.randint(0, n // 3)), g.mutate_dna(a, cfg + 4, inversion=0.0),
                             g.random_dna(cfg + 5, rng.randint(0, n // 2))])
    case = dict(s0=s0, s1=s1, bh=rng.choice([64, 100, 128, 256, 300, 1024]), bw=rng.choice([64, 128, 200, 512, 2048]),
                limit=rng.choice([0, 20 * 1024, 60 * 1024, 150 * 1024, 400 * 1024, 2 * 1024 * 1024]),
                edges=rng.choice(["**", "**", "**", "++", "13", "31", "21", "12", "22", "11", "33", "2*", "3*"]),
                count=rng.choice([1, 1, 1, 2, 3]), mod0={}, mod1={}, flags=[], ram=rng.choice([0, 0, 50 * 1024, 200 * 1024]))
    if rng.random() < 0.3:
        a0, b0 = rng.randint(1, len(s0) // 3), rng.randint(2 * len(s0) // 3, len(s0))
        a1, b1 = rng.randint(1, len(s1) // 3), rng.randint(2 * len(s1) // 3, len(s1))
        case["mod0"], case["mod1"] = dict(trim_start=a0, trim_end=b0), dict(trim_start=a1, trim_end=b1)
        case["flags"] = ["--trim=%d,%d,%d,%d" % (a0, b0, a1, b1)]
    return case


@pytest.mark.parametrize("seed", range(12))
def test_random_configurations_against_the_live_reference(seed, pkg, oracle, tmp_path):
    """differential fuzz (160 such cases were run when the drivers were written, none differed): every crosspoint file
    and every alignment.NN.txt equal MASA-Core's"""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    import random
    from oracle.aligner_double import SerialBlockAligner
    from masa_cudalign_amd import fasta, pipeline
    c = _random_case(pkg, random.Random(1000 + seed))
    edge = {"*": pkg.AT_ANYWHERE, "1": pkg.AT_SEQUENCE_1, "2": pkg.AT_SEQUENCE_2, "3": pkg.AT_SEQUENCE_1_OR_2,
            "+": pkg.AT_SEQUENCE_1_AND_2}
    args = ["--disk-size=%d" % c["limit"], "--ram-size=%d" % c["ram"], "--block=%d,%d" % (c["bh"], c["bw"]),
            "--no-block-pruning", "--max-alignments=%d" % c["count"]] + c["flags"]
    if c["edges"] != "**":
        args.append("--edges=" + c["edges"])
    refdir = tmp_path / "ref"
    refdir.mkdir()
    oracle.run_ref(c["s0"], c["s1"], args, workdir=str(refdir), timeout=600)
    rwork = str(refdir / "work")
    q0 = fasta.parse(b">s0\n" + c["s0"].tobytes() + b"\n", fasta.SequenceModifiers(**c["mod0"]))
    q1 = fasta.parse(b">s1\n" + c["s1"].tobytes() + b"\n", fasta.SequenceModifiers(**c["mod1"]))
    work = str(tmp_path / "native")
    pipeline.align(SerialBlockAligner(c["bh"], c["bw"]), q0, q1, work, alignment_start=edge[c["edges"][0]],
                   alignment_end=edge[c["edges"][1]], sra_limit=c["limit"], ram_limit=c["ram"], block_pruning=False,
                   max_alignments=c["count"])
    rc = os.path.join(rwork, "crosspoints")
    want = sorted(os.listdir(rc)) if os.path.isdir(rc) else []
    got = sorted(os.listdir(os.path.join(work, "crosspoints"))) if os.path.isdir(os.path.join(work, "crosspoints")) else []
    assert got == want, (args, want, got)
    for f in want:
        assert filecmp.cmp(os.path.join(rc, f), os.path.join(work, "crosspoints", f), shallow=False), (args, f)
    for f in sorted(x for x in os.listdir(rwork) if x.startswith("alignment.") and x.endswith(".txt")):
        assert filecmp.cmp(os.path.join(rwork, f), os.path.join(work, f), shallow=False), (args, f)


@pytest.mark.parametrize("name,ram,disk", [("half_and_half", 150 * 1024, 150 * 1024), ("memory_only", 300 * 1024, 0),
                                           ("one_third_memory", 100 * 1024, 200 * 1024), ("mostly_memory", 250 * 1024, 50 * 1024)])
def test_special_rows_in_memory_and_on_disk(name, ram, disk, pkg, oracle, tmp_path):
    """--ram-size / --disk-size: rows alternate between memory and disk in the proportion of the two budgets
    (SpecialRowsPartition::getSpecialRow), the spacing follows their sum, and the later stages find the rows kept in
    memory through the shared area objects (Job::getSpecialRowsArea).  Same rows on disk as MASA-Core -- names and
    bytes, in every stage's directory --, same crosspoint files, same text"""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    from oracle.aligner_double import SerialBlockAligner
    from masa_cudalign_amd import pipeline
    s0, s1 = pkg.seqgen.related_pair(12000, 12000, cfg=21, p_indel=0.02, indel_mean=6.0)
    refdir = tmp_path / "ref"
    refdir.mkdir()
    ref = oracle.run_ref(s0, s1, ["--block=128,128", "--no-block-pruning", "--ram-size=%d" % ram, "--disk-size=%d" % disk],
                         workdir=str(refdir), timeout=600)
    rwork = str(refdir / "work")
    q0, q1 = _fasta(pkg, s0, s1)
    work = str(tmp_path / "native")
    out = pipeline.align(SerialBlockAligner(128, 128), q0, q1, work, sra_limit=disk, ram_limit=ram, block_pruning=False)
    for f in sorted(os.listdir(os.path.join(rwork, "crosspoints"))):
        assert filecmp.cmp(os.path.join(rwork, "crosspoints", f), os.path.join(work, "crosspoints", f), shallow=False), f
    assert out["text"] == ref["alignment_txt"]
    on_disk = 0
    for d in sorted(os.listdir(os.path.join(rwork, "special_rows"))):
        p = subprocess.run(["diff", "-rq", "-x", "%08X" % len(s0), os.path.join(rwork, "special_rows", d),
                            os.path.join(work, "special_rows", d)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert p.returncode == 0, p.stdout.decode()[:2000]
        for _root, _dirs, files in os.walk(os.path.join(rwork, "special_rows", d)):
            on_disk += sum(1 for f in files if len(f) == 8)
    assert (on_disk == 0) == (disk == 0)
    assert len(out["stage2"]["crosspoints"]) > 3              # the rows in memory were found: the traceback used them


def _three_segments(pkg, seed=5):
    g = pkg.seqgen
    a, b, c = g.random_dna(100 + seed, 2000), g.random_dna(200 + seed, 1500), g.random_dna(300 + seed, 900)
    s0 = np.concatenate([a, g.random_dna(400 + seed, 500), b, g.random_dna(401 + seed, 300), c])
    s1 = np.concatenate([g.mutate_dna(b, 500 + seed, inversion=0.0), g.random_dna(402 + seed, 400),
                         g.mutate_dna(c, 501 + seed, inversion=0.0), g.random_dna(403 + seed, 200),
                         g.mutate_dna(a, 502 + seed, inversion=0.0)])
    return s0, s1


@pytest.mark.parametrize("name,pair,count", [
    ("shuffled_segments_2", _three_segments, 2), ("shuffled_segments_5", _three_segments, 5),
    ("unrelated_4", lambda pkg: pkg.seqgen.unrelated_pair(4000, 4000, cfg=8), 4)])
def test_several_alignments(name, pair, count, pkg, oracle, tmp_path):
    """--max-alignments=N: the best-score list keeps end points of different alignments (BestScoreList: shadowed and
    weak candidates dropped, a better newcomer evicts what it shadows), stage 1 writes one crosspoint_01.NN each and the
    traceback runs once per end point -- same crosspoint files and the same alignment.NN.txt as MASA-Core"""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    from oracle.aligner_double import SerialBlockAligner
    from masa_cudalign_amd import fasta, pipeline
    s0, s1 = pair(pkg)
    limit = 150 * 1024
    refdir = tmp_path / "ref"
    refdir.mkdir()
    oracle.run_ref(s0, s1, ["--disk-size=%d" % limit, "--block=128,128", "--no-block-pruning", "--max-alignments=%d" % count],
                   workdir=str(refdir), timeout=600)
    rwork = str(refdir / "work")
    q0, q1 = _fasta(pkg, s0, s1)
    work = str(tmp_path / "native")
    out = pipeline.align(SerialBlockAligner(128, 128), q0, q1, work, sra_limit=limit, block_pruning=False, max_alignments=count)
    want = sorted(os.listdir(os.path.join(rwork, "crosspoints")))
    assert sorted(os.listdir(os.path.join(work, "crosspoints"))) == want
    for f in want:
        assert filecmp.cmp(os.path.join(rwork, "crosspoints", f), os.path.join(work, "crosspoints", f), shallow=False), f
    texts = sorted(f for f in os.listdir(rwork) if f.startswith("alignment.") and f.endswith(".txt"))
    assert len(texts) == len(out["alignments"]) == len(out["stage1"]["bests"]) >= (2 if count > 1 else 1)
    for f in texts:
        assert filecmp.cmp(os.path.join(rwork, f), os.path.join(work, f), shallow=False), f
    scores = [b[2] for b in out["stage1"]["bests"]]
    assert scores == sorted(scores, reverse=True) and len(scores) <= count


def test_best_score_list_rules(pkg):
    from masa_cudalign_amd.manager import BestScoreList
    b = BestScoreList(0, limit=3, seq0_len=10000, seq1_len=10000)
    b.add(500, 500, 400)
    b.add(490, 490, 390)            # ten cells up the same diagonal, ten points less: the same alignment
    assert b.all() == [(500, 500, 400)]
    b.add(3000, 7000, 90)           # under a quarter of the best: not worth a traceback
    assert b.all() == [(500, 500, 400)]
    b.add(3000, 7000, 300)
    b.add(8000, 2000, 300)          # ties order by row, then column
    assert b.all() == [(500, 500, 400), (3000, 7000, 300), (8000, 2000, 300)]
    b.add(9000, 9000, 250)          # list full, worse than its last entry
    assert len(b.all()) == 3 and b.getBestScore() == (500, 500, 400)
    b.add(510, 510, 410)            # a better end of the first alignment evicts the one it shadows
    assert b.all()[0] == (510, 510, 410) and (500, 500, 400) not in b.all()
    one = BestScoreList(0)
    for cell in ((7, 9, 20), (5, 9, 20), (5, 3, 20), (6, 1, 19)):
        one.add(*cell)
    assert one.all() == [(5, 3, 20)]                   # limit 1: max score, min i, min j
    assert BestScoreList(5).getBestScore()[2] < -10 ** 8 and BestScoreList(5).best is None


def test_work_directory_belongs_to_its_sequences(pkg, oracle, tmp_path):
    """<work>/info (Job.cpp:68-90): a second run with other sequences must not continue from the first one's files"""
    from oracle.aligner_double import SerialBlockAligner
    from masa_cudalign_amd import fasta, pipeline
    s0, s1 = pkg.seqgen.related_pair(1500, 1400, cfg=5)
    q0, q1 = fasta.parse(b">chr1 first\n" + s0.tobytes() + b"\n"), fasta.parse(b">chr2\n" + s1.tobytes() + b"\n")
    work = str(tmp_path / "work")
    out = pipeline.align(SerialBlockAligner(128, 128), q0, q1, work, sra_limit=64 * 1024, block_pruning=False)
    assert open(os.path.join(work, "info")).read() == "seq0=chr1 first\nseq1=chr2\n"
    again = pipeline.align(SerialBlockAligner(128, 128), q0, q1, work, sra_limit=64 * 1024, block_pruning=False)
    assert again["text"] == out["text"] and again["stage1"].get("already_done")
    other = fasta.parse(b">chr3\n" + s1.tobytes() + b"\n")
    with pytest.raises(pipeline.WorkDirectoryMismatch):
        pipeline.align(SerialBlockAligner(128, 128), q0, other, work, sra_limit=64 * 1024)


def test_alignment_binary_file(pkg):
    """alignment.NN.bin (AlignmentBinaryFile.cpp): header, big-endian fields, gap lists as position deltas in 7-bit
    groups; written and read back"""
    from masa_cudalign_amd import alignment_file as af, fasta, stage56
    assert [af._u4c(v) for v in (0, 127, 128, 16383, 16384, 0x0FFFFFFF, 0x10000000, 0xFFFFFFFF)] == [
        b"\x00", b"\x7f", b"\x81\x00", b"\xff\x7f", b"\x81\x80\x00", b"\xff\xff\xff\x7f", b"\x81\x80\x80\x80\x00",
        b"\x8f\xff\xff\xff\x7f"]
    q0 = fasta.parse(b">first one\nACGTACGTAC\n", fasta.SequenceModifiers(trim_start=2, trim_end=9, reverse=True))
    q1 = fasta.parse(b">second\nACGTTACGTAC\n", fasta.SequenceModifiers(complement=True, clear_n=True))
    al = stage56.Alignment()
    al.start, al.end = [2, 1], [9, 11]
    al.raw_score, al.matches, al.mismatches, al.gap_open, al.gap_extensions = 3, 8, 0, 1, 1
    al.gaps = ([[5, 1]], [[300, 2], [70000, 129]])
    data = af.dumps(al, q0, q1)
    assert data[:6] == b"CGFF\x00\x01"
    d = af.loads(data)
    assert [s["description"] for s in d["sequences"]] == ["first one", "second"] and [s["size"] for s in d["sequences"]] == [10, 11]
    assert d["params"]["method"] == 2 and (d["params"]["match"], d["params"]["mismatch"]) == (1, -3)
    assert (d["params"]["gap_open"], d["params"]["gap_ext"]) == (-3, -2)
    assert d["params"]["sequences"][0] == {"index": 0, "reverse": True, "complement": False, "clear_n": False,
                                           "trim_start": 2, "trim_end": 9}
    assert d["params"]["sequences"][1]["complement"] and d["params"]["sequences"][1]["clear_n"]
    assert (d["params"]["sequences"][1]["trim_start"], d["params"]["sequences"][1]["trim_end"]) == (1, 11)
    r = d["result"]
    assert (r["raw_score"], r["matches"], r["gap_open"], r["gap_extensions"]) == (3, 8, 1, 1)
    assert r["start"] == [2, 1] and r["end"] == [9, 11] and r["gaps"] == [[[5, 1]], [[300, 2], [70000, 129]]]
    with pytest.raises(ValueError):
        af.loads(b"XXXX" + data[4:])
    with pytest.raises(ValueError):
        af.loads(data[:-3])


def test_nothing_to_trace_back(pkg, oracle, tmp_path):
    """a global start with a local end and nothing above the floor: MASA-Core's best-score list stays empty, it writes
    no crosspoint file and runs no traceback; neither does the native pipeline"""
    from masa_cudalign_amd import pipeline
    from oracle.aligner_double import SerialBlockAligner
    s0, s1 = _contained(pkg)
    q0, q1 = _fasta(pkg, s0, s1)
    work = str(tmp_path / "work")
    out = pipeline.align(SerialBlockAligner(128, 128), q0, q1, work, alignment_start=pkg.AT_SEQUENCE_1_AND_2,
                         alignment_end=pkg.AT_ANYWHERE, sra_limit=150 * 1024, block_pruning=False)
    assert out["text"] is None and out["alignment"] is None
    assert not os.path.exists(os.path.join(work, "crosspoints", "crosspoint_01.00"))


def test_flush_intervals_follow_the_reference_arithmetic(pkg):
    """Job::calculateFlushIntervals: integer arithmetic for stage 1, single precision from stage 2 on, every round at
    most half of the round before last.  317 / 34 are what MASA-Core printed for the 3000 x 2700 fixture
    (statistics_02.00: "Flush Interval: 34"), 6868 / 786 for 60000 x 50000 with 4 MiB."""
    from masa_cudalign_amd import sra
    assert sra.flush_intervals(3000, 2700, 200 * 1024)[:4] == [317, 34, 4, 1]
    assert sra.flush_interval(3000, 2700, 200 * 1024) == 317
    f = sra.flush_intervals(40000, 30000, 1024)          # limit below two rows: raised to two rows
    assert f[:3] == [20001, 10001, 6668] and all(b <= a for a, b in zip(f, f[1:]))
    f = sra.flush_intervals(27648, 27648, 663552)
    assert f[0] == 9217 and f[1] >= 1024 > f[3]
    # 48 M x 46 M with 25 GB (BASELINE C3): single precision decides the last digits
    f = sra.flush_intervals(48000000, 46000000, 25 * 1000 ** 3)
    assert f[0] == 48000000 * 46000000 * 8 // (25 * 1000 ** 3) + 1
    assert f[1] == int(np.float32(f[0] * 46000000 * 8) / np.float32(25 * 1000 ** 3) + np.float32(1))


def test_crosspoints_file_and_reversal(pkg, tmp_path):
    from masa_cudalign_amd.crosspoints import Crosspoint, CrosspointsFile, TYPE_GAP_1, TYPE_GAP_2
    c = Crosspoint(10, 20, 7, TYPE_GAP_1)
    r = c.reverse(100, 200)
    assert r.astuple() == (TYPE_GAP_2, 180, 90, 7) and r.reverse(200, 100) == c
    fn = str(tmp_path / "crosspoints" / "crosspoint_02.00")
    f = CrosspointsFile(fn).open()
    f.write(Crosspoint(0, 0, 0, 0))
    assert os.path.exists(fn + ".tmp") and not os.path.exists(fn)          # visible under its name only once closed
    f.write(c)
    f.close()
    assert open(fn).read() == "START\n0,0,0,0\n1,10,20,7\nEND\n"
    g = CrosspointsFile(fn).load()
    assert g.tuples() == [(0, 0, 0, 0), (1, 10, 20, 7)]
    g.reverse_all(100, 200)
    assert g.tuples() == [(2, 180, 90, 7), (0, 200, 100, 0)]
    g.save()
    assert CrosspointsFile(fn).load().tuples() == g.tuples()
    assert CrosspointsFile(str(tmp_path / "missing")).load() == []


def test_special_rows_are_read_backwards(pkg, tmp_path):
    """a row written left to right by one stage is read right to left by the next: seek(j - j0 + 1), then every read
    hands over the cells in reversed order; row 0 is the partition's first row, made from its border marker"""
    from masa_cudalign_amd import sra
    from masa_cudalign_amd.manager import InitialCellsReader, ReversedCellsReader
    area = sra.SpecialRowsArea(str(tmp_path))
    p = area.create_partition(100, 50, 400, 60)
    p.set_first_row_reader(InitialCellsReader(3, 2))
    p.set_first_column_reader(InitialCellsReader(0, 2))
    cells = np.arange(22, dtype=np.int32).reshape(11, 2)
    for i in (228, 356):
        assert p.write(i, cells[:4] + i) is False
        assert p.write(i, cells[4:] + i) is True
    assert sorted(os.listdir(p.path)) == ["00000080", "00000100", "C00000000.INIT_WITH_GAPS_OPENED", "R00000000.INIT_WITH_GAPS"]
    q = sra.SpecialRowsArea(str(tmp_path)).open_partition_at(101, 51)
    assert (q.i0, q.j0, q.i1, q.j1) == (100, 50, 400, 60) and q.rows_count() == 3 and q.largest_interval == 128
    assert q.first_row_reader.getType() == pkg.INIT_WITH_GAPS and q.first_column_reader.getType() == pkg.INIT_WITH_GAPS_OPENED
    # from DP cell (390, 57): row 356 is closer than 128 rows and skipped; row 228 is the one
    r = q.next_special_row(390, 57, 128)
    assert q.get_reading_row() == 228 and r.getOffset() == 8
    buf = np.empty((3, 2), dtype=np.int32)
    assert r.read(buf, 3) == 3 and np.array_equal(buf, (cells[5:8] + 228)[::-1])
    buf = np.empty((9, 2), dtype=np.int32)
    assert r.read(buf, 9) == 5 and np.array_equal(buf[:5], (cells[:5] + 228)[::-1])      # clamped at the row's start
    with pytest.raises(RuntimeError):
        r.read(buf, 1)
    # from (229, 55): only the first row is left, -2k-3 with the corner 0
    r = q.next_special_row(229, 55, 128)
    assert q.get_reading_row() == 100
    buf = np.empty((6, 2), dtype=np.int32)
    assert r.read(buf, 6) == 6 and list(buf[:, 0]) == [-13, -11, -9, -7, -5, 0]
    assert q.next_special_row(100, 55, 128) is None
    # truncation at a crosspoint: rows at or below it go, the others are cut to its column
    area.truncate_partition(p, 300, 55)
    assert os.path.basename(p.path) == "00000064.00000032.0000012C.00000037"
    assert sorted(f for f in os.listdir(p.path) if len(f) == 8) == ["00000080"]
    assert os.path.getsize(os.path.join(p.path, "00000080")) == 6 * 8
    assert area.rows_count() == 2 and area.partitions_count() == 1
    # a border read backwards
    col = ReversedCellsReader(InitialCellsReader(3, 2))
    col.seek(4)
    buf = np.empty((4, 2), dtype=np.int32)
    assert col.read(buf, 10) == 4 and list(buf[:, 0]) == [-9, -7, -5, 0]
    # a non-persistent area swallows rows and counts nothing
    vol = sra.SpecialRowsArea(str(tmp_path / "none"), persistent=False)
    vp = vol.create_partition(0, 0, 10, 10)
    vp.set_first_row_reader(InitialCellsReader(3, 2))
    assert vp.write(5, cells) is False and vol.rows_count() == 0 and not os.path.exists(str(tmp_path / "none"))


def test_manager_goal_matching(pkg, oracle):
    """AlignerManager: the full-gap shortcut before the sweep, the first matching cell of a dispatched column, the error
    on a border sum above the goal, and the start of a local alignment reported through the block scores"""
    from oracle.aligner_double import SerialBlockAligner
    from masa_cudalign_amd.manager import (AlignerManager, ArrayCellsReader, InitialCellsReader, BacktraceLost,
                                           START_TYPE_MATCH, START_TYPE_GAP_H)
    INF = pkg.INF
    al = SerialBlockAligner(64, 64)
    mg = AlignerManager(al)

    class Part:                                          # what a stage hands over per partition: fresh border streams
        persistent, last_column_writer, last_row_writer = False, None, None

        def __init__(self):
            self.first_row_reader, self.first_column_reader = InitialCellsReader(3, 2), InitialCellsReader(3, 2)
    mg.setSpecialRowsPartition(Part())
    mg.setRecurrenceType(pkg.NEEDLEMAN_WUNSCH)
    # full gap: the forward F of the border cell + a run of 5 columns (opened: the start type is not a gap along S1)
    base = np.array([[50, 40]], dtype=np.int32)
    mg.setGoalScore(40 - 5 * 2 - 3 + 3, pkg.AT_SEQUENCE_1_OR_2)
    mg.setLastColumnReader(ArrayCellsReader(base))
    mg.setLastRowReader(None)
    mg.alignPartition(pkg.Partition(0, 0, 7, 5), START_TYPE_MATCH)
    assert mg.isFoundCrosspoint() and mg.getNextCrosspoint() == (0, 5, 40, 1) and al.partitions == 0
    # the same border with the gap already open on arrival
    mg.setGoalScore(40 - 5 * 2 + 3, pkg.AT_SEQUENCE_1_OR_2)
    mg.alignPartition(pkg.Partition(0, 0, 7, 5), START_TYPE_GAP_H)
    assert mg.isFoundCrosspoint() and al.partitions == 0
    # a real sweep: identical sequences, goal met on the last column where forward + reverse add up
    s = pkg.seqgen.random_dna(5, 40)
    mg.setSequences(s, s, 0, 0, 40, 40)
    fwd = np.zeros((41, 2), dtype=np.int32)
    fwd[:, 0] = 100 - np.arange(41)                      # forward H along the matched border
    fwd[:, 1] = -INF
    mg.setGoalScore(100 - 30 + 30, pkg.AT_SEQUENCE_1_OR_2)          # reverse H on the diagonal is +30 at row 30
    mg.setSpecialRowsPartition(Part())
    mg.setLastColumnReader(ArrayCellsReader(fwd))
    mg.alignPartition(pkg.Partition(0, 0, 40, 30), START_TYPE_MATCH)
    assert mg.isFoundCrosspoint() and mg.getNextCrosspoint() == (30, 30, 70, 0) and not mg.mustContinue()
    # a border that promises more than the goal: the reference exits with "Backtrace lost"
    mg.setGoalScore(20, pkg.AT_SEQUENCE_1_OR_2)
    mg.setSpecialRowsPartition(Part())
    mg.setLastColumnReader(ArrayCellsReader(fwd))
    with pytest.raises(BacktraceLost):
        mg.alignPartition(pkg.Partition(0, 0, 40, 30), START_TYPE_MATCH)
    # local start: no border to match, the block whose best reverse value is the whole goal
    mg.setGoalScore(40, pkg.AT_ANYWHERE)
    mg.setSpecialRowsPartition(Part())
    mg.setLastColumnReader(None)
    mg.alignPartition(pkg.Partition(0, 0, 40, 40), START_TYPE_MATCH)
    assert mg.isFoundCrosspoint() and mg.getNextCrosspoint() == (40, 40, 0, 0)
    mg.unsetSequences()


def test_crosspoint_04_written_by_the_file_thread_is_the_same_file(pkg, oracle, tmp_path):
    """an aligner whose stage4 hands back an ARRAY (the engine's form, millions of points at C3's size) has crosspoint_04 written
    by the areas' file thread while stages 5 and 6 run (pipeline._traceback): in place when align() returns, the same bytes as
    the list form written inline, and the operation really went through the queue"""
    from masa_cudalign_amd import pipeline, sra
    from masa_cudalign_amd.crosspoints import crosspoint_file
    from oracle.aligner_double import SerialBlockAligner
    case = [c for c in FULL if c["name"] == "full_pipeline_3000x2700"][0]
    s0, s1 = make_pair(pkg, case["seq"])
    q0, q1 = _fasta(pkg, s0, s1)

    class ArrayForm(SerialBlockAligner):
        def stage4(self, crosspoints, max_partition_size=16, as_array=False):
            out, st = SerialBlockAligner.stage4(self, crosspoints, max_partition_size)
            return (np.asarray(out, dtype=np.int32).reshape(-1, 4) if as_array else out), st

    queued = []
    submit = sra._files.submit

    def spy(owner, fn, *a, **kw):
        queued.append((fn.__name__, sra._files._enabled()))
        return submit(owner, fn, *a, **kw)
    sra._files.submit = spy
    try:
        work = str(tmp_path / "array")
        out = pipeline.align(ArrayForm(128, 128), q0, q1, work, sra_limit=_limit(case["args"]), block_pruning=False)
    finally:
        sra._files.submit = submit
    assert ("save_array", True) in queued
    assert not sra._files.has_pending()
    assert hashlib.sha256(open(crosspoint_file(work, 4), "rb").read()).hexdigest() == case["crosspoints_4"]["file_sha256"]
    assert hashlib.sha256(out["text"]).hexdigest() == case["alignment_txt_sha256"]


def test_stage3_walks_handed_over_together_give_the_same_files(pkg, oracle, tmp_path):
    """stage 3 hands the pending sweeps of all its walks to the aligner in ONE call when the aligner offers alignPartitions
    (MI355Aligner: one kernel launch) and one by one otherwise (stage3._run_walks).  The CPU double with an alignPartitions
    that serves its partitions in turn: the batched branch runs here too -- same crosspoint files and the same special rows as
    the one-by-one branch, over two rounds of stage 3, and the batches really held several partitions."""
    from oracle.aligner_double import SerialBlockAligner
    from masa_cudalign_amd.stage1 import stage1
    from masa_cudalign_amd.stage2 import stage2
    from masa_cudalign_amd.stage3 import stage3
    batches = []

    class Batched(SerialBlockAligner):
        def alignPartitions(self, partitions, managers):
            assert len(partitions) == len(managers) and len({id(g) for g in managers}) == len(managers)
            batches.append(len(partitions))
            for part, mgr in zip(partitions, managers):
                self.alignPartition(part, mgr)

    s0, s1 = pkg.seqgen.related_pair(27648, 27648, cfg=7)
    limit = 663552
    out = {}
    for name, cls in (("one_by_one", SerialBlockAligner), ("batched", Batched)):
        work = str(tmp_path / name)
        al = cls(256, 256)
        stage1(al, s0, s1, work, sra_limit=limit, block_pruning=False)
        stage2(al, s0, s1, work, sra_limit=limit)
        out[name] = (work, stage3(al, s0, s1, work, sra_limit=limit))
    assert len(out["batched"][1]["rounds"]) == 2 and max(batches) > 1
    assert out["batched"][1]["crosspoints"] == out["one_by_one"][1]["crosspoints"]
    assert out["batched"][1]["rounds"] == out["one_by_one"][1]["rounds"]
    a, b = out["one_by_one"][0], out["batched"][0]
    for f in sorted(os.listdir(os.path.join(a, "crosspoints"))):
        assert filecmp.cmp(os.path.join(a, "crosspoints", f), os.path.join(b, "crosspoints", f), shallow=False), f
    p = subprocess.run(["diff", "-rq", os.path.join(a, "special_rows"), os.path.join(b, "special_rows")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0, p.stdout.decode()[:2000]


SPEC = [
    ("related", lambda pkg: pkg.seqgen.related_pair(27648, 27648, cfg=7), 256, 256, 663552, "**"),
    ("many_indels", lambda pkg: pkg.seqgen.related_pair(12000, 12000, cfg=21, p_indel=0.02, indel_mean=6.0), 128, 128, 300 * 1024, "**"),
    ("wide_long_indels", lambda pkg: pkg.seqgen.related_pair(9000, 14000, cfg=22, p_indel=0.01, indel_mean=20.0), 256, 128, 200 * 1024, "**"),
    ("gap_rich", lambda pkg: _gap_rich(), 128, 128, 100 * 1024, "**"),
    ("global", lambda pkg: pkg.seqgen.related_pair(6000, 6100, cfg=23, inversion=0.0), 128, 128, 150 * 1024, "++"),
    ("contained_semiglobal_21", _contained, 128, 128, 150 * 1024, "21"),
    ("unrelated", lambda pkg: pkg.seqgen.unrelated_pair(5000, 5000, cfg=3), 128, 128, 100 * 1024, "**"),
]


def _same_tree(a, b, sub):
    p = subprocess.run(["diff", "-rq", os.path.join(a, sub), os.path.join(b, sub)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0, p.stdout.decode()[:2000]


@pytest.mark.parametrize("name,pair,bh,bw,limit,edges", SPEC, ids=[x[0] for x in SPEC])
def test_stage2_from_guessed_crosspoints_leaves_the_same_files(name, pair, bh, bw, limit, edges, pkg, oracle, tmp_path, monkeypatch):
    """stage 2 with its sweeps started side by side from GUESSED crosspoints (stage2._Speculation: the row maxima of stage 1's
    special rows) against the plain chain: the same crosspoint_02, the same special rows for stage 3 (and so the same stage 3),
    whether the aligner takes the sweeps together (alignPartitions) or one by one; most guesses are right on related pairs;
    and with every guess made wrong on purpose the walk falls back to the plain step, throws the sweeps away and still
    leaves the same files."""
    from oracle.aligner_double import SerialBlockAligner
    from masa_cudalign_amd.stage1 import stage1
    from masa_cudalign_amd.stage2 import stage2
    from masa_cudalign_amd.stage3 import stage3
    from masa_cudalign_amd import sra
    edge = {"*": pkg.AT_ANYWHERE, "1": pkg.AT_SEQUENCE_1, "2": pkg.AT_SEQUENCE_2, "3": pkg.AT_SEQUENCE_1_OR_2,
            "+": pkg.AT_SEQUENCE_1_AND_2}
    batches = []

    class Batched(SerialBlockAligner):
        def alignPartitions(self, partitions, managers):
            batches.append(len(partitions))
            for part, mgr in zip(partitions, managers):
                self.alignPartition(part, mgr)

    s0, s1 = pair(pkg)
    runs = {}
    monkeypatch.delenv("MI355SW_STAGE2_SPECULATE", raising=False)   # (the whole file also passes with the variable set: every
    #                                                                    other test then walks stage 2 from guesses)
    for mode in ("plain", "guessed", "guessed_batched", "recorded_peaks", "all_guesses_wrong", "cut_partitions"):
        work = str(tmp_path / mode)
        al = (Batched if mode == "guessed_batched" else SerialBlockAligner)(bh, bw)
        areas = {}
        # (round 6: guessing is the default and stage 1 records the row maxima as it writes; with the variable at 0 it does not,
        #  and a stage 2 that is asked to guess all the same reads them back from the rows)
        monkeypatch.setenv("MI355SW_STAGE2_SPECULATE", "1" if mode == "recorded_peaks" else "0")
        if mode == "cut_partitions":
            # guessed sweeps whose partitions END a few rows below where they give up (at C3 the cut lies 1.4 M rows down a
            # 22 M-row partition; at these sizes only a shrunken slack makes it bite): a goal beyond the cut is a sweep given
            # up, the walk makes that step itself -- the files do not change
            import sys
            stage2_mod = sys.modules["masa_cudalign_amd.stage2"]
            monkeypatch.setattr(stage2_mod, "GUESS_CAP_SLACK", 8)
            monkeypatch.setattr(stage2_mod, "GUESS_PARTITION_SLACK", 0)
        if mode == "all_guesses_wrong":
            real = sra.SpecialRowsPartition.row_peak

            def off_by_some(self, rid, max_index):
                p = real(self, rid, max_index)
                return None if p is None else (p[0], max(1, p[1] - 7))
            monkeypatch.setattr(sra.SpecialRowsPartition, "row_peak", off_by_some)
        stage1(al, s0, s1, work, alignment_start=edge[edges[0]], alignment_end=edge[edges[1]], sra_limit=limit, block_pruning=False,
               areas=areas)
        r2 = stage2(al, s0, s1, work, alignment_start=edge[edges[0]], sra_limit=limit, areas=areas,
                    speculate=False if mode == "plain" else None if mode == "recorded_peaks" else True)
        r3 = stage3(al, s0, s1, work, sra_limit=limit, areas=areas)
        monkeypatch.delenv("MI355SW_STAGE2_SPECULATE", raising=False)
        monkeypatch.undo()
        runs[mode] = (work, r2, r3)
    plain = runs["plain"]
    assert plain[1]["speculation"] is None
    for mode in ("guessed", "guessed_batched", "recorded_peaks", "all_guesses_wrong", "cut_partitions"):
        work, r2, r3 = runs[mode]
        assert r2["crosspoints"] == plain[1]["crosspoints"] and r2["partitions"] == plain[1]["partitions"], mode
        assert r3["crosspoints"] == plain[2]["crosspoints"], mode
        _same_tree(plain[0], work, "crosspoints")
        _same_tree(plain[0], work, os.path.join("special_rows", "stage.02.00"))
        sp = r2["speculation"]
        assert sp["accepted"] + sp["discarded"] == sp["sweeps"], (mode, sp)
        if mode == "all_guesses_wrong":
            assert sp["accepted"] <= 1 + (sp["sweeps"] > 0), (mode, sp)     # only sweeps from real crosspoints count
    sp = runs["guessed"][1]["speculation"]
    if name in ("related", "many_indels", "global"):
        assert sp["sweeps"] >= 3 and sp["accepted"] >= sp["sweeps"] - 2, sp  # nearly every guess was the crosspoint
    if runs["guessed_batched"][1]["speculation"]["sweeps"] > 1:
        assert max(batches) > 1
    assert runs["recorded_peaks"][1]["speculation"] == runs["guessed"][1]["speculation"]

"""GPU (-m gpu): END-TO-END DROP-IN.  oracle/_ref/masa_mi355 is the reference's own MASA-Core (stages 1-6,
AlignerManager, SRA, crosspoint files -- compiled from /root/reference, unmodified) driven by the product's
IAligner adapter (masa-cudalign_amd/host/Mi355Aligner.cpp -> C ABI -> HIP engine).  Its outputs must be
byte-identical to the fixture the reference's CPU aligner produced for the same pair: that covers stage 1
(SW, special rows on disk), the stage-2/3 re-entry (NW, custom borders, last-column goal matching, early
stop) and everything downstream."""
import hashlib
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

import __graft_entry__ as graft
from helpers import load_golden, make_pair, digest

pytestmark = pytest.mark.gpu
BIN = os.path.join(graft.ROOT, "oracle", "_ref", "masa_mi355")
G = load_golden()


def _run(pkg, oracle, seq, args):
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/masa_mi355 not prebuilt (needs /root/reference at build time)")
    s0, s1 = make_pair(pkg, seq)
    tmp = tempfile.mkdtemp(prefix="masa_dropin_")
    try:
        from oracle.binding import _write_fasta, read_ref_work
        f0, f1 = os.path.join(tmp, "s0.fasta"), os.path.join(tmp, "s1.fasta")
        _write_fasta(f0, s0, "s0")
        _write_fasta(f1, s1, "s1")
        work = os.path.join(tmp, "work")
        p = subprocess.run([BIN, "--work-dir=" + work] + args + [f0, f1], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=600, cwd=tmp)
        assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
        return read_ref_work(work, log=p.stdout.decode(errors="replace"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


@pytest.mark.parametrize("name", ["full_pipeline_3000x2700_b8192", "full_pipeline_20000x9000_b8192"])
def test_full_pipeline_is_byte_identical(pkg, oracle, name):
    """the fixture was produced by the reference CPU aligner with the same special-row spacing as the engine
    (8192 rows, CUDAlign's MINIMUM_FLUSH_INTERVAL); with another spacing MASA-Core's later stages may pick a
    different, equally optimal traceback."""
    case = [c for c in G["cases"] if c["name"] == name][0]
    out = _run(pkg, oracle, case["seq"], ["--disk-size=200K"])
    assert list(out["best"]) == case["best"]
    assert out.get("crosspoints_2") == [tuple(x) for x in case["crosspoints_2"]]
    assert hashlib.sha256(out["alignment_txt"]).hexdigest() == case["alignment_txt_sha256"]


def test_full_pipeline_other_geometry_same_optimum(pkg, oracle):
    """against the fixture made with 128-row blocks: same score, same start and end of the alignment."""
    case = [c for c in G["cases"] if c["name"] == "full_pipeline_3000x2700"][0]
    out = _run(pkg, oracle, case["seq"], ["--disk-size=200K"])
    assert list(out["best"]) == case["best"]
    assert out["crosspoints_2"][0] == tuple(case["crosspoints_2"][0])
    assert out["crosspoints_2"][-1] == tuple(case["crosspoints_2"][-1])


@pytest.mark.parametrize("name", ["sw_unrelated_ties_20000x17000", "nw_global_3000x2700", "semiglobal_1to3_2500x2600",
                                  "sw_with_N_4000x4100"])
def test_stage1_through_masa_core(pkg, oracle, name):
    case = [c for c in G["cases"] if c["name"] == name][0]
    args = [a for a in case["args"] if not a.startswith("--block=")]
    out = _run(pkg, oracle, case["seq"], args)
    assert list(out["best"]) == case["best"]


def test_special_rows_written_by_masa_core_sra(pkg, oracle):
    """rows dispatched by the engine and written by the reference's SpecialRowsPartition/SpecialRowFile."""
    case = [c for c in G["cases"] if c["name"] == "sw_special_rows_20000x9000"][0]
    out = _run(pkg, oracle, case["seq"], ["--stage-1", "--disk-size=200K", "--no-block-pruning"])
    assert list(out["best"]) == case["best"]
    got = {i: a for (d, i), a in out["special_rows"].items()}
    # the block aligner that produced the fixture also flushes the bottom row of its last block row
    # (AbstractBlockAligner.cpp:418-439); CUDAlign's diagonal aligner -- the one this engine replaces --
    # never flushes rows >= height (AbstractDiagonalAligner.cpp:466-478)
    assert sorted(got) == sorted(int(k) for k in case["special_rows"] if int(k) < case["m"])
    for i, a in got.items():
        assert digest(a) == case["special_rows"][str(i)], i


def test_extension_command_line_options(pkg, oracle):
    """the extension's own options, handed over by MASA-Core like CUDAlign's (X/CUDAlignerParameters.cpp:33-110):
    --list-gpus, --gpu, --blocks and the strip height"""
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/masa_mi355 not prebuilt (needs /root/reference at build time)")
    p = subprocess.run([BIN, "--list-gpus", "a", "b"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 1 and "Available GPUs:" in out and "[fastest]" in out, out
    p = subprocess.run([BIN, "--gpu=-1", "a", "b"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert p.returncode == 2 and b"non-negative" in p.stdout
    case = [c for c in G["cases"] if c["name"] == "sw_unrelated_ties_20000x17000"][0]
    args = [a for a in case["args"] if not a.startswith("--block=")]
    res = _run(pkg, oracle, case["seq"], args + ["--gpu=0", "--blocks=64", "--strip-rows=512"])
    assert list(res["best"]) == case["best"]
    # the engine's own switches (the library reads no environment variable: ABI 7) -- two-phase best tracking + no mixed strip
    # heights + device-made gap columns, with one diagnostic line per partition
    res = _run(pkg, oracle, case["seq"], args + ["--engine-flags=0x1060", "--engine-verbosity=2"])
    assert list(res["best"]) == case["best"]
    assert "[mi355sw] job" in res["log"]
    # the index range is checked when the engine starts (after any fork), not while the options are parsed
    s0, s1 = make_pair(pkg, case["seq"])
    tmp = tempfile.mkdtemp(prefix="masa_gpu99_")
    try:
        from oracle.binding import _write_fasta
        f0, f1 = os.path.join(tmp, "s0.fasta"), os.path.join(tmp, "s1.fasta")
        _write_fasta(f0, s0, "s0")
        _write_fasta(f1, s1, "s1")
        p = subprocess.run([BIN, "--work-dir=" + os.path.join(tmp, "work"), "--gpu=99"] + args + [f0, f1], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=120, cwd=tmp)
        assert p.returncode == 2 and b"out of range" in p.stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_full_pipeline_with_block_pruning_biting(pkg, oracle):
    """stages 1-6 on a fixture the reference produced with block pruning ON and biting (143 of its 392 blocks pruned).
    The engine prunes too -- with its own granularity, so its special rows differ from the reference's off the optimal
    path -- and MASA-Core's stages 2-6 on top of it must still recover the same crosspoints and print the same
    alignment, byte for byte."""
    case = [c for c in G["cases"] if c["name"] == "full_pipeline_pruned_60000x50000_b8192"][0]
    assert case["pruned_blocks"][0] > 0
    out = _run(pkg, oracle, case["seq"], ["--disk-size=4M", "--strip-rows=1024"])
    assert list(out["best"]) == case["best"]
    stats = out["statistics"]["statistics_01.00"]
    pruned = [int(ln.split(":")[1]) for ln in stats.splitlines() if ln.startswith("Pruned cells:")]
    assert pruned and pruned[0] > 0.15 * case["m"] * case["n"], stats[-600:]
    assert out.get("crosspoints_2") == [tuple(x) for x in case["crosspoints_2"]]
    assert hashlib.sha256(out["alignment_txt"]).hexdigest() == case["alignment_txt_sha256"]
    # the special rows MASA-Core stored: same rows as the reference's, each a lower bound with the same maximum
    got = {i: a for (d, i), a in out["special_rows"].items()}
    assert sorted(got) == sorted(int(k) for k in case["special_rows"] if int(k) < case["m"])


def test_global_pipeline_with_pruning_through_masa_core(pkg, oracle):
    """--alignment-edges=++ (a global alignment) through MASA-Core's own stages 1-6 on the engine.  MASA-Core's stage 1 never
    asks for pruning then (sw_stage1.cpp:219-225); the extension's --prune-global makes the engine prune the partitions
    whose score is read from the last cell (AbstractBlockPruning.cpp:92-109, per slab).  More than a quarter of the
    matrix is skipped in stage 1, and MASA-Core's stages 2-6 on top of the lower-bound special rows recover the
    reference's crosspoints and print its alignment, byte for byte; without the option nothing is skipped."""
    case = [c for c in G["cases"] if c["name"] == "full_pipeline_global_60000x50000_b8192"][0]
    for extra, want_pruning in ((["--prune-global"], True), ([], False)):
        out = _run(pkg, oracle, case["seq"], ["--edges=++", "--disk-size=4M", "--strip-rows=1024"] + extra)
        assert list(out["best"]) == case["best"]
        stats = out["statistics"]["statistics_01.00"]
        pruned = [int(ln.split(":")[1]) for ln in stats.splitlines() if ln.startswith("Pruned cells:")]
        assert pruned, stats[-600:]
        if want_pruning:
            assert pruned[0] > 0.25 * case["m"] * case["n"], stats[-600:]
        else:
            assert pruned[0] == 0
        assert out.get("crosspoints_2") == [tuple(x) for x in case["crosspoints_2"]]
        assert hashlib.sha256(out["alignment_txt"]).hexdigest() == case["alignment_txt_sha256"]


def _run_forked(pkg, seq, args):
    from oracle.binding import _write_fasta, read_ref_work
    s0, s1 = make_pair(pkg, seq)
    tmp = tempfile.mkdtemp(prefix="masa_fork_")
    try:
        f0, f1 = os.path.join(tmp, "s0.fasta"), os.path.join(tmp, "s1.fasta")
        _write_fasta(f0, s0, "s0")
        _write_fasta(f1, s1, "s1")
        work = os.path.join(tmp, "work")
        p = subprocess.run([BIN, "--work-dir=" + work] + args + [f0, f1], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=600, cwd=tmp)
        log = p.stdout.decode(errors="replace")
        assert p.returncode == 0, log[-3000:]
        forks = sorted(d for d in os.listdir(work) if d.startswith("FORK."))
        return log, [read_ref_work(os.path.join(work, d)) for d in forks]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_fork_chain_through_masa_core(pkg, oracle):
    """MASA-Core's own multi-GPU mode (--fork: one process per GPU, seq1 split by the aligner's fork weights, boundary
    columns through its socket chain, best score relayed through AlignerPool) with the product adapter in every
    process.  The box has one GPU, so explicit weights give three processes whose fork ids wrap onto it; the running
    bests of the three bands are the reference chain's."""
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/masa_mi355 not prebuilt (needs /root/reference at build time)")
    ch = G["chain"]
    log, forks = _run_forked(pkg, ch["seq"], ["--stage-1", "--no-flush", "--fork=1,1,1"])
    assert len(forks) == 3
    assert [list(f["best"]) for f in forks] == ch["band_bests"]
    assert "Wrapping gpu ID" in log
    # --fork without weights: one instance per GPU, weighted by compute units x clock.  The weights are asked from
    # a throw-away child process (a HIP runtime initialised before MASA-Core forks would be unusable afterwards).
    log, forks = _run_forked(pkg, ch["seq"], ["--stage-1", "--no-flush", "--fork"])
    assert "fork[0+]: 100.00%" in log and len(forks) == 1
    assert list(forks[0]["best"]) == ch["single_best"]


@pytest.mark.parametrize("name", ["full_pipeline_20000x9000_b8192", "full_pipeline_pruned_60000x50000_b8192"])
def test_full_pipeline_with_stage4_on_the_gpu(pkg, oracle, name):
    """MASA-Core's stages 1-3 and 5-6 with the engine as aligner AND the product's stage 4 (mi355sw_stage4 through
    Mi355Aligner::refineCrosspoints) in place of MASA-Core's CPU stage 4: crosspoint_04.00 and alignment.00.txt are
    the reference's, byte for byte."""
    case = [c for c in G["cases"] if c["name"] == name][0]
    disk = [a for a in case["args"] if a.startswith("--disk-size")]
    out = _run(pkg, oracle, case["seq"], disk + ["--strip-rows=1024", "--gpu-stage4"])
    assert list(out["best"]) == case["best"]
    assert "GPU STAGE 4" in out["statistics"]["statistics_04.00"]
    assert hashlib.sha256(out["crosspoints_4_txt"]).hexdigest() == case["crosspoints_4"]["file_sha256"]
    assert hashlib.sha256(out["alignment_txt"]).hexdigest() == case["alignment_txt_sha256"]


def test_dump_blocks_file_equals_the_references(pkg, oracle):
    """--dump-blocks through MASA-Core: AlignerManager::dispatchScore(score, bx, by) -> BlocksFile (AlignerManager.cpp:
    418-423).  With --strip-rows=512 --block-columns=700 the engine's grid is the grid of the reference's block aligner
    run with --block=512,700: the two pruning_dump.txt files must be the same bytes (the expected file is rebuilt from
    the oracle's block table, which tests/test_oracle_vs_reference.py pins on the reference's own file; where the
    reference driver was prebuilt it is run as well)."""
    import struct
    m, n, bh, bw = 6000, 5000, 512, 700
    seq = {"kind": "related", "m": m, "n": n, "cfg": 36}
    s0, s1 = make_pair(pkg, seq)
    out = _run(pkg, oracle, seq, ["--stage-1", "--no-flush", "--no-block-pruning", "--dump-blocks", "--strip-rows=%d" % bh, "--block-columns=%d" % bw])
    o = oracle.stage1(s0, s1, block_h=bh, block_w=bw)
    gh, gw = o["grid"]
    want = struct.pack("<ii", gh, gw) + b"".join(struct.pack("<i", o["block_scores"][(bx, by)][2]) for by in range(gh) for bx in range(gw))
    assert tuple(out["best"]) == tuple(o["best"])
    assert out["blocks_file"] == want
    if oracle.have_ref():
        ref = oracle.run_ref(s0, s1, ["--stage-1", "--no-flush", "--no-block-pruning", "--dump-blocks", "--block=%d,%d" % (bh, bw)])
        assert ref["blocks_file"] == out["blocks_file"]

"""Bodies of tests/test_gpu_zz_native_pipeline.py, runnable on their own:

    python tests/native_pipeline_cases.py <case>        -> one JSON line, exit code 0 = every check held

Each case runs the native pipeline (masa-cudalign_amd/pipeline.py) on the ENGINE (cuda:0) for one full-pipeline fixture
and checks best score, stage-2 crosspoints and alignment.00.txt against what MASA-Core wrote.  The pytest file runs
them in a child process, so that a fault in this not-yet-hardware-proven combination cannot take the test session
down with it."""
import hashlib
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

CASES = {
    # name: (fixture, MI355Aligner kwargs, pipeline kwargs, exact crosspoints + text?)
    "b8192_3000x2700": ("full_pipeline_3000x2700_b8192", {}, dict(sra_limit=200 * 1024), True),
    "b8192_20000x9000": ("full_pipeline_20000x9000_b8192", {}, dict(sra_limit=200 * 1024), True),
    "other_geometry": ("full_pipeline_3000x2700", {}, dict(sra_limit=200 * 1024), False),
    # 1024-row strips like the drop-in test's --strip-rows=1024 (rows_per_lane = rows / 64)
    "pruned_60000x50000": ("full_pipeline_pruned_60000x50000_b8192", dict(rows_per_lane=16),
                           dict(sra_limit=4 * 1024 * 1024, block_pruning=True), True),
    # GLOBAL alignment (--alignment-edges=++), stage 1 with block pruning -- the reference run of the fixture did not prune
    # (its stage 1 never does for global alignments); special rows off the optimal paths are lower bounds here
    "global_pruned_60000x50000": ("full_pipeline_global_60000x50000_b8192", dict(rows_per_lane=16),
                                  dict(sra_limit=4 * 1024 * 1024, block_pruning=True, prune_global=True, alignment_start=4, alignment_end=4), True),
    "global_unpruned_60000x50000": ("full_pipeline_global_60000x50000_b8192", dict(rows_per_lane=16),
                                    dict(sra_limit=4 * 1024 * 1024, block_pruning=False, alignment_start=4, alignment_end=4), True),
}


def run(case_name):
    import __graft_entry__ as graft
    from helpers import load_golden, make_pair
    pkg = graft.load_package()
    from masa_cudalign_amd import fasta, pipeline
    from masa_cudalign_amd.crosspoints import CrosspointsFile, crosspoint_file
    fixture, akw, pkw, exact = CASES[case_name]
    case = [c for c in load_golden()["cases"] if c["name"] == fixture][0]
    s0, s1 = make_pair(pkg, case["seq"])
    q0, q1 = fasta.parse(b">s0\n" + s0.tobytes() + b"\n"), fasta.parse(b">s1\n" + s1.tobytes() + b"\n")
    work = tempfile.mkdtemp(prefix="mi355_native_")
    al = pkg.MI355Aligner(device=0, **akw)
    try:
        out = pipeline.align(al, q0, q1, work, **pkw)
    finally:
        al.close()
    cp2 = CrosspointsFile(crosspoint_file(work, 2)).load().tuples()
    want2 = [tuple(p) for p in case["crosspoints_2"]]
    checks = {"best": list(out["best"]) == case["best"],
              "alignment_score": out["alignment"] is not None and out["alignment"].raw_score == case["best"][2]}
    if exact:
        checks["crosspoints_2"] = cp2 == want2
        checks["alignment_txt"] = hashlib.sha256(out["text"]).hexdigest() == case["alignment_txt_sha256"]
        from masa_cudalign_amd import alignment_file as af
        checks["alignment_bin"] = af.canonical(af.loads(open(os.path.join(work, "alignment.00.bin"), "rb").read())) == \
            af.canonical(af.loads(bytes.fromhex(case["alignment_bin_hex"])))
    else:                      # another special-row spacing may pick another, equally optimal path
        checks["start_and_end"] = bool(cp2) and cp2[0] == want2[0] and cp2[-1] == want2[-1]
    if pkw.get("block_pruning"):
        checks["pruned"] = out["stage1"]["pruned_cells"] > 0.15 * case["m"] * case["n"]
    if "crosspoints_4" in case and exact:
        checks["crosspoints_4"] = hashlib.sha256(open(crosspoint_file(work, 4), "rb").read()).hexdigest() == case["crosspoints_4"]["file_sha256"]
    res = {"case": case_name, "checks": checks, "ok": all(checks.values()), "best": list(out["best"]),
           "crosspoints": out["crosspoints"], "seconds": out["seconds"], "stage3_rounds": out["stage3"]["rounds"],
           "pruned_fraction": out["stage1"]["pruned_cells"] / float(case["m"]) / case["n"]}
    print(json.dumps(res), flush=True)
    return 0 if res["ok"] else 1


def run_at_size(m, n):
    """A size no fixture covers (no CPU oracle can produce one): the NATIVE stages 1-6 against the DROP-IN run of the
    same pair -- MASA-Core's own stages 2-6 compiled from the reference, on the same engine (oracle/_ref/masa_mi355) --
    with block pruning and special rows on disk in both.  crosspoint_02/03/04 and alignment.00.txt must be the same
    bytes, and the alignment must re-score to the stage-1 best."""
    import shutil
    import subprocess
    import time
    import __graft_entry__ as graft
    pkg = graft.load_package()
    from masa_cudalign_amd import fasta, pipeline
    from oracle.binding import _write_fasta, read_ref_work
    binary = os.path.join(graft.ROOT, "oracle", "_ref", "masa_mi355")
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=61)
    tmp = tempfile.mkdtemp(prefix="mi355_native_size_")
    try:
        disk = "2G"
        f0, f1 = os.path.join(tmp, "s0.fasta"), os.path.join(tmp, "s1.fasta")
        _write_fasta(f0, s0, "s0")
        _write_fasta(f1, s1, "s1")
        t0 = time.time()
        p = subprocess.run([binary, "--work-dir=" + os.path.join(tmp, "dropin"), "--disk-size=" + disk, "--strip-rows=1024", "--gpu-stage4", f0, f1],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, cwd=tmp)
        t_dropin = time.time() - t0
        if p.returncode != 0:
            print(p.stdout.decode(errors="replace")[-3000:])
            return 1
        ref = read_ref_work(os.path.join(tmp, "dropin"))
        q0, q1 = fasta.parse(open(f0, "rb").read()), fasta.parse(open(f1, "rb").read())
        work = os.path.join(tmp, "native")
        al = pkg.MI355Aligner(device=0, rows_per_lane=16)
        try:
            t0 = time.time()
            out = pipeline.align(al, q0, q1, work, sra_limit=2 << 30, block_pruning=True)
            t_native = time.time() - t0
        finally:
            al.close()
        rd = lambda st: open(os.path.join(work, "crosspoints", "crosspoint_%02d.00" % st), "rb").read()
        checks = {"best": tuple(out["best"]) == tuple(ref["best"]),
                  "alignment_score": out["alignment"].raw_score == out["best"][2],
                  "pruned": out["stage1"]["pruned_cells"] > 0.2 * m * n,
                  "special_rows": len(out["stage1"]["special_rows"]) >= 16}
        for st in (2, 3, 4):
            checks["crosspoints_%d" % st] = rd(st) == ref["crosspoints_%d_txt" % st]
        checks["alignment_txt"] = hashlib.sha256(out["text"]).hexdigest() == hashlib.sha256(ref["alignment_txt"]).hexdigest()
        res = {"case": "at_size_%dx%d" % (m, n), "checks": checks, "ok": all(checks.values()), "best": list(out["best"]),
               "crosspoints": out["crosspoints"], "seconds": out["seconds"], "native_seconds": t_native, "dropin_seconds": t_dropin,
               "alignment_txt_sha256": hashlib.sha256(out["text"]).hexdigest(), "pruned_fraction": out["stage1"]["pruned_cells"] / float(m) / n}
        print(json.dumps(res), flush=True)
        return 0 if res["ok"] else 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    if sys.argv[1] == "at_size":
        sys.exit(run_at_size(int(sys.argv[2]), int(sys.argv[3])))
    sys.exit(run(sys.argv[1]))

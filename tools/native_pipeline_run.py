"""The whole pipeline natively on one MI355X (masa-cudalign_amd/pipeline.py: stages 1-3 drive the engine, stage 4 is
mi355sw_stage4, stages 5-6 host code), timed per stage -- the counterpart of tools/dropin_scale.py, which runs
MASA-Core's own stages on top of the engine.

    python tools/native_pipeline_run.py M N [sra_bytes] [out.json] [cfg]

Related pair from seqgen (cfg as in BASELINE.md section 2), local alignment, block pruning on.  Checks: the text
re-scores itself to the best score (stage 5 refuses anything else), and -- for sizes the oracle finishes in seconds --
the best score equals the oracle's.  `MI355SW_WORK` overrides the work directory (default: a temporary one, removed)."""
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
from masa_cudalign_amd import fasta, pipeline  # noqa: E402


def main():
    m, n = int(sys.argv[1]), int(sys.argv[2])
    # default budget: a special row every 8192 rows, but no more than 4 GiB of them on disk
    limit = int(float(sys.argv[3])) if len(sys.argv) > 3 and sys.argv[3] != "-" else min(max(200 * 1024, (m // 8192 + 1) * n * 8), 4 << 30)
    outfn = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] != "-" else None
    cfg = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=cfg)
    q0 = fasta.Sequence(">s0", s0, fasta.SequenceModifiers())
    q1 = fasta.Sequence(">s1", s1, fasta.SequenceModifiers())
    work = os.environ.get("MI355SW_WORK") or tempfile.mkdtemp(prefix="mi355_native_")
    al = pkg.MI355Aligner(device=0)
    prof_fn = os.environ.get("MI355SW_PROFILE_STAGE2")        # cProfile of stage 2 alone, as text
    if prof_fn:
        import cProfile, io, pstats
        real_stage2 = pipeline.stage2

        def profiled_stage2(*a, **kw):
            pr = cProfile.Profile()
            try:
                return pr.runcall(real_stage2, *a, **kw)
            finally:
                buf = io.StringIO()
                pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(45)
                open(prof_fn, "w").write(buf.getvalue())
        pipeline.stage2 = profiled_stage2
    t0 = time.time()
    try:
        out = pipeline.align(al, q0, q1, work, sra_limit=limit)
    finally:
        al.close()
    total = time.time() - t0
    res = {"workload": "%dx%d related pair (seqgen cfg=%d), local, stages 1-6 natively" % (m, n, cfg), "sra_bytes": limit,
           "best": list(out["best"]), "seconds": {str(k): v for k, v in out["seconds"].items()}, "total_seconds": total,
           "crosspoints": {str(k): v for k, v in out["crosspoints"].items()},
           "stage1_gcups": out["stage1"]["gcups"], "stage1_kernel_ms": out["stage1"].get("kernel_ms"),
           "stage1_strip_rows": out["stage1"].get("strip_rows"), "stage1_pruned_fraction": out["stage1"].get("pruned_cells", 0) / float(m) / n,
           "stage3_rounds": out.get("stage3", {}).get("rounds"), "stage2_speculation": out.get("stage2", {}).get("speculation"),
           "stage4": out.get("stage4"), "alignment_score": out["alignment"].raw_score if out["alignment"] else None,
           "text_bytes": len(out["text"]) if out["text"] else 0}
    # the same digests tools/dropin_scale.py records for MASA-Core's own stages on the engine: equal = the same files
    import hashlib
    res["alignment_sha256"] = hashlib.sha256(out["text"]).hexdigest() if out["text"] else None
    cp4 = os.path.join(work, "crosspoints", "crosspoint_04.00")
    res["crosspoint_04_sha256"] = hashlib.sha256(open(cp4, "rb").read()).hexdigest() if os.path.exists(cp4) else None
    if float(m) * n <= 4e9:
        oracle = g.load_oracle()
        res["oracle_best"] = list(oracle.stage1(s0, s1)["best"])
        res["ok"] = res["oracle_best"][2] == res["best"][2] == res["alignment_score"]
    else:
        res["ok"] = res["best"][2] == res["alignment_score"]
    print(json.dumps(res), flush=True)
    if outfn:
        json.dump(res, open(outfn, "w"), indent=1)
    if not os.environ.get("MI355SW_WORK"):
        shutil.rmtree(work, ignore_errors=True)
    assert res["ok"], res


if __name__ == "__main__":
    main()

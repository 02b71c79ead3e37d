"""MASA-Core (the reference's own stages 1-6, compiled from its sources into oracle/_ref/masa_mi355) with the HIP
engine as its aligner, on a pair too large for the reference's CPU aligner: per-stage times from the reference's
own statistics files, and self-consistency of the result (the alignment that stage 6 prints re-scores to the best
score stage 1 reported; stage-2 crosspoints start/end where stage 1 says the alignment ends).
  python tools/dropin_scale.py m n [disk-size] [out.json]        (DROPIN_EXTRA="--gpu-stage4 ..." adds options)
This is test infrastructure (it runs a binary from oracle/_ref); the product is the library behind it."""
import json, os, re, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

pkg = g.load_package()
g.load_oracle()
from oracle.binding import _write_fasta, read_ref_work

MATCH, MISMATCH, GAP_OPEN, GAP_EXT = 1, -3, 3, 2


def rescore(txt):
    """score of the alignment text written by stage 6 (M/stage6): pairs of sequence lines with '-' for gaps"""
    q, s = [], []
    for ln in txt.decode(errors="replace").splitlines():
        m = re.match(r"^(Query|Sbjct):\s*\d+\s+([A-Za-z\-]+)\s+\d+", ln)
        if m:
            (q if m.group(1) == "Query" else s).append(m.group(2))
    a, b = "".join(q), "".join(s)
    if not a or len(a) != len(b):
        return None, len(a), len(b)
    A, B = np.frombuffer(a.encode(), np.uint8), np.frombuffer(b.encode(), np.uint8)
    ga, gb = A == ord("-"), B == ord("-")
    gap = ga | gb
    score = int(((A == B) & ~gap).sum()) * MATCH + int(((A != B) & ~gap).sum()) * MISMATCH
    # gap runs: every gap column costs EXT, every run additionally OPEN
    for gm in (ga, gb):
        runs = int((gm[1:] & ~gm[:-1]).sum()) + int(gm[0])
        score -= GAP_EXT * int(gm.sum()) + GAP_OPEN * runs
    return score, len(a), int(gap.sum())


def main():
    m, n = int(sys.argv[1]), int(sys.argv[2])
    disk = sys.argv[3] if len(sys.argv) > 3 else "2G"
    outfn = sys.argv[4] if len(sys.argv) > 4 else None
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=5)
    tmp = tempfile.mkdtemp(prefix="masa_dropin_scale_", dir=os.environ.get("TMPDIR", "/tmp"))
    res = {"m": m, "n": n, "disk_size": disk}
    try:
        f0, f1 = os.path.join(tmp, "s0.fasta"), os.path.join(tmp, "s1.fasta")
        _write_fasta(f0, s0, "s0")
        _write_fasta(f1, s1, "s1")
        work = os.path.join(tmp, "work")
        t0 = time.time()
        p = subprocess.run([os.path.join(g.ROOT, "oracle", "_ref", "masa_mi355"), "--work-dir=" + work,
                            "--disk-size=" + disk] + os.environ.get("DROPIN_EXTRA", "").split() + [f0, f1],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=tmp)
        res["extra_options"] = os.environ.get("DROPIN_EXTRA", "")
        res["wall_s"] = time.time() - t0
        res["returncode"] = p.returncode
        log = p.stdout.decode(errors="replace")
        res["log_tail"] = log[-1500:]
        if os.environ.get("DROPIN_LOG"):          # the whole log (e.g. with --engine-verbosity=2: one line per mi355sw_align_partition job)
            open(os.environ["DROPIN_LOG"], "w").write(log)
        # special rows stay on disk (gigabytes at these sizes): count the files, read everything else
        sra = os.path.join(work, "special_rows")
        n_rows = sum(len([f for f in fs if len(f) == 8]) for _, _, fs in os.walk(sra)) if os.path.isdir(sra) else 0
        sra_bytes = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(sra) for f in fs) if os.path.isdir(sra) else 0
        shutil.rmtree(sra, ignore_errors=True)
        if os.environ.get("DROPIN_KEEP_CROSSPOINTS"):     # crosspoint_02 / _03 of this run, for a line-by-line comparison of two runs
            os.makedirs(os.environ["DROPIN_KEEP_CROSSPOINTS"], exist_ok=True)
            for st in (2, 3):
                fn = os.path.join(work, "crosspoints", "crosspoint_%02d.00" % st)
                if os.path.exists(fn):
                    shutil.copy(fn, os.path.join(os.environ["DROPIN_KEEP_CROSSPOINTS"], "crosspoint_%02d.00" % st))
        out = read_ref_work(work, log=log)
        res["best"] = list(out["best"]) if out["best"] else None
        for st in range(1, 7):
            fn = os.path.join(work, "statistics_%02d.00" % st)
            if os.path.exists(fn):
                txt = open(fn).read()
                tm = re.findall(r"^\s*(\w[\w ]*):\s+([\d.]+)", txt, re.M)
                res["stage%d" % st] = {k.strip(): float(v) for k, v in tm[:30]}
        for st in (2, 3, 4):
            pts = out.get("crosspoints_%d" % st)
            if pts:
                res["crosspoints_%d" % st] = {"count": len(pts), "first": list(pts[0]), "last": list(pts[-1])}
        import hashlib
        for st in (2, 3):
            if out.get("crosspoints_%d" % st):
                res["crosspoint_%02d_sha256" % st] = hashlib.sha256(repr([tuple(p) for p in out["crosspoints_%d" % st]]).encode()).hexdigest()
        if "crosspoints_4_txt" in out:
            res["crosspoint_04_sha256"] = hashlib.sha256(out["crosspoints_4_txt"]).hexdigest()
        gl = [ln for ln in out.get("statistics", {}).get("statistics_04.00", "").splitlines() if ln.startswith("GPU STAGE 4")]
        if gl:
            res["gpu_stage4"] = gl[0]
        if "alignment_txt" in out:
            res["alignment_sha256"] = hashlib.sha256(out["alignment_txt"]).hexdigest()
            sc, length, gaps = rescore(out["alignment_txt"])
            res["alignment"] = {"bytes": len(out["alignment_txt"]), "columns": length, "gap_columns": gaps, "rescored": sc}
            res["rescore_equals_best"] = (sc == res["best"][2]) if (sc is not None and res["best"]) else None
        res["special_row_files"], res["special_row_bytes"] = n_rows, sra_bytes
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(res, indent=1))
    if outfn:
        json.dump(res, open(outfn, "w"), indent=1)


if __name__ == "__main__":
    main()

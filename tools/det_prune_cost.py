"""What the reproducible pruning mode costs on a SEEDED run: a related M x N pair (default 16 M x 14.65 M, the size VERDICT r5 asked about),
local SW, block pruning behind the seed, without / with MI355SW_F_DETERMINISTIC_PRUNE (twice: the two runs' special rows, last row and last
column must be the same bytes).    python tools/det_prune_cost.py [M N]"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import __graft_entry__ as g  # noqa: E402


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 16000000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 14650000
    pkg = g.load_package()
    from test_gpu_bound import _stream
    from masa_cudalign_amd.engine import SMITH_WATERMAN, F_DETERMINISTIC_PRUNE
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=5)
    out = {"workload": "%dx%d related pair (seqgen cfg=5), local SW, block pruning behind the seed, special rows every 2 Mi rows" % (m, n), "runs": []}
    for name, flags in (("running best", 0), ("reproducible", F_DETERMINISTIC_PRUNE), ("reproducible again", F_DETERMINISTIC_PRUNE)):
        al = pkg.MI355Aligner(device=0, flags=flags, max_special_bytes=32 << 30)
        try:
            al.setSequences(s0, s1)
            t0 = time.time()
            r = _stream(pkg, al, m, n, SMITH_WATERMAN, None, interval=2 << 20)
            dt = time.time() - t0
        finally:
            al.close()
        h = hashlib.sha256()
        for dp in sorted(r["rows"]):
            h.update(np.ascontiguousarray(r["rows"][dp], dtype=np.int32).tobytes())
        h.update(np.ascontiguousarray(r["last_row"], dtype=np.int32).tobytes())
        h.update(np.ascontiguousarray(r["last_col"], dtype=np.int32).tobytes())
        st = r["stats"]
        rec = {"mode": name, "seconds": dt, "kernel_ms": st["kernel_ms"], "seed_ms": st["seed_ms"], "pruned_fraction": st["pruned_cells"] / float(m) / n,
               "best": list(r["best"]), "rows": len(r["rows"]), "sha256_rows_last_row_last_column": h.hexdigest(), "kernel": st["kernel"]}
        out["runs"].append(rec)
        print(json.dumps(rec), flush=True)
    a, b, c = out["runs"]
    out["check"] = {"same_best": a["best"] == b["best"] == c["best"], "two_reproducible_runs_same_bytes": b["sha256_rows_last_row_last_column"] == c["sha256_rows_last_row_last_column"],
                    "cost": b["kernel_ms"] / a["kernel_ms"] - 1.0}
    out["library_build_id"] = pkg.engine.library_build_id()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

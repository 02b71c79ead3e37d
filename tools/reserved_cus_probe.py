"""What leaving K compute units without a strip wavefront costs (bench.py --reserve-cus / MI355SW_RESERVE_CUS): a tall
unrelated pair (24 M x 3 M, the per-GPU share of the N > 1 bench) with 4 * (256 - K) persistent wavefronts.  One JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
from masa_cudalign_amd.bands import BandRunner  # noqa: E402

m, n = int(sys.argv[1]) if len(sys.argv) > 1 else 24000000, int(sys.argv[2]) if len(sys.argv) > 2 else 3000000
s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=2)
out = {"workload": "%dx%d unrelated random ACGT, local SW, score-only" % (m, n), "runs": []}
for k in (0, 2, 4, 8, 16):
    al = pkg.MI355Aligner(device=0, waves=4 * (256 - k) if k else 0)
    try:
        al.setSequences(s0, s1)
        BandRunner(al).run(min(m, 3000000), 0, n)
        t0 = time.time()
        best = BandRunner(al).run(m, 0, n)
        dt = time.time() - t0
        st = al.getStatistics()
        out["runs"].append({"reserved_cus": k, "waves": st["waves"], "strip_rows": st["strip_rows"], "kernel": st["kernel"], "kernel_ms": st["kernel_ms"],
                            "gcups": float(m) * n / st["kernel_ms"] / 1e6, "best": list(best)})
    finally:
        al.close()
base = out["runs"][0]["gcups"]
for r in out["runs"]:
    r["relative"] = r["gcups"] / base
print(json.dumps(out))

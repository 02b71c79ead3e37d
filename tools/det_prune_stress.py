"""Reproducible pruning under load: the same pruning run (MI355SW_F_DETERMINISTIC_PRUNE) REPS times while NOISE other processes keep
the GPU busy with small alignments; reports which of special rows / last row / last column differ between repetitions, and where.
    python tools/det_prune_stress.py [m n reps noise]"""
import hashlib
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import __graft_entry__ as g  # noqa: E402

NOISE = """
import sys, time
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package()
s0, s1 = pkg.seqgen.related_pair(600000, 500000, cfg=3)
al = pkg.MI355Aligner(device=0)
al.setSequences(s0, s1)
t0 = time.time()
while time.time() - t0 < %d:
    part = pkg.Partition(0, 0, 600000, 500000)
    mg = pkg.Stage1Manager(part, block_pruning=True)
    al.alignPartition(part, mg)
"""


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 2600000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2300000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    noise = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    pkg = g.load_package()
    from test_gpu_bound import _stream
    from masa_cudalign_amd.engine import SMITH_WATERMAN, F_DETERMINISTIC_PRUNE
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, "-c", NOISE % (root, 90)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for _ in range(noise)]
    time.sleep(8.0)
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=611)
    runs = []
    for k in range(reps):
        al = pkg.MI355Aligner(device=0, flags=F_DETERMINISTIC_PRUNE)
        try:
            al.setSequences(s0, s1)
            t0 = time.time()
            r = _stream(pkg, al, m, n, SMITH_WATERMAN, None, interval=max(8192, m // 24))
            print("run %d: %.2f s, kernel %.0f ms, pruned %.4f, best %s" % (k, time.time() - t0, r["stats"]["kernel_ms"], r["stats"]["pruned_cells"] / float(m) / n, r["best"]), flush=True)
            runs.append(r)
        finally:
            al.close()
    for p in procs:
        p.terminate()
    a = runs[0]
    for k, b in enumerate(runs[1:], 1):
        for dp in sorted(a["rows"]):
            d = np.flatnonzero(np.any(a["rows"][dp] != b["rows"][dp], axis=1))
            if len(d):
                print("run %d: special row %d differs in %d cells, first at column %d: %s vs %s" % (k, dp, len(d), d[0], a["rows"][dp][d[0]], b["rows"][dp][d[0]]))
        for key in ("last_row", "last_col"):
            d = np.flatnonzero(np.any(a[key] != b[key], axis=1))
            if len(d):
                print("run %d: %s differs in %d cells, first at %d: %s vs %s; last at %d" % (k, key, len(d), d[0], a[key][d[0]], b[key][d[0]], d[-1]))
    print("done")


if __name__ == "__main__":
    main()

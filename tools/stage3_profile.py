"""Where a native stage-3 sweep spends its time: stages 1-2 as usual, stage 3 under cProfile.
    python tools/stage3_profile.py M N"""
import cProfile, pstats, io, os, sys, tempfile, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
from masa_cudalign_amd import fasta
from masa_cudalign_amd.stage1 import stage1
from masa_cudalign_amd.stage2 import stage2
from masa_cudalign_amd.stage3 import stage3
import numpy as np
m, n = int(sys.argv[1]), int(sys.argv[2])
s0, s1 = pkg.seqgen.related_pair(m, n, cfg=2)
work = tempfile.mkdtemp(prefix="s3prof_")
limit = min(max(200 * 1024, (m // 8192 + 1) * n * 8), 4 << 30)
al = pkg.MI355Aligner(device=0)
areas = {}
try:
    t = time.time(); r1 = stage1(al, s0, s1, work, sra_limit=limit, areas=areas); print("stage1 %.2f s best %s" % (time.time() - t, r1["best"]))
    acc2 = {"kernel_ms": 0.0, "calls": 0, "rows": 0, "cols": 0, "total_ms": 0.0}
    orig2 = al.alignPartition
    def counted2(part, mgr):
        orig2(part, mgr)
        st = al.getStatistics()
        acc2["kernel_ms"] += st["kernel_ms"]; acc2["total_ms"] += st["total_ms"]; acc2["calls"] += 1
        acc2["rows"] += part.i1 - part.i0; acc2["cols"] += part.j1 - part.j0
    al.alignPartition = counted2
    pr2 = cProfile.Profile()
    t = time.time(); pr2.enable(); r2 = stage2(al, s0, s1, work, sra_limit=limit, areas=areas); pr2.disable()
    print("stage2 %.2f s, %d crosspoints" % (time.time() - t, len(r2["crosspoints"])))
    print("stage2 engine: %d calls, kernel %.1f ms (%.2f per call), wall in the engine %.1f ms, mean partition %d x %d" % (
        acc2["calls"], acc2["kernel_ms"], acc2["kernel_ms"] / max(1, acc2["calls"]), acc2["total_ms"], acc2["rows"] // max(1, acc2["calls"]), acc2["cols"] // max(1, acc2["calls"])))
    out2 = io.StringIO(); pstats.Stats(pr2, stream=out2).sort_stats("tottime").print_stats(12); print(out2.getvalue()[:2600])
    al.alignPartition = orig2
    acc = {"kernel_ms": 0.0, "calls": 0, "cells": 0, "rows": 0, "cols": 0, "launches": 0}
    orig = al.alignPartition
    def counted(part, mgr):
        orig(part, mgr)
        st = al.getStatistics()
        acc["kernel_ms"] += st["kernel_ms"]; acc["calls"] += 1; acc["cells"] += st["cells"]
        acc["rows"] += part.i1 - part.i0; acc["cols"] += part.j1 - part.j0; acc["launches"] += st["kernel_launches"]
    al.alignPartition = counted
    pr = cProfile.Profile()
    t = time.time()
    pr.enable(); r3 = stage3(al, s0, s1, work, sra_limit=limit, areas=areas); pr.disable()
    print("stage3 %.2f s, rounds %s" % (time.time() - t, r3["rounds"]))
    print("engine: %d calls, kernel %.1f ms total (%.2f ms per call), mean partition %d x %d, %.1f Gcells" % (
        acc["calls"], acc["kernel_ms"], acc["kernel_ms"] / max(1, acc["calls"]), acc["rows"] // max(1, acc["calls"]), acc["cols"] // max(1, acc["calls"]), acc["cells"] / 1e9))
    out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(8); print(out.getvalue()[:2500])
finally:
    al.close(); shutil.rmtree(work, ignore_errors=True)

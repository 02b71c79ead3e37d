"""Stage-3-shaped calls (M/stage3/sw_stage3.cpp): many small NW partitions (about 10k x 10k) with custom borders,
last column / last row dispatched, one alignPartition per partition.  python tools/stage3_probe.py m n reps"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
from masa_cudalign_amd.manager import Stage1Manager, ArrayCellsReader, AT_SEQUENCE_1_AND_2
m, n, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
S0, S1 = pkg.seqgen.related_pair(m * 4, n * 4, cfg=9)
INF = pkg.engine.INF
col = np.zeros((m + 1, 2), dtype=np.int32); col[:, 0] = -2 * np.arange(m + 1) - 3; col[0, 0] = 0; col[:, 1] = -INF
row = np.zeros((n + 1, 2), dtype=np.int32); row[:, 0] = -2 * np.arange(n + 1) - 3; row[0, 0] = 0; row[:, 1] = -INF
al = pkg.MI355Aligner(device=0)
al.setSequences(S0, S1)
ts = []
for rep in range(reps):
    i0, j0 = (rep % 3) * m, (rep % 3) * n
    part = pkg.Partition(i0, j0, i0 + m, j0 + n)
    mg = Stage1Manager(part, alignment_start=AT_SEQUENCE_1_AND_2, alignment_end=AT_SEQUENCE_1_AND_2, keep_last_column=True,
                       keep_last_row=True, first_row_reader=ArrayCellsReader(row), first_column_reader=ArrayCellsReader(col))
    t0 = time.time()
    al.alignPartition(part, mg)
    ts.append(time.time() - t0)
    st = al.getStatistics()
print("m=%d n=%d: wall per call ms: first %.2f, median %.2f, min %.2f; kernel %.2f ms strips=%d strip_rows=%d" % (
    m, n, ts[0] * 1e3, sorted(ts)[len(ts) // 2] * 1e3, min(ts) * 1e3, st["kernel_ms"], st["strips"], st["strip_rows"]))
al.close()

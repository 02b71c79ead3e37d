"""Stage-2-shaped call (M/stage2/sw_stage2.cpp:387-441): a TALL, NARROW NW partition whose first column and first
row are custom data, whose last column is dispatched progressively and whose manager says stop (mustContinue = 0)
once the goal was seen in the first rows.  python tools/stage2_probe.py m n stop_after_rows [special row interval] [rows per lane]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
from masa_cudalign_amd.manager import Stage1Manager, ArrayCellsReader, AT_SEQUENCE_1_AND_2

m, n, stop_rows = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
interval = int(sys.argv[4]) if len(sys.argv) > 4 else 0
s0, s1 = pkg.seqgen.related_pair(m, n, cfg=9)
INF = pkg.engine.INF
col = np.zeros((m + 1, 2), dtype=np.int32); col[:, 0] = -2 * np.arange(m + 1) - 3; col[0, 0] = 0; col[:, 1] = -INF
row = np.zeros((n + 1, 2), dtype=np.int32); row[:, 0] = -2 * np.arange(n + 1) - 3; row[0, 0] = 0; row[:, 1] = -INF

class Mgr(Stage1Manager):
    nrows = 0
    def dispatchRow(self, i, buf, length):
        self.nrows += 1
    def dispatchColumn(self, j, buf, length):
        self.last_column_pos += length
        if self.last_column_pos >= stop_rows:
            self.active = False

al = pkg.MI355Aligner(device=0, rows_per_lane=int(sys.argv[5]) if len(sys.argv) > 5 else 0)
al.setSequences(s0, s1)
part = pkg.Partition(0, 0, m, n)
for rep in range(3):
    mg = Mgr(part, alignment_start=AT_SEQUENCE_1_AND_2, alignment_end=AT_SEQUENCE_1_AND_2, keep_last_column=True, special_row_interval=interval,
             first_row_reader=ArrayCellsReader(row), first_column_reader=ArrayCellsReader(col))
    t0 = time.time()
    al.alignPartition(part, mg)
    dt = time.time() - t0
    st = al.getStatistics()
    print("m=%d n=%d stop_after=%d: wall %.1f ms kernel %.1f ms strips=%d strip_rows=%d waves=%d last-col rows seen %d, row dispatches %d" % (
        m, n, stop_rows, dt * 1e3, st["kernel_ms"], st["strips"], st["strip_rows"], st["waves"], mg.last_column_pos, mg.nrows), flush=True)
al.close()

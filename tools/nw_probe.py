"""NW on the packed kernel vs oracle at small sizes: python tools/nw_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); oracle = g.load_oracle()
for (m, n) in [(300, 500), (3000, 2000), (40000, 30000)]:
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=5)
    al = pkg.MI355Aligner(device=0)
    al.setSequences(s0, s1)
    part = pkg.Partition(0, 0, m, n)
    try:
        al.streamBegin(part, recurrence_type=pkg.NEEDLEMAN_WUNSCH, first_row_init_type=pkg.INIT_WITH_GAPS,
                       first_column_init_type=pkg.INIT_WITH_GAPS, track_best=False, want_last_row=True)
        while True:
            rows, fin = al.streamPoll()
            if fin: break
            time.sleep(0.002)
        last = al.streamReadLastRow(0, n)
        best, _ = al.streamEnd()
        st = al.getStatistics()
        ref = oracle.stage1(s0, s1, recurrence=0, first_row_type=1, first_col_type=1, want_last_row=True)
        print(m, n, "kernel", st["profile_kernel"], "R", st["strip_rows"], "last cell", last[-1].tolist(), "oracle", ref["last_row"][-1].tolist(),
              "row equal", np.array_equal(last, ref["last_row"][1:]))
    except Exception as e:
        print(m, n, "ERR", e)
    al.close()

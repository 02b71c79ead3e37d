"""first-segment shape of the 'ties' pair: local SW over (0,0)-(36864,12864), values only, pruning on: H of the last cell"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
oracle = g.load_oracle()
from masa_cudalign_amd.engine import F_NO_WINDOW, SMITH_WATERMAN
from test_gpu_bound import _pairs
s0, s1 = _pairs(pkg, "ties")
M, N = 36864, 12864
ref = oracle.stage1(s0[:M], s1[:N], want_last_row=True, threads=0)
print("oracle best", ref["best"], "H last cell", ref["last_row"][-1], flush=True)
part = pkg.Partition(0, 0, M, N)
for R in (8, 4):
    for flags in (0, F_NO_WINDOW):
        for track in (False, True):
            for prune in (True, False):
                al = pkg.MI355Aligner(device=0, rows_per_lane=R, flags=flags)
                al.setSequences(s0, s1)
                al.streamBegin(part, track_best=track, prune_blocks=prune, want_last_row=True)
                while True:
                    rows, fin = al.streamPoll()
                    if fin:
                        break
                    time.sleep(0.001)
                lr = al.streamReadLastRow()
                best, _ = al.streamEnd()
                st = al.getStatistics()
                bad = int((lr[:, 0] > ref["last_row"][1:, 0]).sum())
                print("R", R, "flags", flags, "track", track, "prune", prune, "best", best, "H last", lr[-1], "max last row", int(lr[:, 0].max()), "cells above oracle", bad,
                      "pruned %.2f" % (st["pruned_cells"] / st["cells"]), st["kernel"], flush=True)
                al.close()

"""What an UNRELATED pair pays for block pruning "left on" (MASA-Core's default): 3 M x 3 M through mi355sw_align_partition, wall time
of the call without and with the request (the probe -- left edge of every row block, then stripes across the width -- included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
m = n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000000
s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=2)
al = pkg.MI355Aligner(device=0)
al.setSequences(s0, s1)
part = pkg.Partition(0, 0, m, n)
for rep in range(3):
    for prune in (False, True):
        mg = pkg.Stage1Manager(part, block_pruning=prune)
        t0 = time.time()
        al.alignPartition(part, mg)
        dt = time.time() - t0
        st = al.getStatistics()
        print("pruning requested %-5s wall %.1f ms kernel %.1f ms  %s  best %s  skipped %.3f" % (prune, dt * 1e3, st["kernel_ms"], st["kernel"], tuple(mg.getBestScore()), st["pruned_cells"] / st["cells"]), flush=True)
al.close()

"""Block pruning on/off on a related pair: python tools/prune_probe.py m n [rows per lane [passes, e.g. 1 = one pruned pass, 01 = unpruned then pruned]]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
m, n = int(sys.argv[1]), int(sys.argv[2])
R = int(sys.argv[3]) if len(sys.argv) > 3 else 0
s0, s1 = (pkg.seqgen.unrelated_pair if os.environ.get('PROBE_UNRELATED') else pkg.seqgen.related_pair)(m, n, cfg=5)
al = pkg.MI355Aligner(device=0, rows_per_lane=R)
al.setSequences(s0, s1)
part = pkg.Partition(0, 0, m, n)
for prune in ([c == '1' for c in sys.argv[4]] if len(sys.argv) > 4 else (False, True, True)):
    al.streamBegin(part, prune_blocks=prune)
    while True:
        rows, fin = al.streamPoll()
        if fin: break
        time.sleep(0.002)
    best, _ = al.streamEnd()
    st = al.getStatistics()
    print("prune=%d k=%d R=%d kernel_ms=%.1f GCUPS(m*n)=%.1f pruned=%.1f%% best=%s" % (
        prune, st["profile_kernel"], st["strip_rows"] // 64, st["kernel_ms"], m * n / st["kernel_ms"] / 1e6,
        100.0 * st["pruned_cells"] / st["cells"], best), flush=True)
al.close()

import sys, json, time
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
pkg = g.load_package()
out = []
for (m, n, rel) in ((3000000, 3000000, False), (4000000, 3000000, True), (1000000, 1000000, False), (1000000, 1000000, True)):
    s0, s1 = (pkg.seqgen.related_pair if rel else pkg.seqgen.unrelated_pair)(m, n, cfg=2)
    al = pkg.MI355Aligner(device=0)
    al.setSequences(s0, s1)
    part = pkg.Partition(0, 0, m, n)
    for prune in (False, True):
        for rep in range(2):
            mg = pkg.Stage1Manager(part, block_pruning=prune)
            al.alignPartition(part, mg)
            st = al.getStatistics()
        out.append(dict(m=m, n=n, related=rel, prune=prune, kernel=st["kernel"], strip_rows=st["strip_rows"], kernel_ms=round(st["kernel_ms"], 1),
                        total_ms=round(st["total_ms"], 1), gcups=round(m * n / st["kernel_ms"] / 1e6), pruned=round(st["pruned_cells"] / m / n, 3), best=list(mg.getBestScore())))
        print(json.dumps(out[-1]), flush=True)
    al.close()

"""debug: why does a band fed through a port not prune?  two bands in one process, sequential."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
import numpy as np
m, n, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
s0, s1 = pkg.seqgen.related_pair(m, n, cfg=2)
h = n // 2
a0, a1 = pkg.MI355Aligner(device=0, rows_per_lane=R), pkg.MI355Aligner(device=0, rows_per_lane=R)
for al in (a0, a1):
    al.setSequences(s0, s1)
a1.portCreate(m)
a0.portAttach(a1)
def run(al, part, **kw):
    al.streamBegin(part, **kw)
    try:
        while not al.streamPoll()[1]:
            time.sleep(0.001)
        best, _ = al.streamEnd()
    except pkg.AlignerError as e:
        try:
            al.streamAbort(); al.streamEnd()
        except pkg.AlignerError:
            pass
        return "ERROR " + str(e)[:120]
    st = al.getStatistics()
    return best, st["pruned_cells"] / st["cells"], st["kernel_ms"]
print("band 0 port out :", run(a0, pkg.Partition(0, 0, m, h), prune_blocks=True, prune_rows=m, prune_cols=n, share_best=True, last_column_port=True))
col = np.concatenate([np.array([[0, -pkg.INF]], dtype=np.int32), a1.portRead(0, m)])
print("boundary column H: max %d, rows with H > 16000: %d, H > 0: %d" % (col[:, 0].max(), int((col[:, 0] > 16000).sum()), int((col[:, 0] > 0).sum())))
for r0 in range(0, m, 512):
    w = col[1 + r0:1 + r0 + 512]
    if int(w[:, 0].max()) - int(w[:, 0].min()) > 20000:
        print("rows %d..%d: H %d..%d  E max %d; first cells %s" % (r0, r0 + 512, w[:, 0].min(), w[:, 0].max(), w[:, 1].max(), w[:6].tolist()))
        jump = np.flatnonzero(np.abs(np.diff(w[:, 0].astype(np.int64))) > 1000)
        print("   jumps at", [(int(r0 + k), int(w[k, 0]), int(w[k + 1, 0])) for k in jump[:6]])
        break
for shared in (True, False):
    print("band 1 from port, share_best=%s:" % shared, run(a1, pkg.Partition(0, h, m, n), prune_blocks=True, prune_rows=m, prune_cols=n - h, share_best=shared,
          first_column_init_type=pkg.INIT_WITH_CUSTOM_DATA, first_column_port=True, first_column=col[:1]))
print("band 1 custom column (device copy):", run(a1, pkg.Partition(0, h, m, n), prune_blocks=True, prune_rows=m, prune_cols=n - h,
      first_column_init_type=pkg.INIT_WITH_CUSTOM_DATA, first_column=col))
print("band 1 custom column, no pruning   :", run(a1, pkg.Partition(0, h, m, n), first_column_init_type=pkg.INIT_WITH_CUSTOM_DATA, first_column=col))
print("band 1 columns alone, zero column  :", run(a1, pkg.Partition(0, h, m, n), prune_blocks=True, prune_rows=m, prune_cols=n - h))
print("whole matrix                       :", run(a0, pkg.Partition(0, 0, m, n), prune_blocks=True))

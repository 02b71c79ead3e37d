import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); oracle = g.load_oracle()
from helpers import load_golden, make_pair, digest
G = load_golden(); ch = G["chain"]
s0, s1 = make_pair(pkg, ch["seq"])
from masa_cudalign_amd.bands import band_limits
lim = band_limits(len(s1), [1]*3)
dummy = pkg.MI355Aligner(device=0)
if len(sys.argv) > 1:
    dummy.setSequences(s0, s1); r_ = np.zeros((100,2),np.int32); c_ = np.zeros((201,2),np.int32); dummy.processBlock(r_, c_, 0, 0, 200, 100, 1); dummy.unsetSequences()
al = pkg.MI355Aligner(device=0)
al.setSequences(s0, s1)
col = None
for k in range(3):
    part = pkg.Partition(0, lim[k], len(s0), lim[k+1])
    kw = dict(want_last_column=True)
    if col is not None:
        kw.update(first_column_init_type=pkg.INIT_WITH_CUSTOM_DATA, stream_first_column=True, first_column=col[:1])
    al.streamBegin(part, **kw)
    fed = 0; t0 = time.time(); it = 0
    while True:
        if col is not None and fed < len(s0):
            ln = min(777, len(s0) - fed)
            al.streamFeedColumn(fed, col[1+fed:1+fed+ln]); fed += ln
        rows, fin = al.streamPoll(); it += 1
        if fin: break
        if time.time() - t0 > 3:
            print("stuck", k, fed, rows, al.getProgressString(), flush=True); al.streamAbort(); time.sleep(0.5); print(al.getProgressString()); os._exit(1)
    newcol = np.concatenate([np.array([[0, -pkg.INF]], dtype=np.int32), al.streamReadColumn(0, len(s0))])
    best, _ = al.streamEnd()
    print(k, best, it, digest(newcol)["sha256"][:12], ch["boundary_columns"].get("STEP-%d.tmp" % (k+1), {}).get("sha256", "")[:12], flush=True)
    col = newcol

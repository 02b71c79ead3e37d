import sys, time, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("start", flush=True)
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); oracle = g.load_oracle()
print("loaded", flush=True)
al = pkg.MI355Aligner(device=0, rows_per_lane=4)
print("created", al.getCapabilities()["smith_waterman"], flush=True)
s0, s1 = pkg.seqgen.related_pair(100, 90, cfg=3)
al.setSequences(s0, s1)
print("seq set", flush=True)
part = pkg.Partition(0, 0, 100, 90)
al.streamBegin(part)
print("begun", flush=True)
t0 = time.time()
while time.time() - t0 < 8:
    rows, fin = al.streamPoll()
    if fin: break
    time.sleep(0.01)
print("poll:", rows, fin, flush=True)
if fin:
    print(al.streamEnd(), al.getStatistics(), flush=True)
    print("oracle", oracle.stage1(s0, s1)["best"])
else:
    print(al.getProgressString(), flush=True)
    al.streamAbort()
    time.sleep(1)
    print("after abort", al.streamPoll(), al.getProgressString(), flush=True)
    os._exit(3)

"""Do two persistent strip kernels of ONE process (two engine handles = two high-priority streams) run at the same
time on the GPU?  (Precondition of the mixed-strip-height plan in docs/NOTEBOOK_r1-r3.md section 8.)
python tools/concurrency_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
m, n = 256 * 1536, 1000000
s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=2)
part = pkg.Partition(0, 0, m, n)
als = []
for k in range(2):
    if len(sys.argv) > 1 and k == 1:
        os.environ["MI355SW_STREAM_PRIO"] = sys.argv[1]     # second engine's stream in another priority pool
    als.append(pkg.MI355Aligner(device=0, rows_per_lane=24, waves=256))
for al in als:
    al.setSequences(s0, s1)
def run(handles):
    t0 = time.time()
    for al in handles:
        al.streamBegin(part)
    left = list(handles)
    while left:
        left = [al for al in left if not al.streamPoll()[1]]
        time.sleep(0.001)
    res = [al.streamEnd()[0] for al in handles]
    return time.time() - t0, res, [al.getStatistics()["kernel_ms"] for al in handles]
for rep in range(2):
    print("one handle : %.1f ms %s" % (run(als[:1])[0] * 1e3, run(als[:1])[2]))
    dt, res, k = run(als)
    print("two handles: %.1f ms kernel_ms=%s same result: %s" % (dt * 1e3, k, res[0] == res[1]))
for al in als:
    al.close()

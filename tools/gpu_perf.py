"""On-box throughput probe: python tools/gpu_perf.py m,n,R,waves,flags,track,reps,prune ..."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
def perf(m, n, R=0, waves=0, flags=0, track=1, reps=2, prune=0):
    s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=2)
    al = pkg.MI355Aligner(device=0, rows_per_lane=R, waves=waves, flags=flags)
    al.setSequences(s0, s1)
    part = pkg.Partition(0, 0, m, n)
    for it in range(reps):
        t0 = time.time()
        al.streamBegin(part, track_best=bool(track), prune_blocks=bool(prune))
        while True:
            rows, fin = al.streamPoll()
            if fin: break
            time.sleep(0.002)
        best, _ = al.streamEnd()
        dt = time.time() - t0
        st = al.getStatistics()
        print("perf k=%d m=%d n=%d R=%d waves=%d best=%s kernel_ms=%.2f wall=%.3fs GCUPS=%.1f" % (
            st["profile_kernel"], m, n, st["strip_rows"]//64, st["waves"], best, st["kernel_ms"], dt, m*n/st["kernel_ms"]/1e6), flush=True)
    al.close()
if __name__ == "__main__":
    cfgs = sys.argv[1:]
    for c in cfgs:
        v = [int(x) for x in c.split(",")]
        perf(*v)

"""Quick on-box probe: correctness vs oracle on a few sizes + a first throughput number."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); oracle = g.load_oracle()

def check(m, n, related=True, rec=pkg.SMITH_WATERMAN, R=0, **kw):
    s0, s1 = (pkg.seqgen.related_pair if related else pkg.seqgen.unrelated_pair)(m, n, cfg=m % 97)
    al = pkg.MI355Aligner(device=0, rows_per_lane=R)
    al.setSequences(s0, s1)
    part = pkg.Partition(0, 0, m, n)
    al.streamBegin(part, recurrence_type=rec, want_last_column=True, want_last_row=True, **kw)
    while True:
        rows, fin = al.streamPoll()
        if fin: break
        time.sleep(0.001)
    lr = al.streamReadLastRow(); lc = al.streamReadColumn(0, m)
    best, nsp = al.streamEnd()
    st = al.getStatistics()
    ref = oracle.stage1(s0, s1, recurrence=rec, block_h=st["strip_rows"], block_w=997, want_last_row=True, want_last_col=True,
                        first_row_type=kw.get("first_row_init_type", 0), first_col_type=kw.get("first_column_init_type", 0))
    st["k"] = st["profile_kernel"]
    rb = ref["best"]; rb0 = (rb[0]-1, rb[1]-1, rb[2]) if rb[0] >= 0 else rb
    ok = (tuple(best) == tuple(rb0)) and np.array_equal(lr, ref["last_row"][1:]) and np.array_equal(lc, ref["last_col"][1:])
    print("k=%d m=%d n=%d R=%d rec=%d best=%s ref=%s row_ok=%s col_ok=%s -> %s" % (st["profile_kernel"], m, n, st["strip_rows"]//64, rec, best, rb0,
          np.array_equal(lr, ref["last_row"][1:]), np.array_equal(lc, ref["last_col"][1:]), "OK" if ok else "FAIL"), flush=True)
    al.close()
    return ok

def perf(m, n, R=0, waves=0):
    s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=2)
    al = pkg.MI355Aligner(device=0, rows_per_lane=R, waves=waves)
    al.setSequences(s0, s1)
    part = pkg.Partition(0, 0, m, n)
    for it in range(2):
        t0 = time.time()
        al.streamBegin(part)
        while True:
            rows, fin = al.streamPoll()
            if fin: break
            time.sleep(0.002)
        best, _ = al.streamEnd()
        dt = time.time() - t0
        st = al.getStatistics()
        print("perf m=%d n=%d R=%d waves=%d best=%s kernel_ms=%.2f wall=%.3fs GCUPS(kernel)=%.1f" % (
            m, n, st["strip_rows"]//64, st["waves"], best, st["kernel_ms"], dt, m*n/st["kernel_ms"]/1e6), flush=True)
    al.close()

if __name__ == "__main__":
    allok = True
    for (m, n) in [(100, 90), (513, 700), (5000, 4321), (2048, 64), (1, 1), (3000, 10000), (1025, 130), (257, 3)]:
        for R in (4, 8, 16, 32):
            allok &= check(m, n, R=R)
            allok &= check(m, n, R=R, force_int32=True)
    allok &= check(4000, 3000, rec=pkg.NEEDLEMAN_WUNSCH, first_row_init_type=pkg.INIT_WITH_GAPS, first_column_init_type=pkg.INIT_WITH_GAPS, R=8)
    allok &= check(20000, 20000, related=False, R=8)
    allok &= check(70000, 66000, related=True, R=8)
    allok &= check(70000, 66000, related=True, R=16)
    allok &= check(50000, 90000, related=True, R=4)
    print("ALL OK" if allok else "SOME FAILED", flush=True)
    if allok and len(sys.argv) > 1:
        perf(200000, 200000, R=8)
        perf(1000000, 1000000, R=8)
        perf(1000000, 1000000, R=4)

"""How long does a stream take to end once the host says stop (mi355sw_stream_abort = mustContinue() == 0)?
A tall NW partition with a gap-initialised first column that is RESIDENT on the device (every row there from the start:
as many strips in flight as there are wavefronts), stopped once `stop` rows are complete.
    python tools/stop_latency.py m n stop [rows_per_lane] [reps]
Prints per repetition: strips in flight at the stop (wavefronts), milliseconds from the abort call to the kernel's end
(host clock around polling hipEventQuery through streamPoll), kernel ms.  Round 5: ~13 ms with hundreds of strips in flight
(each gave up at ITS next poll of the host's word, its follower then ran to its own); round 6: one device word
(KernelArgs::stop_word) every strip looks at once per chunk.  Reference: AbstractDiagonalAligner::alignPartition tests
mustContinue() once per external diagonal (AbstractDiagonalAligner.cpp:64) -- all blocks end together."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import __graft_entry__ as g  # noqa: E402


def measure(pkg, al, m, n, stop, sw=False):
    part = pkg.Partition(0, 0, m, n)
    if sw:
        kw = dict(recurrence_type=pkg.SMITH_WATERMAN, track_best=True)
    else:
        kw = dict(recurrence_type=pkg.NEEDLEMAN_WUNSCH, track_best=False, first_row_init_type=pkg.INIT_WITH_GAPS,
                  first_column_init_type=pkg.INIT_WITH_GAPS)
    al.streamBegin(part, want_last_column=True, **kw)
    rows = 0
    while True:
        rows, fin = al.streamPoll()
        if rows >= stop or fin:
            break
    t0 = time.perf_counter()
    al.streamAbort()
    while True:
        rows2, fin = al.streamPoll()
        if fin:
            break
    t1 = time.perf_counter()
    al.streamEnd()
    st = al.getStatistics()
    return dict(rows_at_stop=int(rows), stop_ms=(t1 - t0) * 1e3, kernel_ms=st["kernel_ms"], strip_rows=st["strip_rows"],
                waves=st["waves"], strips=st["strips"], kernel=st["kernel"], processed_cells=st["processed_cells"])


def main():
    pkg = g.load_package()
    m, n, stop = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    R = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=9)
    al = pkg.MI355Aligner(device=0, rows_per_lane=R)
    al.setSequences(s0, s1)
    out = []
    for sw in (False, True):
        for rep in range(reps):
            r = measure(pkg, al, m, n, stop, sw)
            r["recurrence"] = "SW" if sw else "NW"
            out.append(r)
            print("%s %d x %d stop after %d rows: %d strips of %d rows, %d wavefronts: kernel gone %.3f ms after the stop (kernel %.1f ms) %s" % (
                r["recurrence"], m, n, r["rows_at_stop"], r["strips"], r["strip_rows"], r["waves"], r["stop_ms"], r["kernel_ms"], r["kernel"]), flush=True)
    al.close()
    print(json.dumps({"m": m, "n": n, "stop": stop, "runs": out}))


if __name__ == "__main__":
    main()

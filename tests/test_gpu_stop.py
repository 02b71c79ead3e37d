"""A stop every strip sees at once (reference: AbstractDiagonalAligner::alignPartition tests mustContinue() once per external
diagonal and all its blocks end together, AbstractDiagonalAligner.cpp:64; the callers that stop a sweep early are stage 2's
and stage 3's goal matching, M/stage2/sw_stage2.cpp:49-129).  Round 5's strips each polled the HOST's word at their own
64-chunk marks and gave up one after the other: 13 ms per stop with hundreds of strips in flight.  Now the first strip that
sees the host's word sets a device word (KernelArgs::stop_word) which every strip reads once per chunk and wherever it waits."""
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("recurrence", ["NW", "SW"])
def test_a_stop_ends_a_thousand_strips_within_a_millisecond_or_so(pkg, recurrence):
    """2 M x 700 k (a stage-2 sweep's width), 512-row strips, every row of the first column there from the start: by the time
    the first strips are complete every wavefront holds one (a strip follows the one above by ~46 us, a sweep takes ~100 ms).
    Then the host says stop.  Measured alone on the box: ~0.1-0.3 ms from the call to the kernel's end (round 5: 13 ms);
    asserted: 5 ms -- the host clock also sees whatever else runs on the GPU and its cores next to this test."""
    import stop_latency
    m, n = 2000000, 700000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=9)
    al = pkg.MI355Aligner(device=0, rows_per_lane=8)
    try:
        al.setSequences(s0, s1)
        runs = [stop_latency.measure(pkg, al, m, n, 512 * 40, sw=(recurrence == "SW")) for _ in range(3)]
    finally:
        al.close()
    best = min(r["stop_ms"] for r in runs)
    print("stop latency %s: %s ms with %d wavefronts, strips of %d rows" % (recurrence, [round(r["stop_ms"], 3) for r in runs], runs[0]["waves"], runs[0]["strip_rows"]))
    assert runs[0]["waves"] >= 1000 and runs[0]["strips"] > 3 * runs[0]["waves"]
    assert all(r["rows_at_stop"] < m // 2 for r in runs)                   # the stop came in mid-flight ...
    assert all(r["processed_cells"] < 0.5 * m * n for r in runs)           # ... and most of the partition never ran (what was complete or in flight)
    assert best < 5.0, runs


def test_a_stopped_stream_can_be_followed_by_an_exact_one(pkg, oracle):
    """the stop word lives in the stream's control block and is cleared with it: the stream after a stopped one computes
    everything and equals the oracle"""
    m, n = 300000, 3000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=4)
    al = pkg.MI355Aligner(device=0, rows_per_lane=4)
    try:
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        al.streamBegin(part, want_last_column=True)
        while al.streamPoll()[0] < 1024:
            time.sleep(0.0002)
        al.streamAbort()
        al.streamEnd()
        al.streamBegin(part, want_last_column=True)
        while not al.streamPoll()[1]:
            time.sleep(0.001)
        col = al.streamReadColumn(0, m)
        best, _ = al.streamEnd()
    finally:
        al.close()
    ref = oracle.stage1(s0, s1, want_last_col=True, threads=8)
    assert (best[0] + 1, best[1] + 1, best[2]) == tuple(ref["best"])       # (the stream reports the 0-based cell)
    assert np.array_equal(col, ref["last_col"][1:])

"""GPU: two strip heights in one launch (sw_strip_kernel_pk16_mixed: 1536- and 1408-row strips, chosen when the
strips would leave part of the last round of wavefronts idle).  Same best cell and same last row as the one-height
kernel and the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(al, pkg, m, n, **kw):
    part = pkg.Partition(0, 0, m, n)
    al.streamBegin(part, **kw)
    while not al.streamPoll()[1]:
        pass
    row = al.streamReadLastRow() if kw.get("want_last_row") else None
    best, _ = al.streamEnd()
    return best, row, al.getStatistics()


@pytest.mark.parametrize("kind,m,n,waves,expect_mixed", [("related", 21300, 12000, 8, True), ("unrelated", 33211, 12000, 8, True), ("related", 58000, 12000, 8, True), ("related", 51777, 12000, 8, False)])
def test_mixed_heights_equal_the_single_height_kernel_and_the_oracle(pkg, oracle, monkeypatch, kind, m, n, waves, expect_mixed):
    s0, s1 = (pkg.seqgen.related_pair if kind == "related" else pkg.seqgen.unrelated_pair)(m, n, cfg=81)
    ref = oracle.stage1(s0, s1, want_last_row=True)
    want = (ref["best"][0] - 1, ref["best"][1] - 1, ref["best"][2])
    res = {}
    for mixed in (True, False):
        if mixed:
            monkeypatch.delenv("MI355SW_NO_MIXED", raising=False)
        else:
            monkeypatch.setenv("MI355SW_NO_MIXED", "1")
        al = pkg.MI355Aligner(device=0, waves=waves)
        try:
            al.setSequences(s0, s1)
            # (the mixed form is the engine's own choice, taken when its cost model picks 1536-row strips that leave
            #  part of the last round idle: the shapes were chosen for that -- 14 of 16, 22 of 24 and 38 of 40 strips; 34 of 40 is refused: the last strip must hold the last row)
            best, row, st = _run(al, pkg, m, n, want_last_row=True)
            res[mixed] = (best, row, st["strips"], st["strip_rows"])
        finally:
            al.close()
    assert res[False][3] == 1536, "shape no longer picks 1536-row strips: choose another for this test"
    if expect_mixed:
        assert res[True][2] % waves == 0 and res[True][2] > res[False][2]   # a whole number of rounds, more (shorter) strips
    else:                                                                   # 40 strips of 1408 rows would leave the last ones past the matrix
        assert res[True][2] == res[False][2]
    for mixed in (True, False):
        assert tuple(res[mixed][0]) == want
        assert np.array_equal(res[mixed][1], ref["last_row"][1:])

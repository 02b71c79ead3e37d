"""GPU: mi355sw_align_partitions -- independent partitions side by side in ONE kernel launch (what stage 3 offers:
M/stage3/sw_stage3.cpp:210-262 refines every stage-2 partition on its own).  Every manager of a batch must receive
exactly what a call of its own would have given it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _managers(pkg, parts, kinds):
    out = []
    for p, kind in zip(parts, kinds):
        if kind == "nw":        # global: gap-initialised borders, last cell only
            out.append(pkg.Stage1Manager(p, alignment_start=pkg.AT_SEQUENCE_1_AND_2, alignment_end=pkg.AT_SEQUENCE_1_AND_2,
                                         special_row_interval=8192, keep_last_row=True, keep_last_column=True))
        elif kind == "semi":    # NW recurrence, best anywhere (scores dispatched)
            out.append(pkg.Stage1Manager(p, alignment_start=pkg.AT_SEQUENCE_1_AND_2, alignment_end=pkg.AT_ANYWHERE,
                                         keep_last_row=True, keep_last_column=True))
        else:                   # local
            out.append(pkg.Stage1Manager(p, special_row_interval=8192, keep_last_row=True, keep_last_column=True))
    return out


def _same(a, b):
    assert tuple(a.getBestScore()) == tuple(b.getBestScore())
    assert np.array_equal(a.lastRow(), b.lastRow())
    assert np.array_equal(a.lastColumn(), b.lastColumn())
    assert sorted(a.special_rows) == sorted(b.special_rows)
    for k in a.special_rows:
        assert np.array_equal(np.concatenate(a.special_rows[k]), np.concatenate(b.special_rows[k])), k


@pytest.mark.parametrize("R", [4, 8, 16])
def test_batched_partitions_equal_single_calls(pkg, oracle, R):
    """R = the strip height of the batch's one launch (mi355sw_config.batch_rows_per_lane: 256 rows for stage 3's small
    partitions, 512 / 1024 -- round 6, sw_batch_kernel_pk16<8,...> -- for batches of tall ones: stage 2's guessed sweeps)"""
    m, n = 60000, 50000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=71)
    rng = np.random.RandomState(5)
    parts, kinds = [], []
    for k in range(24):
        i0, j0 = int(rng.randint(0, m - 30000)), int(rng.randint(0, n - 9000))
        h, w = int(rng.randint(1, 30000)), int(rng.randint(1, 9000))
        parts.append(pkg.Partition(i0, j0, i0 + h, j0 + w))
        kinds.append(("nw", "semi", "sw")[k % 3])
    parts.append(pkg.Partition(0, 0, 1, 1)); kinds.append("nw")                   # the smallest partition there is
    parts.append(pkg.Partition(100, 100, 100, 4000)); kinds.append("nw")          # spans no cells: skipped (AlignerManager.cpp:96-99)
    al = pkg.MI355Aligner(device=0, rows_per_lane=R)
    try:
        al.setSequences(s0, s1)
        single = _managers(pkg, parts, kinds)
        for p, mg in zip(parts, single):
            al.alignPartition(p, mg)
        batch = _managers(pkg, parts, kinds)
        al.alignPartitions(parts, batch, **({} if R == 4 else {"rows_per_lane": R}))
        st = al.getStatistics()
        al.unsetSequences()
    finally:
        al.close()
    assert st["kernel_launches"] >= 1 and st["strip_rows"] == 64 * R
    assert st["kernel"].startswith("sw_batch_kernel_pk16<%d," % (R // 2))
    for k, (a, b) in enumerate(zip(single, batch)):
        if parts[k].getHeight() == 0 or parts[k].getWidth() == 0:
            continue
        _same(a, b)
    # and one of them against the oracle, so that "equal" is not "equally wrong"
    p = parts[0]
    ref = oracle.stage1(s0[p.i0:p.i1], s1[p.j0:p.j1], recurrence=oracle.NEEDLEMAN_WUNSCH, first_row_type=oracle.INIT_WITH_GAPS,
                        first_col_type=oracle.INIT_WITH_GAPS, want_last_row=True, want_last_col=True, best_mode=oracle.BEST_LAST_CELL)
    assert np.array_equal(batch[0].lastRow(), ref["last_row"]) and np.array_equal(batch[0].lastColumn(), ref["last_col"])


def test_batch_with_an_overflow_falls_back_for_that_partition_only(pkg, monkeypatch):
    """a partition whose packed run reports an overflow (fault injection) is re-run on the int32 kernels after the
    batch; its neighbours are not disturbed"""
    m, n = 9000, 7000
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=72)
    parts = [pkg.Partition(0, 0, 4000, 3000), pkg.Partition(1000, 500, 9000, 6500), pkg.Partition(200, 100, 700, 6900)]
    kinds = ["nw", "sw", "semi"]
    al = pkg.MI355Aligner(device=0, rows_per_lane=4)
    try:
        al.setSequences(s0, s1)
        single = _managers(pkg, parts, kinds)
        for p, mg in zip(parts, single):
            al.alignPartition(p, mg)
        monkeypatch.setenv("MI355SW_FAULT_OVERFLOW_STRIP", "9")      # only the second partition has a strip 9 ... the first too
        batch = _managers(pkg, parts, kinds)
        al.alignPartitions(parts, batch)
        monkeypatch.delenv("MI355SW_FAULT_OVERFLOW_STRIP")
    finally:
        al.close()
    for a, b in zip(single, batch):
        _same(a, b)

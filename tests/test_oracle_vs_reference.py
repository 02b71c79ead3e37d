"""CPU, build container only: the C restatement against LIVE runs of the reference's MASA-Core CPU
path (oracle/_ref/ref_driver).  Skipped where oracle/_ref was not built."""
import numpy as np
import pytest

from helpers import flush_interval


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref/ref_driver not built (needs /root/reference)")
    return oracle


@pytest.mark.parametrize("m,n,bh,bw,cfg", [(1500, 1700, 100, 130, 31), (2049, 511, 128, 128, 32), (640, 3000, 64, 1000, 33)])
def test_special_rows_and_best_live(ref, pkg, m, n, bh, bw, cfg):
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=cfg)
    r = ref.run_ref(s0, s1, ["--stage-1", "--disk-size=100K", "--no-block-pruning", "--block=%d,%d" % (bh, bw)])
    o = ref.stage1(s0, s1, block_h=bh, block_w=bw, special_row_interval=flush_interval(m, n, 100 * 1024), want_last_row=True)
    assert tuple(r["best"]) == tuple(o["best"])
    ids = o["special_row_ids"]
    assert len(r["special_rows"]) >= 1
    for (d, i), row in r["special_rows"].items():
        mine = o["special_rows"][ids.index(i)] if i in ids else o["last_row"]
        assert np.array_equal(row, mine), "row %d" % i


def test_pruning_live(ref, pkg):
    s0, s1 = pkg.seqgen.related_pair(5000, 5000, cfg=34)
    r = ref.run_ref(s0, s1, ["--stage-1", "--no-flush", "--block=200,200"])
    o = ref.stage1(s0, s1, block_h=200, block_w=200, pruning=True)
    assert tuple(r["best"]) == tuple(o["best"])
    assert o["blocks_pruned"] > 0
    off = ref.stage1(s0, s1, block_h=200, block_w=200, pruning=False)
    assert off["best"] == o["best"]

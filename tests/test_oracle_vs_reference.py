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


@pytest.mark.parametrize("m,n,bh,bw,edges", [(3000, 2500, 512, 700, None), (2049, 1000, 256, 333, None), (1800, 2100, 256, 512, "++")])
def test_block_scores_live(ref, pkg, m, n, bh, bw, edges):
    """--dump-blocks: the best score of every block as MASA-Core's BlocksFile stores it (AlignerManager.cpp:418-423,
    BlocksFile.cpp:44-64) = the oracle's block table -- which is what pins the engine's block scores on the GPU."""
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=35)
    args = ["--stage-1", "--no-flush", "--no-block-pruning", "--block=%d,%d" % (bh, bw), "--dump-blocks"]
    kw = {}
    if edges == "++":
        args.append("--edges=++")
        kw = dict(recurrence=ref.NEEDLEMAN_WUNSCH, first_row_type=ref.INIT_WITH_GAPS, first_col_type=ref.INIT_WITH_GAPS,
                  best_mode=ref.BEST_LAST_CELL)
    r = ref.run_ref(s0, s1, args)
    o = ref.stage1(s0, s1, block_h=bh, block_w=bw, **kw)
    gh, gw = o["grid"]
    assert r["blocks"].shape == (gh, gw)
    mine = np.array([[o["block_scores"][(bx, by)][2] for bx in range(gw)] for by in range(gh)])
    if edges == "++":     # the last cell is dispatched as the score of the last block (AbstractBlockAligner.cpp:343-346 after :330-338)
        mine[-1, -1] = o["best"][2]
    assert np.array_equal(mine, r["blocks"])

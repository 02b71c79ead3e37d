"""GPU: per-block scores (config.block_score_columns) -- what CUDAligner::getBlockScores +
AbstractDiagonalAligner::flushBlockScores hand to AlignerManager::dispatchScore(score, bx, by)
(X/CUDAligner.cpp:441-452, AbstractDiagonalAligner.cpp:392-403, AlignerManager.cpp:411-450; consumer: --dump-blocks).
The engine's grid is "strip x W columns"; every block's best cell must be the oracle's for the same grid."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _blocks(oracle, ref):
    return {k: (v[0] + 1, v[1] + 1, v[2]) if v[2] > -oracle.INF else None for k, v in ref["block_scores"].items()}


@pytest.mark.parametrize("kind,m,n,R,W,flags", [
    ("related", 5000, 4321, 4, 700, 0),          # ragged last strip and last column block
    ("unrelated", 3000, 3100, 4, 512, 0),        # low scores: ties everywhere, the canonical cell must win in every block
    ("related", 9000, 7000, 8, 1000, 0),
    ("related", 5000, 4321, 4, 700, 2),          # int32 kernels
    ("semiglobal", 4000, 3500, 4, 600, 0),       # NW recurrence, gap-initialised borders, scores anywhere
])
def test_every_block_best_equals_the_oracles(pkg, oracle, kind, m, n, R, W, flags):
    if kind == "unrelated":
        s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=51)
    else:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=52)
    al = pkg.MI355Aligner(device=0, rows_per_lane=R, flags=flags, block_score_columns=W)
    try:
        assert al.getCapabilities()["dispatch_block_scores"] == 1
        al.setSequences(s0, s1)
        part = pkg.Partition(0, 0, m, n)
        if kind == "semiglobal":
            mg = pkg.Stage1Manager(part, alignment_start=pkg.AT_SEQUENCE_1_AND_2, alignment_end=pkg.AT_ANYWHERE, keep_last_row=True)
            ref = oracle.stage1(s0, s1, recurrence=oracle.NEEDLEMAN_WUNSCH, first_row_type=oracle.INIT_WITH_GAPS,
                                first_col_type=oracle.INIT_WITH_GAPS, block_h=64 * R, block_w=W, want_last_row=True)
        else:
            mg = pkg.Stage1Manager(part, keep_last_row=True)
            ref = oracle.stage1(s0, s1, block_h=64 * R, block_w=W, want_last_row=True)
        al.alignPartition(part, mg)
        st = al.getStatistics()
        al.unsetSequences()
    finally:
        al.close()
    assert st["strip_rows"] == 64 * R
    gh, gw = ref["grid"]
    assert (gh, gw) == (-(-m // (64 * R)), -(-n // W))
    want = _blocks(oracle, ref)
    got = {k: (v if v[2] > -pkg.INF else None) for k, v in mg.block_scores.items()}
    assert set(got) == set(want)
    bad = [(k, got[k], want[k]) for k in sorted(want) if got[k] != want[k]]
    assert not bad, bad[:5]
    # the main pass is untouched by the second sweep: best cell and last row as without block scores
    assert tuple(mg.getBestScore()) == tuple(ref["best"])
    assert np.array_equal(mg.lastRow(), ref["last_row"])

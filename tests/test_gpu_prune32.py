"""GPU (-m gpu): block pruning in the int32 kernel family (round 5) -- the pairs the packed kernel cannot take (more than 14
byte values common to both sequences: IUPAC-rich FASTA), MI355SW_F_FORCE_INT32 and the reruns after an overflow report.
Reference: the reference prunes in every instantiation of its kernels (X/CUDAligner.cu:950-960, AbstractBlockPruning.cpp:70-111)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
IUPAC = np.frombuffer(b"ACGTNRYKMSWBDHV", dtype=np.uint8)          # 15 letters: one too many for the packed kernel's profile


def _iupac_pair(pkg, m, n, cfg):
    """a related ACGT pair with every 50th base of both sequences replaced by an ambiguity code (the same code in both where the
    positions coincide: raw byte equality is what scores, X/CUDAligner.cu:276-289)"""
    s0, s1 = pkg.seqgen.related_pair(m, n, cfg=cfg)
    s0, s1 = s0.copy(), s1.copy()
    rng = np.random.default_rng(cfg)
    for s in (s0, s1):
        pos = np.arange(7, len(s), 50)
        s[pos] = IUPAC[4 + rng.integers(0, 11, size=len(pos))]
    return s0, s1


@pytest.mark.parametrize("mode", ["iupac", "forced"])
def test_int32_family_prunes_and_keeps_the_oracles_best(pkg, oracle, mode):
    from helpers import oracle_full
    from masa_cudalign_amd.engine import F_FORCE_INT32
    m, n = 60000, 50000
    if mode == "iupac":
        s0, s1 = _iupac_pair(pkg, m, n, 801)
        flags = 0
    else:
        s0, s1 = pkg.seqgen.related_pair(m, n, cfg=802)
        flags = F_FORCE_INT32
    ref = oracle_full(oracle, s0, s1)
    want_rows = dict(zip(ref["special_row_ids"], ref["special_rows"]))
    for R in (0, 4, 16):
        al = pkg.MI355Aligner(device=0, rows_per_lane=R, flags=flags)
        try:
            al.setSequences(s0, s1)
            part = pkg.Partition(0, 0, m, n)
            out = {}
            for prune in (False, True):
                mg = pkg.Stage1Manager(part, special_row_interval=8192, keep_last_row=True, keep_last_column=True, block_pruning=prune)
                al.alignPartition(part, mg)
                st = al.getStatistics()
                assert st["profile_kernel"] in (0, 1) and st["kernel"].startswith("sw_strip_kernel<"), st["kernel"]      # the int32 family
                out[prune] = (tuple(mg.getBestScore()), st, mg)
            assert out[False][0] == out[True][0] == tuple(ref["best"])
            assert out[False][1]["pruned_cells"] == 0
            st = out[True][1]
            assert st["kernel"].endswith(",true>") and st["pruned_cells"] > 0.15 * m * n, st
            assert st["pruned_cells"] + st["processed_cells"] == m * n
            mgp = out[True][2]
            assert np.all(mgp.lastRow() <= ref["last_row"]) and np.all(mgp.lastColumn() <= ref["last_col"])
            for i in sorted(mgp.special_rows):
                if i not in want_rows:                  # (the manager keeps the last row under its row number too)
                    continue
                got, want = mgp.specialRow(i), want_rows[i]
                assert np.all(got <= want) and np.all(got[1:, 0] >= 0), i
                if i <= ref["best"][0]:
                    assert got[:, 0].max() == want[:, 0].max() and got[:, 0].argmax() == want[:, 0].argmax(), i
        finally:
            al.close()


def test_int32_pruning_with_an_initial_bound_and_the_tie_path(pkg, oracle):
    """the bound handed in as initial_bound (optimum and optimum + 1 -> MI355SW_EBOUND) and the second-copy path that can only tie
    (tests/test_gpu_bound.py) on the int32 kernels"""
    import time
    from masa_cudalign_amd.engine import F_FORCE_INT32, AlignerError
    from test_gpu_bound import _pairs
    s0, s1 = _pairs(pkg, "ties")
    M, N = 36864, 12864
    ref = oracle.stage1(s0[:M], s1[:N], want_last_row=True)
    want = int(ref["last_row"][-1, 0])
    part = pkg.Partition(0, 0, M, N)
    for R in (4, 8, 16):
        al = pkg.MI355Aligner(device=0, rows_per_lane=R, flags=F_FORCE_INT32)
        try:
            al.setSequences(s0, s1)
            for bound in (None, want, want + 1):
                al.streamBegin(part, prune_blocks=True, want_last_row=True, initial_bound=bound)
                while not al.streamPoll()[1]:
                    time.sleep(0.001)
                lr = al.streamReadLastRow()
                if bound == want + 1:
                    with pytest.raises(AlignerError, match="EBOUND"):
                        al.streamEnd()
                    continue
                best, _ = al.streamEnd()
                st = al.getStatistics()
                assert (best[0] + 1, best[1] + 1, best[2]) == tuple(ref["best"]), (R, bound)
                assert int(lr[-1, 0]) == want and np.all(lr <= ref["last_row"][1:]) and st["pruned_cells"] > 0.3 * M * N, (R, bound)
        finally:
            al.close()

"""the anchored seed on the small pairs of tests/test_gpu_bound.py, every anchor and segment printed"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as g
pkg = g.load_package()
oracle = g.load_oracle()
from masa_cudalign_amd.engine import NEEDLEMAN_WUNSCH, SMITH_WATERMAN, V_MESSAGES, V_SEED_TILES
from test_gpu_bound import _pairs
from helpers import oracle_full

for kind in sys.argv[1:] or ["ties", "related", "inversion"]:
    s0, s1 = _pairs(pkg, kind)
    m, n = len(s0), len(s1)
    ref = oracle_full(oracle, s0, s1)
    print(kind, m, n, "oracle best", ref["best"], flush=True)
    al = pkg.MI355Aligner(device=0, rows_per_lane=4, verbosity=V_MESSAGES | V_SEED_TILES)
    al.setSequences(s0, s1)
    for rec in (SMITH_WATERMAN, NEEDLEMAN_WUNSCH):
        print(" seedBound", rec, al.seedBound(pkg.Partition(0, 0, m, n), rec), flush=True)
    al.close()

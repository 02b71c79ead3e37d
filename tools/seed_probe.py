"""The first bound of a pruning run on a related M x N pair: the anchored seed (segments between anchors, side by side) against
the staircase (one chain of tiles, MI355SW_F_STAIRCASE_SEED) -- value and milliseconds, local and global.
python tools/seed_probe.py M N [cfg]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
from masa_cudalign_amd.engine import F_STAIRCASE_SEED, NEEDLEMAN_WUNSCH, SMITH_WATERMAN, V_MESSAGES, V_SEED_TILES

m, n = int(sys.argv[1]), int(sys.argv[2])
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 5
s0, s1 = pkg.seqgen.related_pair(m, n, cfg=cfg)
part = pkg.Partition(0, 0, m, n)
for name, flags in (("anchored", 0), ("staircase", F_STAIRCASE_SEED), ("anchored", 0)):
    al = pkg.MI355Aligner(device=0, flags=flags, verbosity=V_MESSAGES | (V_SEED_TILES if os.environ.get("SEED_TILES") else 0))
    al.setSequences(s0, s1)
    for rec, rn in ((SMITH_WATERMAN, "local"), (NEEDLEMAN_WUNSCH, "global")):
        t0 = time.time()
        b = al.seedBound(part, rec)
        print("%-10s %-6s bound %s in %.0f ms" % (name, rn, b, (time.time() - t0) * 1e3), flush=True)
    al.close()

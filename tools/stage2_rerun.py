"""Stage 2 alone, again and again, on a work directory in which stage 1 has run (tools/native_pipeline_run.py with
MI355SW_WORK set): python tools/stage2_rerun.py M N sra_bytes cfg [repeats]   (MI355SW_WORK = the work directory)
Each repeat removes what the stage wrote the time before.  For timing experiments with the engine's environment knobs."""
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
import importlib  # noqa: E402
from masa_cudalign_amd import fasta  # noqa: E402
stage2_mod = importlib.import_module("masa_cudalign_amd.stage2")      # (the package exports the function under the same name)

m, n, limit, cfg = int(sys.argv[1]), int(sys.argv[2]), int(float(sys.argv[3])), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 1
work = os.environ["MI355SW_WORK"]
s0, s1 = pkg.seqgen.related_pair(m, n, cfg=cfg)
q0 = fasta.Sequence(">s0", s0, fasta.SequenceModifiers())
q1 = fasta.Sequence(">s1", s1, fasta.SequenceModifiers())
import numpy as np
d0, d1 = np.ascontiguousarray(q0.data()), np.ascontiguousarray(q1.data())
al = pkg.MI355Aligner(device=0)
try:
    for k in range(reps):
        shutil.rmtree(os.path.join(work, "special_rows", "stage.02.00"), ignore_errors=True)
        t0 = time.time()
        r = stage2_mod.stage2(al, d0, d1, work, sra_limit=limit)
        print("stage 2, run %d: %.2f s, %d partitions, %d crosspoints" % (k, time.time() - t0, r["partitions"], len(r["crosspoints"])), flush=True)
finally:
    al.close()

"""The whole pipeline natively: stage 1 (score + special rows), stage 2 (crosspoints on stage 1's special rows),
stage 3 (crosspoints on the rows stages 2 and 3 save), stage 4 (Myers-Miller refinement down to 16 x 16, on the GPU:
mi355sw_stage4), stage 5 (exact alignment of the small partitions), stage 6 (text) -- what MASA-Core's
executeTraceback() runs after stage 1 (M/libmasa/libmasa.cpp:643-657), with the work directory in MASA-Core's layout:

    <work>/crosspoints/crosspoint_01.00 .. crosspoint_04.00 (+ crosspoint_03.00.rNN per round of stage 3)
    <work>/special_rows/stage.01.00, stage.02.00, stage.03.00.rNN
    <work>/info, <work>/status, <work>/alignment.00.bin (alignment_file.py), <work>/alignment.00.txt

The aligner is an MI355Aligner (or anything with its setSequences / alignPartition / matchLastColumn / stage4 /
unsetSequences); every DP cell of every stage is computed by it.  The sequence modifiers of fasta.py (--trim,
--reverse, --complement, --clear-n) are carried through all stages as in MASA-Core."""
import os
import time

import numpy as np

from .manager import AT_ANYWHERE
from .engine import INF
from .crosspoints import Crosspoint, CrosspointsFile, crosspoint_file, save_array
from .stage1 import stage1
from .stage2 import stage2
from .stage3 import stage3
from . import stage56, alignment_file
from . import sra as sra_mod


class WorkDirectoryMismatch(RuntimeError):
    """MASA-Core prints "Sequence mismatch from previous run. Try cleaning work directory (--clean)" and stops"""


def check_work_directory(work, seq0, seq1):
    """Job::initialize (M/common/Job.cpp:68-90): `<work>/info` names the two sequences a work directory belongs to
    ("seq0=<description>", "seq1=<description>"); a run that finds other names there must not continue from its
    special rows, status and crosspoints"""
    os.makedirs(work, exist_ok=True)
    fn = os.path.join(work, "info")
    if os.path.exists(fn):
        prop = {}
        for line in open(fn, encoding="latin-1"):
            line = line[:-1]
            pos = line.find("=")
            if pos > 0:
                prop[line[:pos]] = line[pos + 1:]
        bad = [(k, prop.get(k, ""), s.description) for k, s in (("seq0", seq0), ("seq1", seq1)) if prop.get(k, "") != s.description]
        if bad:
            raise WorkDirectoryMismatch("sequence mismatch from a previous run in %s (clean the work directory): %s"
                                        % (work, "; ".join("%s: %r != %r" % b for b in bad)))
    else:
        with open(fn, "w", encoding="latin-1") as f:
            f.write("seq0=%s\nseq1=%s\n" % (seq0.description, seq1.description))


@sra_mod.with_async_files
def align(aligner, seq0, seq1, work, alignment_start=AT_ANYWHERE, alignment_end=AT_ANYWHERE, sra_limit=0,
          block_pruning=True, max_partition_size=16, progress=None, max_alignments=1, ram_limit=0, prune_global=False):
    """seq0, seq1: fasta.Sequence.  Returns {"best", "alignment": stage56.Alignment or None, "text": bytes of
    alignment.00.txt or None when nothing scored above the floor, "crosspoints": {2: n, 3: n, 4: n},
    "seconds": {stage: s}}; with max_alignments > 1 also "alignments": one such record per end point stage 1 kept
    (alignment.NN.txt, crosspoint_0S.NN, special_rows/stage.0S.NN: executeTraceback, libmasa.cpp:643-657)"""
    check_work_directory(work, seq0, seq1)
    # the data the aligner compares: forward or reversed, complemented, N-cleared (fasta.py); --trim only selects the
    # part of the matrix stage 1 sweeps, every coordinate of every stage stays absolute (Sequence.cpp:117-159)
    d0, d1 = np.ascontiguousarray(seq0.data()), np.ascontiguousarray(seq1.data())
    bounds = (seq0.offset0 - 1, seq1.offset0 - 1, seq0.offset1, seq1.offset1)
    secs = {}
    areas = {}                     # Job's cache of special-rows areas: rows kept in memory live here between the stages
    t = time.time()
    r1 = stage1(aligner, d0, d1, work, alignment_start=alignment_start, alignment_end=alignment_end, sra_limit=sra_limit,
                block_pruning=block_pruning, bounds=bounds, progress=progress, max_alignments=max_alignments,
                ram_limit=ram_limit, areas=areas, prune_global=prune_global)
    secs[1] = time.time() - t
    out = {"best": r1["best"], "alignment": None, "text": None, "crosspoints": {}, "seconds": secs, "stage1": r1,
           "alignments": []}
    if r1["best"] is None or r1["best"][2] <= -INF or r1["best"][0] < 0:
        return out                                        # an empty best-score list: MASA-Core runs no traceback either
    for ident in range(max(len(r1.get("bests", [])), 1)):
        rec = _traceback(aligner, seq0, seq1, d0, d1, work, ident, alignment_start, sra_limit, max_partition_size, bounds, secs,
                         ram_limit, areas)
        out["alignments"].append(rec)
    out.update(out["alignments"][0])
    return out


def _traceback(aligner, seq0, seq1, d0, d1, work, ident, alignment_start, sra_limit, max_partition_size, bounds, secs,
               ram_limit=0, areas=None):
    """stages 2-6 for the alignment that ends in crosspoint_01.<ident>"""
    def clock(stage, t0):
        secs[stage] = secs.get(stage, 0.0) + time.time() - t0
    t = time.time()
    r2 = stage2(aligner, d0, d1, work, alignment_start=alignment_start, sra_limit=sra_limit, ident=ident, bounds=bounds,
                ram_limit=ram_limit, areas=areas)
    clock(2, t)
    t = time.time()
    r3 = stage3(aligner, d0, d1, work, sra_limit=sra_limit, ident=ident, ram_limit=ram_limit, areas=areas)
    clock(3, t)
    t = time.time()
    aligner.setSequences(d0, d1)
    try:
        try:
            cp4, st4 = aligner.stage4(r3["crosspoints"], max_partition_size, as_array=True)
        except TypeError:              # an aligner double without the array form
            cp4, st4 = aligner.stage4(r3["crosspoints"], max_partition_size)
    finally:
        aligner.unsetSequences()
    # crosspoint_04: millions of points at sizes like C3 (130 MB of text, seconds of formatting).  Nothing in this run reads the
    # file back -- stage 5 takes the array -- so it is written by the areas' file thread while stages 5 and 6 run in the library
    # (their calls release the interpreter); in place when align() returns, like every queued file (sra.async_files)
    if isinstance(cp4, np.ndarray):
        sra_mod.queue_file_operation(_traceback, save_array, crosspoint_file(work, 4, ident), cp4)
    else:
        save_array(crosspoint_file(work, 4, ident), cp4)
    clock(4, t)
    t = time.time()
    al = stage56.stage5(seq0, seq1, cp4)
    clock(5, t)
    t = time.time()
    with open(os.path.join(work, "alignment.%02d.bin" % ident), "wb") as f:   # what stage 5 leaves for stage 6 and the viewers
        f.write(alignment_file.dumps(al, seq0, seq1))
    text = stage56.stage6_text(al, seq0, seq1)
    with open(os.path.join(work, "alignment.%02d.txt" % ident), "wb") as f:
        f.write(text)
    clock(6, t)
    return dict(alignment=al, text=text, crosspoints={2: len(r2["crosspoints"]), 3: len(r3["crosspoints"]), 4: len(cp4)},
                stage2=r2, stage3=r3, stage4=st4)

"""Stages 5 and 6 of the pipeline, natively: from the refined crosspoints (stage 4) to the alignment text.

Restates MASA-Core's M/stage5/sw_stage5.cpp (sw() :83-319: exact alignment of every partition of at most 16 x 16 with a
full-matrix traceback that honours the crosspoint types; stage5() :322-485), M/common/biology/Alignment.cpp (gap lists:
addGap :207-218, finalize :120-130) and the default text output of M/stage6/sw_stage6.cpp (printText :60-262).
Host code like the reference's (single-threaded CPU code there): the per-partition matrices of stage 5 run in the C
library (csrc/stage5.cpp), the gap lists and the text are assembled here.

    crosspoints: [(type, i, j, score), ...] of crosspoint_04 (type 0 aligned, 1 gap in S0 -- the path moves along S1 --,
                 2 gap in S1), i/j = DP coordinates
    seq0, seq1 : fasta.Sequence objects (description, forward data, modifiers)
"""
import numpy as np

GAP_OPEN, GAP_EXT, MATCH, MISMATCH = 3, 2, 1, -3
GAP_FIRST = GAP_OPEN + GAP_EXT
INF = 999999999
TYPE_MATCH, TYPE_GAP_1, TYPE_GAP_2 = 0, 1, 2


class Alignment:
    """M/common/biology/Alignment.cpp: two gap lists [(pos, len)], start / end positions (1-based, absolute)"""

    def __init__(self):
        self.gaps = ([], [])
        self.start, self.end = [-1, -1], [-1, -1]
        self.raw_score = self.matches = self.mismatches = self.gap_open = self.gap_extensions = 0

    def add_gap(self, seq, pos):                       # :207-218
        g = self.gaps[seq]
        if g and g[-1][0] == pos:
            g[-1][1] += 1
        else:
            g.append([pos, 1])

    def finalize(self):                                # :120-130
        for g in self.gaps:
            g.sort(key=lambda x: x[0])


def _gap_list(positions):
    """Alignment::addGap (:207-218) over a run of events -- an event at the position of the entry before it extends that
    entry -- then Alignment::finalize's sort by position (:120-130; stable, as list.sort was)"""
    if len(positions) == 0:
        return []
    p = np.asarray(positions, dtype=np.int64)
    starts = np.flatnonzero(np.concatenate(([True], p[1:] != p[:-1])))
    lens = np.diff(np.concatenate((starts, [len(p)])))
    pos = p[starts]
    order = np.argsort(pos, kind="stable")
    return [[int(a), int(b)] for a, b in zip(pos[order], lens[order])]


def stage5(seq0, seq1, crosspoints):
    """stage5(), sw_stage5.cpp:322-485: the Alignment of the path through `crosspoints`.  The per-partition matrices and
    tracebacks (sw(), :83-319) run in the library (mi355sw_stage5, csrc/stage5.cpp: host code like the reference's, a
    10 M-column alignment takes well under a second); here the gap events become Alignment.cpp's gap lists."""
    from .engine import stage5_events, AlignerError
    al = Alignment()
    try:
        rows0, cols1, tot = stage5_events(seq0.data(), seq1.data(), crosspoints)
    except AlignerError as e:
        if "ETOOLARGE" in str(e):
            raise ValueError("stage5: a partition is larger than the reference's W_MAX; run stage 4 first")
        raise RuntimeError(str(e))
    # _dot (:64-80): DP row / column -> absolute position in the original sequence (Sequence::getAbsolutePos)
    p0 = rows0.astype(np.int64) + (0 if seq0.modifiers.reverse else 1)
    p1 = cols1.astype(np.int64) + (0 if seq1.modifiers.reverse else 1)
    if seq0.modifiers.reverse:
        p0 = seq0.original_size + 1 - p0
    if seq1.modifiers.reverse:
        p1 = seq1.original_size + 1 - p1
    al.gaps = (_gap_list(p0), _gap_list(p1))
    start, end = crosspoints[0], crosspoints[-1]
    if len(crosspoints) != 1:
        al.start = [seq0.absolute_pos(start[1] + 1), seq1.absolute_pos(start[2] + 1)]
        al.end = [seq0.absolute_pos(end[1]), seq1.absolute_pos(end[2])]
    correct = end[3] - start[3]
    if correct != tot["score"]:
        raise RuntimeError("stage5: Wrong Alignment Score: %d != %d" % (tot["score"], correct))
    al.raw_score, al.matches, al.mismatches = tot["score"], tot["matches"], tot["mismatches"]
    al.gap_open, al.gap_extensions = tot["gap_open"], tot["gap_extensions"]
    return al


def stage6_text(al, seq0, seq1):
    """printText, sw_stage6.cpp:60-262 -- the bytes of alignment.NN.txt"""
    out = []
    d0, d1 = seq0.forward, seq1.forward
    out.append("Query: %s " % seq0.description)
    out.append("(%d)\n" % len(seq0) if seq0.original_size == len(seq0) else "[%d..%d](%d)\n" % (seq0.offset0, seq0.offset1, len(seq0)))
    out.append("Sbjct: %s " % seq1.description)
    # (the reference compares SEQUENCE 0's size with sequence 1's length here, :76)
    out.append("(%d)\n" % len(seq1) if seq0.original_size == len(seq1) else "[%d..%d](%d)\n" % (seq1.offset0, seq1.offset1, len(seq1)))
    out.append("\n")
    # the blocks and the summary: csrc/stage6.cpp (mi355sw_stage6_text), host code like the reference's
    from .engine import stage6_body
    return "".join(out).encode("latin-1") + stage6_body(d0, d1, al.start, al.end, al.gaps[0], al.gaps[1], al.raw_score)

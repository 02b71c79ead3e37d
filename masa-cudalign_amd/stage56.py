"""Stages 5 and 6 of the pipeline, natively: from the refined crosspoints (stage 4) to the alignment text.

Restates MASA-Core's M/stage5/sw_stage5.cpp (sw() :83-319: exact alignment of every partition of at most 16 x 16 with a
full-matrix traceback that honours the crosspoint types; stage5() :322-485), M/common/biology/Alignment.cpp (gap lists:
addGap :207-218, finalize :120-130) and the default text output of M/stage6/sw_stage6.cpp (printText :60-262).
Pure host code (the reference's is single-threaded CPU code too): ~0.3 % of the pipeline's time on top of the engine.

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


def _dot(al, seq0, seq1, i, j, typ):
    """sw_stage5.cpp:64-80"""
    if typ == 1:
        al.add_gap(1, seq1.absolute_pos(j + (0 if seq1.modifiers.reverse else 1)))
    elif typ == 2:
        al.add_gap(0, seq0.absolute_pos(i + (0 if seq0.modifiers.reverse else 1)))


def _sw(al, seq0, seq1, d0, d1, i0, j0, i1, j1, type_s, type_e, tot):
    """sw_stage5.cpp:83-319; d0/d1 = the data arrays (Sequence::getData()), rows (i0, i1], columns (j0, j1]"""
    if i0 == i1:
        s = (j1 - j0) * -GAP_EXT
        if type_s != TYPE_GAP_1:
            tot["gapOpen"] += 1
            s += -GAP_OPEN
        for j in range(j1, j0, -1):
            _dot(al, seq0, seq1, i0, j, 2)
            tot["gapExtensions"] += 1
        tot["score"] += s
        return s
    if j0 == j1:
        s = (i1 - i0) * -GAP_EXT
        if type_s != TYPE_GAP_2:
            tot["gapOpen"] += 1
            s += -GAP_OPEN
        for i in range(i1, i0, -1):
            _dot(al, seq0, seq1, i, j0, 1)
            tot["gapExtensions"] += 1
        tot["score"] += s
        return s
    rows, cols = i1 - i0, j1 - j0
    # (with an aligned end point the reference computes one more row and column and steps back over them before the
    #  traceback starts, :104-108 / :186-190: the traceback never looks at them)
    a, b = d0[i0:i1], d1[j0:j1]
    h = [[0] * (cols + 1) for _ in range(rows + 1)]
    e = [[-INF] * (cols + 1) for _ in range(rows + 1)]
    f = [[-INF] * (cols + 1) for _ in range(rows + 1)]
    for j in range(1, cols + 1):
        h[0][j] = -j * GAP_EXT - GAP_OPEN * (type_s != TYPE_GAP_1)
    h[0][0] = -INF if type_s != 0 else 0
    for i in range(1, rows + 1):
        hi, hp, ei, ep, fi = h[i], h[i - 1], e[i], e[i - 1], f[i]
        hi[0] = -i * GAP_EXT - GAP_OPEN * (type_s != TYPE_GAP_2)
        s = a[i - 1]
        for j in range(1, cols + 1):
            ev = max(hp[j] - GAP_FIRST, ep[j] - GAP_EXT)
            fv = max(hi[j - 1] - GAP_FIRST, fi[j - 1] - GAP_EXT)
            ei[j], fi[j] = ev, fv
            hi[j] = max(hp[j - 1] + (MATCH if s == b[j - 1] else MISMATCH), ev, fv)
    i, j = rows, cols
    c = {0: TYPE_MATCH, TYPE_GAP_2: TYPE_GAP_2, TYPE_GAP_1: TYPE_GAP_1}[type_e]
    total = 0
    while i > 0 and j > 0:
        _eh = h[i - 1][j] - GAP_FIRST
        _fh = h[i][j - 1] - GAP_FIRST
        _h11 = h[i - 1][j - 1] + (MATCH if a[i - 1] == b[j - 1] else MISMATCH)
        _h10, _h01, _h00 = e[i][j], f[i][j], h[i][j]
        if c == 0:
            if _h00 == _h11:
                d, c = 0, TYPE_MATCH
            elif _h00 == _h10:
                d, c = 1, (TYPE_MATCH if _h10 == _eh else TYPE_GAP_2)
            elif _h00 == _h01:
                d, c = 2, (TYPE_MATCH if _h01 == _fh else TYPE_GAP_1)
            else:
                raise RuntimeError("stage5: traceback lost at (%d,%d)" % (i0 + i, j0 + j))
        elif c == TYPE_GAP_2:
            d, c = 1, (TYPE_MATCH if _h10 == _eh else TYPE_GAP_2)
        else:
            d, c = 2, (TYPE_MATCH if _h01 == _fh else TYPE_GAP_1)
        _dot(al, seq0, seq1, i0 + i, j0 + j, d)
        if d == 0:
            if a[i - 1] == b[j - 1]:
                tot["matches"] += 1
                total += MATCH
            else:
                tot["mismatches"] += 1
                total += MISMATCH
            i -= 1
            j -= 1
        else:
            if c == TYPE_MATCH:
                tot["gapOpen"] += 1
                total += -GAP_FIRST
            else:
                total += -GAP_EXT
            tot["gapExtensions"] += 1
            if d == 1:
                i -= 1
            else:
                j -= 1
    while i > 0:
        _dot(al, seq0, seq1, i0 + i, j0 + j, 1)
        i -= 1
        tot["gapExtensions"] += 1
        c = TYPE_GAP_2
        total += -GAP_EXT
    while j > 0:
        _dot(al, seq0, seq1, i0 + i, j0 + j, 2)
        j -= 1
        tot["gapExtensions"] += 1
        c = TYPE_GAP_1
        total += -GAP_EXT
    if type_s == TYPE_MATCH and c != TYPE_MATCH:
        total -= GAP_OPEN
    tot["score"] += total
    return total


def stage5(seq0, seq1, crosspoints):
    """stage5(), sw_stage5.cpp:322-485: the Alignment of the path through `crosspoints`"""
    al = Alignment()
    tot = dict(score=0, matches=0, mismatches=0, gapOpen=0, gapExtensions=0)
    d0, d1 = seq0.data(), seq1.data()
    if max((max(abs(q[1] - p[1]), abs(q[2] - p[2])) for p, q in zip(crosspoints, crosspoints[1:])
            if q[1] != p[1] and q[2] != p[2]), default=0) > 8192:
        raise ValueError("stage5: a partition is larger than the reference's W_MAX; run stage 4 first")
    m0 = crosspoints[0]
    for m1 in crosspoints[1:]:
        _sw(al, seq0, seq1, d0, d1, m0[1], m0[2], m1[1], m1[2], m0[0], m1[0], tot)
        m0 = m1
    start, end = crosspoints[0], crosspoints[-1]
    if len(crosspoints) != 1:
        al.start = [seq0.absolute_pos(start[1] + 1), seq1.absolute_pos(start[2] + 1)]
        al.end = [seq0.absolute_pos(end[1]), seq1.absolute_pos(end[2])]
    correct = end[3] - start[3]
    if correct != tot["score"]:
        raise RuntimeError("stage5: Wrong Alignment Score: %d != %d" % (tot["score"], correct))
    al.raw_score, al.matches, al.mismatches = tot["score"], tot["matches"], tot["mismatches"]
    al.gap_open, al.gap_extensions = tot["gapOpen"], tot["gapExtensions"]
    al.finalize()
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
    i0, j0, i1, j1 = al.start[0], al.start[1], al.end[0], al.end[1]
    gaps0, gaps1 = [list(g) for g in al.gaps[0]], [list(g) for g in al.gaps[1]]
    COLS = 60
    dir_i = 1 if i1 > i0 else -1
    dir_j = 1 if j1 > j0 else -1
    c0 = 0 if dir_i > 0 else len(gaps0) - 1
    c1 = 0 if dir_j > 0 else len(gaps1) - 1
    end_gap = [-1, -1]
    cur0 = list(gaps0[c0]) if 0 <= c0 < len(gaps0) else list(end_gap)
    cur1 = list(gaps1[c1]) if 0 <= c1 < len(gaps1) else list(end_gap)
    i, j = i0, j0
    end_i = end_j = False
    score = gap_openings = gap_extentions = matches = mismatches = 0
    qgap = sgap = 0
    if i0 == -1 and j0 == -1 and i1 == -1 and j1 == -1:
        end_i = end_j = True
        out.append("There was no alignment produced!\n\n")
    while not end_i or not end_j:
        query, qp = [], i
        k = 0
        while k < COLS and not end_i:
            if cur0[0] == i + (0 if dir_i > 0 else 1):
                query.append(45)
                cur0[1] -= 1
                if cur0[1] == 0:
                    c0 += dir_i
                    cur0 = list(gaps0[c0]) if 0 <= c0 < len(gaps0) else list(end_gap)
            else:
                query.append(int(d0[i - 1]))
                if i == i1:
                    end_i = True
                    break
                i += dir_i
            k += 1
        subject, sp = [], j
        k = 0
        while k < COLS and not end_j:
            if cur1[0] == j + (0 if dir_j > 0 else 1):
                subject.append(45)
                cur1[1] -= 1
                if cur1[1] == 0:
                    c1 += dir_j
                    cur1 = list(gaps1[c1]) if 0 <= c1 < len(gaps1) else list(end_gap)
            else:
                subject.append(int(d1[j - 1]))
                if j == j1:
                    end_j = True
                    break
                j += dir_j
            k += 1
        if len(subject) < len(query):
            subject += [45] * (len(query) - len(subject))
        else:
            query += [45] * (len(subject) - len(query))
        qs, ss = bytes(query).decode("latin-1"), bytes(subject).decode("latin-1")
        out.append("Query: %8d %s %8d\n" % (qp, qs, i))
        out.append("                ")
        temp = 0
        marks = []
        for q, s in zip(query, subject):
            marks.append("|" if q == s else " ")
            if q == 45:
                if qgap:
                    temp += -GAP_EXT
                else:
                    temp += -GAP_OPEN - GAP_EXT
                    gap_openings += 1
                gap_extentions += 1
                qgap, sgap = 1, 0
            elif s == 45:
                if sgap:
                    temp += -GAP_EXT
                else:
                    temp += -GAP_OPEN - GAP_EXT
                    gap_openings += 1
                gap_extentions += 1
                qgap, sgap = 0, 1
            else:
                if q == s:
                    temp += MATCH
                    matches += 1
                else:
                    temp += MISMATCH
                    mismatches += 1
                qgap = sgap = 0
        score += temp
        out.append("".join(marks))
        out.append(" [%d/%d]\n" % (temp, score))
        out.append("Sbjct: %8d %s %8d\n" % (sp, ss, j))
        out.append("\n\n")
    if score != al.raw_score:
        raise RuntimeError("Stage6 error: Alignment score is different (%d != %d)" % (score, al.raw_score))
    out.append("Summary:\n\n")
    out.append("Total Score:    %10d\n" % score)
    out.append("Matches:        %10d (+%d)\n" % (matches, MATCH))
    out.append("Mismatches:     %10d (%d)\n" % (mismatches, MISMATCH))
    out.append("Gap Openings:   %10d (%d)\n" % (gap_openings, -GAP_OPEN))
    out.append("Gap Extentions: %10d (%d)\n" % (gap_extentions, -GAP_EXT))
    return "".join(out).encode("latin-1")

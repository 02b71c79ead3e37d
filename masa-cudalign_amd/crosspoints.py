"""Crosspoints and their files: the hand-over format between the traceback stages.

Counterpart of MASA-Core's M/common/Crosspoint.hpp (crosspoint_t, reverse :68-81) and M/common/CrosspointsFile.cpp
(text format :99-150, loadCrosspoints :42-71, reverse :163-167).  A crosspoint is (type, i, j, score): a cell of the DP
matrix the optimal alignment passes through, `type` telling how it arrives there (0 aligned, 1 inside a gap in S0 --
moving along S1 --, 2 inside a gap in S1), `score` what the file's stage defines it to be.

    START
    type,i,j,score
    ...
    END

Files are written through `<name>.tmp` and renamed when closed (CrosspointsFile::open/close, :117-138)."""
import os

TYPE_MATCH, TYPE_GAP_1, TYPE_GAP_2 = 0, 1, 2


class Crosspoint:
    __slots__ = ("i", "j", "type", "score")

    def __init__(self, i=-1, j=-1, score=0, type=TYPE_MATCH):
        self.i, self.j, self.score, self.type = int(i), int(j), int(score), int(type)

    def copy(self):
        return Crosspoint(self.i, self.j, self.score, self.type)

    def reverse(self, seq0_len, seq1_len):
        """Crosspoint.hpp:68-81: the same cell seen from the reversed, transposed matrix
        (i' = |S1| - j, j' = |S0| - i; the two gap types swap)"""
        t = {TYPE_GAP_1: TYPE_GAP_2, TYPE_GAP_2: TYPE_GAP_1}.get(self.type, self.type)
        return Crosspoint(seq1_len - self.j, seq0_len - self.i, self.score, t)

    def astuple(self):
        """(type, i, j, score): the order of the file and of mi355sw_crosspoint"""
        return (self.type, self.i, self.j, self.score)

    def __eq__(self, o):
        return isinstance(o, Crosspoint) and self.astuple() == o.astuple()

    def __repr__(self):
        return "Crosspoint(type=%d, i=%d, j=%d, score=%d)" % self.astuple()


class CrosspointsFile(list):
    def __init__(self, filename):
        list.__init__(self)
        self.filename = filename
        self.tmp = filename + ".tmp"
        self.file = None

    def load(self):
        """loadCrosspoints (:42-71); a missing file gives an empty list"""
        del self[:]
        if not os.path.exists(self.filename):
            return self
        started = False
        for line in open(self.filename):
            if line == "END\n":
                break
            if started:
                t, i, j, s = (int(x) for x in line.strip().split(","))
                self.append(Crosspoint(i, j, s, t))
            if line == "START\n":
                started = True
                del self[:]
        return self

    def open(self):
        """setAutoSave + open (:95-98, :117-125): every write goes to disk at once"""
        os.makedirs(os.path.dirname(self.filename) or ".", exist_ok=True)
        self.file = open(self.tmp, "w")
        self.file.write("START\n")
        return self

    def write(self, c):
        if self.file is None:
            raise RuntimeError("crosspoints file not opened: " + self.filename)
        self.file.write("%d,%d,%d,%d\n" % c.astuple())
        self.file.flush()
        self.append(c.copy())

    def close(self):
        if self.file is not None:
            self.file.write("END\n")
            self.file.close()
            self.file = None
            os.replace(self.tmp, self.filename)

    def save(self):
        """(:152-160): the whole list at once"""
        points = list(self)
        self.open()
        for c in points:
            self.file.write("%d,%d,%d,%d\n" % c.astuple())
        self.close()

    def reverse_all(self, seq0_len, seq1_len):
        """CrosspointsFile::reverse (:163-167): every point reversed, and the order of the list too"""
        pts = [c.reverse(seq0_len, seq1_len) for c in self]
        pts.reverse()
        self[:] = pts

    def tuples(self):
        return [c.astuple() for c in self]


def save_array(filename, points):
    """CrosspointsFile::save (:152-160) for an (N, 4) array of (type, i, j, score): the same bytes as CrosspointsFile.save
    writes for the same points, formatted by the library (mi355sw_crosspoints_text: crosspoint_04 of BASELINE config 3 is
    4.6 M lines -- 2-4 s of "%d,%d,%d,%d" in Python, a tenth of a second there)"""
    from .engine import crosspoints_text
    tmp = filename + ".tmp"
    with open(tmp, "wb") as f:
        f.write(crosspoints_text(points))
    os.replace(tmp, filename)


def crosspoint_file(work, stage, ident=0, deep=-1):
    """Job::getCrosspointFile, M/common/Job.cpp:192-200"""
    d = os.path.join(work, "crosspoints")
    os.makedirs(d, exist_ok=True)
    if deep <= -1:
        return os.path.join(d, "crosspoint_%02d.%02d" % (stage, ident))
    return os.path.join(d, "crosspoint_%02d.%02d.r%02d" % (stage, ident, deep))

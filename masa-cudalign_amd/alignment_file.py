"""`alignment.NN.bin`: the binary alignment file stage 5 leaves for stage 6 and for MASA's viewers -- MASA-Core's
M/common/biology/AlignmentBinaryFile.cpp, written and read here so that a natively traced alignment can be handed to
MASA-Core's tools and the other way round.

Layout ("CGFF", version 0.1; integers big-endian, strings length-prefixed, field lists ended by a 0 byte):

    "CGFF" major minor
    int4 #sequences, per sequence: [1 description] [2 type: 1 = DNA] [3 size] 0
    alignment parameters:  [1 method: 2 = local] [2 score system 1: match, mismatch] [3 penalty system 2: gap open, gap ext]
                           [4 #sequences, per sequence: index, flags (1 reverse, 2 complement, 4 clear-n), trim start, trim end] 0
    int4 #results (1), result: [1 raw score] [4 matches, mismatches, gap openings, gap extensions]
                               [5 per sequence: start, end, #gaps, per gap: (position - previous position, length) as
                                  7-bit groups, most significant first, high bit = "more follows"] 0

A gap run that a partition border of stage 5 cuts in two is stored as two entries with the same position; the reference
sorts its gap list with std::sort on the position alone (Alignment.cpp:112-124), so the order of such twins is whatever
the C++ library makes of it.  Readers, MASA-Core's included, treat them alike; `canonical()` orders them for comparisons.

The parameters are MASA-Core's defaults (M/libmasa/libmasa.cpp:772-777: method LOCAL whatever the edges, penalties stored
NEGATED: -3, -2)."""
import struct

MAGIC = b"CGFF"
VERSION = (0, 1)
SEQUENCE_TYPE_DNA = 1
ALIGNMENT_METHOD_LOCAL = 2
SCORE_MATCH_MISMATCH = 1
PENALTY_AFFINE_GAP = 2
FLAG_REVERSE, FLAG_COMPLEMENT, FLAG_CLEAR_N = 1, 2, 4


def _i4(v):
    return struct.pack(">i", int(v))


def _str(s):
    b = s.encode("latin-1")
    return _i4(len(b)) + b


def _u4c(v):
    """fwrite_uint4_compressed (:507-536)"""
    v = int(v) & 0xFFFFFFFF
    groups = [(v >> 28) & 0xF, (v >> 21) & 0x7F, (v >> 14) & 0x7F, (v >> 7) & 0x7F, v & 0x7F]
    k = 0
    while k < 4 and groups[k] == 0:
        k += 1
    return bytes([0x80 | g for g in groups[k:4]] + [groups[4]])


def _gaps_bytes(gaps):
    """the gap list of one sequence as the file holds it: (position - previous position, length) per gap, both
    fwrite_uint4_compressed -- _u4c over the whole list at once (a chromosome pair has 10^5..10^6 gaps)"""
    import numpy as np
    if len(gaps) == 0:
        return b""
    g = np.asarray(gaps, dtype=np.int64).reshape(-1, 2)
    v = np.empty((len(g), 2), dtype=np.int64)
    v[:, 0] = np.diff(g[:, 0], prepend=0)
    v[:, 1] = g[:, 1]
    v = v.reshape(-1) & 0xFFFFFFFF
    groups = np.stack([(v >> 28) & 0xF, (v >> 21) & 0x7F, (v >> 14) & 0x7F, (v >> 7) & 0x7F, v & 0x7F], axis=1).astype(np.uint8)
    nz = groups[:, :4] != 0
    first = np.where(nz.any(axis=1), nz.argmax(axis=1), 4)            # leading zero groups are not written
    keep = np.arange(5)[None, :] >= first[:, None]
    keep[:, 4] = True
    groups[:, :4] |= 0x80
    return groups[keep].tobytes()


def _flags(seq):
    """fwrite_flags (:402-415); the trim range is the normalised one (open ends filled in: 1 .. size)"""
    mod = seq.modifiers
    f = (FLAG_CLEAR_N if mod.clear_n else 0) | (FLAG_COMPLEMENT if mod.complement else 0) | (FLAG_REVERSE if mod.reverse else 0)
    return _i4(f) + _i4(seq.offset0) + _i4(seq.offset1)


def dumps(alignment, seq0, seq1, match=1, mismatch=-3, gap_open=3, gap_ext=2):
    """AlignmentBinaryFile::write (:68-88): alignment = stage56.Alignment, seq0 / seq1 = fasta.Sequence"""
    seqs = (seq0, seq1)
    out = [MAGIC, bytes(VERSION), _i4(len(seqs))]
    for s in seqs:
        out += [b"\x01", _str(s.description), b"\x02", bytes([SEQUENCE_TYPE_DNA]), b"\x03", _i4(s.original_size), b"\x00"]
    out += [b"\x01", bytes([ALIGNMENT_METHOD_LOCAL]),
            b"\x02", bytes([SCORE_MATCH_MISMATCH]), _i4(match), _i4(mismatch),
            b"\x03", bytes([PENALTY_AFFINE_GAP]), _i4(-gap_open), _i4(-gap_ext),
            b"\x04", _i4(len(seqs))]
    for k, s in enumerate(seqs):
        out += [_i4(k), _flags(s)]
    out += [b"\x00", _i4(1),
            b"\x01", _i4(alignment.raw_score),
            b"\x04", _i4(alignment.matches), _i4(alignment.mismatches), _i4(alignment.gap_open), _i4(alignment.gap_extensions),
            b"\x05"]
    for k in range(2):
        out += [_i4(alignment.start[k]), _i4(alignment.end[k]), _i4(len(alignment.gaps[k]))]
        out.append(_gaps_bytes(alignment.gaps[k]))
    out.append(b"\x00")
    return b"".join(out)


class _Reader:
    def __init__(self, data):
        self.d, self.p = data, 0

    def take(self, n):
        b = self.d[self.p:self.p + n]
        if len(b) != n:
            raise ValueError("alignment file ends inside a field at byte %d" % self.p)
        self.p += n
        return b

    def i1(self):
        return self.take(1)[0]

    def i4(self):
        return struct.unpack(">i", self.take(4))[0]

    def s(self):
        n = self.i4()
        if n > 1000:
            raise ValueError("string of %d bytes in an alignment file" % n)
        return self.take(n).decode("latin-1")

    def u4c(self):
        b = self.i1()
        v = b & 0x7F
        while b >= 128:
            b = self.i1()
            v = (v << 7) | (b & 0x7F)
        return v


def loads(data):
    """AlignmentBinaryFile::read (:90-102) into plain dictionaries"""
    r = _Reader(data)
    if r.take(4) != MAGIC:
        raise ValueError("not an alignment file (CGFF header missing)")
    major, minor = r.i1(), r.i1()
    if major > VERSION[0]:
        raise ValueError("alignment file version %d.%d not supported" % (major, minor))
    seqs = []
    for _ in range(r.i4()):
        info = {}
        while True:
            f = r.i1()
            if f == 0:
                break
            if f == 1:
                info["description"] = r.s()
            elif f == 2:
                info["type"] = r.i1()
            elif f == 3:
                info["size"] = r.i4()
            elif f == 4:
                info["hash"] = r.s()
            elif f in (5, 6):
                r.take(r.i4())
            else:
                raise ValueError("unknown sequence field %d" % f)
        seqs.append(info)
    params = {"sequences": []}
    while True:
        f = r.i1()
        if f == 0:
            break
        if f == 1:
            params["method"] = r.i1()
        elif f == 2:
            params["score_system"] = r.i1()
            params["match"], params["mismatch"] = r.i4(), r.i4()
        elif f == 3:
            params["penalty_system"] = r.i1()
            params["gap_open"] = r.i4() if params["penalty_system"] == PENALTY_AFFINE_GAP else 0
            params["gap_ext"] = r.i4()
        elif f == 4:
            for _ in range(r.i4()):
                idx, flags, t0, t1 = r.i4(), r.i4(), r.i4(), r.i4()
                params["sequences"].append({"index": idx, "reverse": bool(flags & FLAG_REVERSE),
                                            "complement": bool(flags & FLAG_COMPLEMENT), "clear_n": bool(flags & FLAG_CLEAR_N),
                                            "trim_start": t0, "trim_end": t1})
        else:
            raise ValueError("unknown parameter field %d" % f)
    if r.i4() != 1:
        raise ValueError("more than one result in an alignment file")
    res = {"start": [None, None], "end": [None, None], "gaps": [[], []]}
    while True:
        f = r.i1()
        if f == 0:
            break
        if f == 1:
            res["raw_score"] = r.i4()
        elif f == 4:
            res["matches"], res["mismatches"], res["gap_open"], res["gap_extensions"] = r.i4(), r.i4(), r.i4(), r.i4()
        elif f == 5:
            for k in range(len(params["sequences"])):
                res["start"][k], res["end"][k] = r.i4(), r.i4()
                last = 0
                for _ in range(r.i4()):
                    last += r.u4c()
                    res["gaps"][k].append([last, r.u4c()])
        elif f == 6:
            h, w = r.i4(), r.i4()
            r.take(4 * h * w)
        else:
            raise ValueError("unknown result field %d" % f)
    return {"sequences": seqs, "params": params, "result": res}


def canonical(parsed):
    """`loads()` output with gap entries of equal position in a fixed order (by length)"""
    out = dict(parsed)
    res = dict(parsed["result"])
    res["gaps"] = [sorted([list(g) for g in gaps]) for gaps in parsed["result"]["gaps"]]
    out["result"] = res
    return out

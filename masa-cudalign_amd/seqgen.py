"""Deterministic synthetic DNA generator.

splitmix64(seed) stream, base = "ACGT"[x >> 62].  "unrelated" pairs are two
independent streams; "related" pairs derive seq1 from seq0 with per-base
substitutions, geometric indels and one inverted segment.  Pure numpy so the
GPU box regenerates byte-identical inputs from (seed, length, spec).

The reference has no generator (its inputs are NCBI accessions, README.md:82-93);
the byte-level conventions the engine relies on are the reference's FASTA
normalisation: upper-case ASCII, no line breaks (SequenceData.cpp:67-114).
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def splitmix64(seed, count, offset=0):
    """count consecutive outputs of splitmix64 seeded with `seed`, starting at output `offset`."""
    with np.errstate(over="ignore"):
        k = np.arange(offset + 1, offset + count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + k * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def random_dna(seed, length, chunk=1 << 24):
    """Uniform i.i.d. ACGT of `length` bytes (uint8 ASCII)."""
    out = np.empty(length, dtype=np.uint8)
    for off in range(0, length, chunk):
        n = min(chunk, length - off)
        out[off:off + n] = _ACGT[(splitmix64(seed, n, off) >> np.uint64(62)).astype(np.intp)]
    return out


def mutate_dna(seq0, seed, p_sub=0.02, p_indel=0.002, indel_mean=3.0, inversion=0.05):
    """seq1 derived from seq0: substitutions, geometric indels, one inverted segment."""
    n = len(seq0)
    r = splitmix64(seed, n)
    u = (r >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    r2 = splitmix64(seed ^ 0x5DEECE66D, n)
    # substitution: rotate the base by 1..3 so it always differs
    codes = np.searchsorted(_ACGT, seq0)
    codes = np.where(_ACGT[np.clip(codes, 0, 3)] == seq0, codes, 0)
    rot = (r2 & np.uint64(0xFFFF)).astype(np.int64) % 3 + 1
    sub = u < p_sub
    new_codes = np.where(sub, (codes + rot) % 4, codes)
    out = _ACGT[new_codes]
    # indels: geometric length with the given mean
    glen = 1 + np.floor(np.log1p(-((r2 >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.float64)
                                 / float(1 << 24)) / np.log(1.0 - 1.0 / indel_mean)).astype(np.int64)
    is_del = (u >= p_sub) & (u < p_sub + p_indel / 2)
    is_ins = (u >= p_sub + p_indel / 2) & (u < p_sub + p_indel)
    keep = np.ones(n, dtype=bool)
    for pos in np.nonzero(is_del)[0]:
        keep[pos:pos + glen[pos]] = False
    reps = np.where(is_ins, 1 + glen, 1)
    reps = np.where(keep, reps, 0)
    out = np.repeat(out, reps)
    # inserted copies get fresh random bases
    ins_mask = np.ones(len(out), dtype=bool)
    starts = np.cumsum(reps) - reps
    ins_mask[starts[reps > 0]] = False
    n_ins = int(ins_mask.sum())
    if n_ins:
        out[ins_mask] = random_dna(seed ^ 0xABCDEF, n_ins)
    if inversion > 0 and len(out) > 20:
        seg = int(len(out) * inversion)
        start = int(len(out) * 0.6)
        out[start:start + seg] = out[start:start + seg][::-1].copy()
    return np.ascontiguousarray(out)


# seeds fixed per BASELINE.md section 2
SEED0 = 0xC0FFEE00
SEED1 = 0xBADC0DE0
# identity of the generator: recorded results (bench.py's expected best cells, the digests under profiles/) name the pairs they
# belong to by (kind, m, n, cfg) AND this number; whoever changes what a (kind, m, n, cfg) produces raises it
GENERATOR_VERSION = 1


def unrelated_pair(m, n, cfg=0):
    return random_dna(SEED0 + cfg, m), random_dna(SEED1 + cfg, n)


def related_pair(m, n, cfg=0, **kw):
    s0 = random_dna(SEED0 + cfg, m)
    s1 = mutate_dna(random_dna(SEED0 + cfg, max(m, n) + n // 8 + 64), SEED1 + cfg, **kw)
    if len(s1) < n:
        s1 = np.concatenate([s1, random_dna(SEED1 + cfg + 77, n - len(s1))])
    return s0, np.ascontiguousarray(s1[:n])


def write_fasta(path, seq, name="synthetic", width=70):
    with open(path, "wb") as f:
        f.write(b">" + name.encode() + b"\n")
        b = seq.tobytes()
        for i in range(0, len(b), width):
            f.write(b[i:i + width] + b"\n")

"""FASTA loading with MASA-Core's normalisation and sequence modifiers -- the bytes the aligner compares.

Restates M/common/biology/SequenceData.cpp:67-114 (loadFile), SequenceModifiers and the data view of
M/common/biology/Sequence.cpp (:117-159 trim / setBoundaries / getData):

  * the first line is the description, whatever it holds; every other byte except CR, LF and blank is a residue;
  * residues are upper-cased; --complement maps A<->T and C<->G (after upper-casing, other letters unchanged);
    --clear-n turns N/n into lower-case 'n' (so that it no longer equals an upper-case N of the other sequence --
    two cleared sequences still match each other's 'n', as in the reference);
  * --reverse reads the data back to front; --trim=a,b keeps residues a..b (1-based, inclusive; 0 = open end), counted
    on the forward data like the reference's Sequence::setBoundaries.

`load()` returns the uint8 array an aligner gets (what MASA-Core hands to IAligner::setSequences after trimming) and the
description; `original_size` counts the residues in the file (SequenceInfo::getSize)."""
import numpy as np


class SequenceModifiers:
    def __init__(self, clear_n=False, reverse=False, complement=False, trim_start=0, trim_end=0):
        self.clear_n, self.reverse, self.complement = clear_n, reverse, complement
        self.trim_start, self.trim_end = trim_start, trim_end


def _byte_map(mod):
    m = np.arange(256, dtype=np.uint8)
    for c in range(ord("a"), ord("z") + 1):
        m[c] = c - 32                                   # toupper
    if mod.complement:
        for a, b in (("A", "T"), ("T", "A"), ("C", "G"), ("G", "C")):
            m[ord(a)] = m[ord(a.lower())] = ord(b)
    if mod.clear_n:
        m[ord("N")] = m[ord("n")] = ord("n")
    return m


class Sequence:
    """forward data + the view the modifiers select"""

    def __init__(self, description, forward, modifiers):
        self.raw_description = description
        # SequenceInfo::setDescription (M/common/biology/SequenceInfo.cpp:56-67): without the '>' and the line end
        d = description[1:] if description.startswith(">") else description
        nl = d.find("\n")
        self.description = d[:nl] if nl >= 0 else d
        self.forward, self.modifiers = forward, modifiers
        self.original_size = len(forward)
        t0 = modifiers.trim_start if modifiers.trim_start > 0 else 1
        t1 = modifiers.trim_end if modifiers.trim_end > 0 else len(forward)
        self.offset0, self.offset1 = t0, t1              # Sequence::setBoundaries (:117-128)

    def __len__(self):
        return self.offset1 - self.offset0 + 1

    def data(self, reverse=False):
        """Sequence::getData(reverse): the whole forward or reversed data (trimming is applied by the stage drivers
        through getTrimStart/End)"""
        return self.forward[::-1] if (self.modifiers.reverse ^ reverse) else self.forward

    def trimmed(self):
        """the residues a stage-1 partition (trimStart-1 .. trimEnd) covers, in aligner order"""
        d = self.data()
        return np.ascontiguousarray(d[self.offset0 - 1:self.offset1])

    def absolute_pos(self, relative_pos):
        """Sequence::getAbsolutePos (:130-136)"""
        return self.original_size + 1 - relative_pos if self.modifiers.reverse else relative_pos


def parse(raw, modifiers=None):
    """`raw`: the bytes of a FASTA file"""
    mod = modifiers or SequenceModifiers()
    nl = raw.find(b"\n")
    if nl < 0:
        # fgets() took everything (up to 499 bytes) as the description; nothing is left for residues
        description, body = raw[:499], raw[499:]
    else:
        first = raw[:nl + 1]
        description, body = first[:499], raw[len(first[:499]):]   # fgets(line, 500): a longer first line spills over
    a = np.frombuffer(body, dtype=np.uint8)
    keep = (a != 13) & (a != 10) & (a != 32)
    fwd = _byte_map(mod)[a[keep]]
    return Sequence(description.decode("latin-1"), np.ascontiguousarray(fwd), mod)


def load(path, modifiers=None):
    with open(path, "rb") as f:
        return parse(f.read(), modifiers)

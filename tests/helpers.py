"""Shared helpers for the parity tests."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stage1_cases.json")


def load_golden():
    with open(GOLDEN) as f:
        return json.load(f)


def make_pair(pkg, spec):
    """Must stay identical to oracle/make_golden.py:make_pair (the fixtures pin the sha256 of both)."""
    sg = pkg.seqgen
    kind = spec["kind"]
    if kind == "related":
        return sg.related_pair(spec["m"], spec["n"], cfg=spec["cfg"])
    if kind == "unrelated":
        return sg.unrelated_pair(spec["m"], spec["n"], cfg=spec["cfg"])
    if kind == "with_n":
        s0, s1 = sg.related_pair(spec["m"], spec["n"], cfg=spec["cfg"])
        s0, s1 = s0.copy(), s1.copy()
        s0[spec["m"] // 3: spec["m"] // 3 + 50] = ord("N")
        s1[spec["n"] // 3 + 10: spec["n"] // 3 + 70] = ord("N")
        s1[5::97] = ord("R")
        return s0, s1
    if kind == "literal":
        return (np.frombuffer(spec["s0"].encode(), dtype=np.uint8), np.frombuffer(spec["s1"].encode(), dtype=np.uint8))
    raise ValueError(kind)


def digest(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return {"len": int(a.shape[0]), "sha256": hashlib.sha256(a.tobytes()).hexdigest(),
            "head": a[:4].tolist(), "tail": a[-4:].tolist()}


def parse_args(args):
    """Reference CLI flags of a fixture -> stage-1 parameters (libmasa.cpp:1054-1103, sw_stage1.cpp:137-161)."""
    edges = "**"
    out = dict(pruning=True, disk=0, block=(1024, 1024), stage1_only=False)
    for a in args:
        if a.startswith("--edges="):
            edges = a[8:10]
        elif a == "--no-block-pruning":
            out["pruning"] = False
        elif a == "--no-flush":
            out["disk"] = -1
        elif a.startswith("--disk-size=") and out["disk"] != -1:
            v = a[12:]
            mult = {"K": 1024, "M": 1024 ** 2, "G": 1024 ** 3}[v[-1]]
            out["disk"] = int(float(v[:-1]) * mult)
        elif a.startswith("--block="):
            h, w = a[8:].split(",")
            out["block"] = (int(h), int(w))
        elif a == "--stage-1":
            out["stage1_only"] = True
    flag = {"*": 0, "1": 1, "2": 2, "3": 3, "+": 4}
    out["start"], out["end"] = flag[edges[0]], flag[edges[1]]
    return out


def flush_interval(m, n, limit):
    """Job::calculateFlushIntervals, M/common/Job.cpp:231-241 (first interval only)."""
    if limit <= 0:
        return 0
    if limit < n * 8 * 2:
        limit = n * 8 * 2
    return int(m * n * 8 // limit + 1)


def oracle_kwargs(oracle, p, m, n):
    """stage-1 set-up of sw_stage1.cpp:137-161/:318-322/:219-225 expressed as oracle.stage1 arguments."""
    start, end = p["start"], p["end"]
    kw = dict(block_h=p["block"][0], block_w=p["block"][1])
    kw["recurrence"] = oracle.SMITH_WATERMAN if start == 0 else oracle.NEEDLEMAN_WUNSCH
    Z, G = oracle.INIT_WITH_ZEROES, oracle.INIT_WITH_GAPS
    kw["first_row_type"], kw["first_col_type"] = {0: (Z, Z), 1: (Z, G), 2: (G, Z), 3: (Z, Z), 4: (G, G)}[start]
    kw["best_mode"] = {0: oracle.BEST_ANYWHERE, 1: oracle.BEST_LAST_ROW, 2: oracle.BEST_LAST_COL,
                       3: oracle.BEST_LAST_ROW_OR_COL, 4: oracle.BEST_LAST_CELL}[end]
    kw["want_last_row"] = end in (1, 3)
    kw["want_last_col"] = end in (2, 3)
    kw["pruning"] = p["pruning"] and end == 0
    kw["special_row_interval"] = flush_interval(m, n, p["disk"])
    if kw["special_row_interval"]:
        kw["want_last_row"] = True     # SpecialRowsPartition always hands out a last-row writer
    return kw


def oracle_full(oracle, seq0, seq1, edge=0, special_row_interval=8192):
    """the oracle's whole-matrix answer for the larger GPU parity cases -- best cell, last row, last column, special rows every
    `special_row_interval` rows -- on every host core (oracle_stage1_mt: the same cells as the serial schedule, pinned on it by
    tests/test_oracle_golden.py::test_threaded_oracle_equals_the_serial_one).  edge: 0 = local (**), 4 = global (++)."""
    m, n = len(seq0), len(seq1)
    kw = oracle_kwargs(oracle, dict(start=edge, end=edge, pruning=False, disk=-1, block=(1024, 1024)), m, n)
    kw.update(want_last_row=True, want_last_col=True, special_row_interval=special_row_interval, threads=min(64, os.cpu_count() or 1))
    return oracle.stage1(seq0, seq1, **kw)

#!/usr/bin/env python3
"""Generates tests/golden/stage1_cases.json from the REFERENCE's own CPU path (oracle/_ref/ref_driver,
real MASA-Core compiled from /root/reference).  Run in the build container only; the GPU box and the
tests read the committed JSON.  A fixture is data: generator parameters + expected outputs."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
oracle = graft.load_oracle()
sg = pkg.seqgen


def make_pair(spec):
    kind = spec["kind"]
    if kind == "related":
        s0, s1 = sg.related_pair(spec["m"], spec["n"], cfg=spec["cfg"])
    elif kind == "unrelated":
        s0, s1 = sg.unrelated_pair(spec["m"], spec["n"], cfg=spec["cfg"])
    elif kind == "with_n":
        s0, s1 = sg.related_pair(spec["m"], spec["n"], cfg=spec["cfg"])
        s0, s1 = s0.copy(), s1.copy()
        s0[spec["m"] // 3: spec["m"] // 3 + 50] = ord("N")
        s1[spec["n"] // 3 + 10: spec["n"] // 3 + 70] = ord("N")
        s1[5::97] = ord("R")          # IUPAC code only in seq1: never matches
    elif kind == "literal":
        s0 = np.frombuffer(spec["s0"].encode(), dtype=np.uint8)
        s1 = np.frombuffer(spec["s1"].encode(), dtype=np.uint8)
    else:
        raise ValueError(kind)
    return s0, s1


def cells_digest(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return {"len": int(a.shape[0]), "sha256": hashlib.sha256(a.tobytes()).hexdigest(),
            "head": a[:4].tolist(), "tail": a[-4:].tolist()}


CASES = [
    dict(name="sw_related_3000x2700", seq=dict(kind="related", m=3000, n=2700, cfg=1),
         args=["--stage-1", "--no-flush", "--no-block-pruning", "--block=128,128"]),
    dict(name="sw_related_pruned_6000x6000", seq=dict(kind="related", m=6000, n=6000, cfg=2),
         args=["--stage-1", "--no-flush", "--block=256,256"]),
    dict(name="sw_unrelated_ties_20000x17000", seq=dict(kind="unrelated", m=20000, n=17000, cfg=3),
         args=["--stage-1", "--no-flush", "--no-block-pruning"]),
    dict(name="sw_special_rows_20000x9000", seq=dict(kind="related", m=20000, n=9000, cfg=4),
         args=["--stage-1", "--disk-size=200K", "--no-block-pruning", "--block=8192,1000"], special=True),
    dict(name="nw_global_3000x2700", seq=dict(kind="related", m=3000, n=2700, cfg=5),
         args=["--stage-1", "--no-flush", "--edges=++", "--block=500,700"]),
    dict(name="nw_global_ragged_1537x2049", seq=dict(kind="related", m=1537, n=2049, cfg=6),
         args=["--stage-1", "--no-flush", "--edges=++"]),
    dict(name="sw_with_N_4000x4100", seq=dict(kind="with_n", m=4000, n=4100, cfg=7),
         args=["--stage-1", "--no-flush", "--no-block-pruning"]),
    dict(name="sw_tiny_1x1_match", seq=dict(kind="literal", s0="A", s1="A"), args=["--stage-1", "--no-flush", "--no-block-pruning"]),
    dict(name="sw_tiny_1x1_mismatch", seq=dict(kind="literal", s0="A", s1="C"), args=["--stage-1", "--no-flush", "--no-block-pruning"]),
    dict(name="sw_tiny_3x70", seq=dict(kind="literal", s0="ACG", s1="TTACGTTTACGACGACGTTTTTTTTTTACGTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTACGA"),
         args=["--stage-1", "--no-flush", "--no-block-pruning"]),
    dict(name="sw_tiny_70x3", seq=dict(kind="literal", s0="TTACGTTTACGACGACGTTTTTTTTTTACGTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTACGA", s1="ACG"),
         args=["--stage-1", "--no-flush", "--no-block-pruning"]),
    dict(name="sw_ragged_513x65", seq=dict(kind="related", m=513, n=65, cfg=8), args=["--stage-1", "--no-flush", "--no-block-pruning"]),
    dict(name="semiglobal_1to3_2500x2600", seq=dict(kind="related", m=2500, n=2600, cfg=9),
         args=["--stage-1", "--no-flush", "--edges=13", "--block=300,300"]),
    dict(name="full_pipeline_3000x2700", seq=dict(kind="related", m=3000, n=2700, cfg=1),
         args=["--disk-size=200K", "--block=128,128"], full=True),
    # same special-row geometry as the MI355X engine (CUDAlign's 8192-row minimum flush interval,
    # AbstractDiagonalAligner.cpp:35,:466-478): co-optimal tracebacks then coincide byte for byte
    dict(name="full_pipeline_3000x2700_b8192", seq=dict(kind="related", m=3000, n=2700, cfg=1),
         args=["--disk-size=200K", "--block=8192,8192"], full=True),
    dict(name="full_pipeline_20000x9000_b8192", seq=dict(kind="related", m=20000, n=9000, cfg=4),
         args=["--disk-size=200K", "--block=8192,8192"], full=True),
    # block pruning ON and biting in the reference run (143 of 392 blocks of 8192 x 1024 pruned): the engine prunes
    # with its own granularity (64-column slabs of a strip), so special rows differ off the optimal path, yet
    # stages 2-6 must come out identical
    dict(name="full_pipeline_pruned_60000x50000_b8192", seq=dict(kind="related", m=60000, n=50000, cfg=31),
         args=["--disk-size=4M", "--block=8192,1024"], full=True, pruned=True),
    # GLOBAL alignment through all six stages (round 4).  The reference never prunes a global stage 1
    # (sw_stage1.cpp:219-225); the engine does (AbstractBlockPruning.cpp:92-109, per slab) -- its special rows are then
    # lower bounds off every optimal path, and stages 2-6 on top of them must reproduce these files
    dict(name="full_pipeline_global_60000x50000_b8192", seq=dict(kind="related", m=60000, n=50000, cfg=34),
         args=["--edges=++", "--disk-size=4M", "--block=8192,1024"], full=True),
]

CHAIN = dict(name="sw_chain3_9000x9000", seq=dict(kind="related", m=9000, n=9000, cfg=11), parts=3)


def main():
    assert oracle.have_ref(), "build oracle/_ref first (oracle/build_ref.sh)"
    out = {"generator": "oracle/make_golden.py", "reference": "masa-cudalign-4.0.2.1028 MASA-Core CPU path", "cases": []}
    path = os.path.join(ROOT, "tests", "golden", "stage1_cases.json")
    # `--add`: only the cases the committed file does not hold yet are run and appended (everything else stays as it is)
    add_only = "--add" in sys.argv[1:]
    if add_only:
        with open(path) as f:
            out = json.load(f)
    have = set(c["name"] for c in out["cases"])
    for case in CASES:
        if case["name"] in have:
            continue
        s0, s1 = make_pair(case["seq"])
        ref = oracle.run_ref(s0, s1, case["args"])
        rec = {"name": case["name"], "seq": case["seq"], "args": case["args"], "m": len(s0), "n": len(s1),
               "seq0_sha256": hashlib.sha256(s0.tobytes()).hexdigest(),
               "seq1_sha256": hashlib.sha256(s1.tobytes()).hexdigest(),
               "best": list(ref["best"])}
        if case.get("pruned"):
            rec["pruned_blocks"] = ref["pruned_blocks"]
            assert rec["pruned_blocks"][0] > 0
        if case.get("special"):
            rec["special_rows"] = {str(i): cells_digest(a) for (d, i), a in sorted(ref["special_rows"].items())}
            rec["sra_listing"] = ref["sra_listing"]          # directory and file names + sizes as MASA-Core wrote them
            rec["status_txt"] = ref["status_txt"]
            rec["crosspoint_txt"] = ref["crosspoint_txt"]
        if case.get("full"):
            rec["alignment_txt_sha256"] = hashlib.sha256(ref["alignment_txt"]).hexdigest()
            rec["alignment_bin_sha256"] = hashlib.sha256(ref["alignment_bin"]).hexdigest()   # AlignmentBinaryFile ("CGFF")
            rec["alignment_bin_hex"] = ref["alignment_bin"].hex()      # 1-2 KB: gap entries with equal positions come out of
            #                                                            the reference's std::sort in a library-defined order
            rec["crosspoints_2"] = ref.get("crosspoints_2")
            # stage 3 -> stage 4 (Myers-Miller refinement): the input list, and the output file's digest
            rec["crosspoints_3"] = ref.get("crosspoints_3")
            rec["crosspoints_4"] = {"count": len(ref["crosspoints_4"]), "head": ref["crosspoints_4"][:4], "tail": ref["crosspoints_4"][-4:],
                                    "file_sha256": hashlib.sha256(ref["crosspoints_4_txt"]).hexdigest()}
            rec["special_rows"] = {str(i): cells_digest(a) for (d, i), a in sorted(ref["special_rows"].items())}
        out["cases"].append(rec)
        print(case["name"], rec["best"], flush=True)
    if add_only:
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        print("wrote", path)
        return
    # chained column bands: --split=N --part=k with file:// boundary columns (libmasa.cpp:497-535)
    import shutil
    import tempfile
    s0, s1 = make_pair(CHAIN["seq"])
    tmp = tempfile.mkdtemp(prefix="masa_chain_")
    try:
        bests = []
        P = CHAIN["parts"]
        for part in range(1, P + 1):
            args = ["--stage-1", "--no-flush", "--split=%d" % P, "--part=%d" % part]
            if part > 1:
                args.append("--load-column=file://%s/STEP-%d.tmp" % (tmp, part - 1))
            if part < P:
                args.append("--flush-column=file://%s/STEP-%d.tmp" % (tmp, part))
            ref = oracle.run_ref(s0, s1, args, workdir=tmp, timeout=120)
            bests.append(list(ref["best"]))   # work/shared must survive: AlignerPool relays the best score through it
        cols = {}
        for fn in sorted(os.listdir(tmp)):
            if fn.startswith("STEP-"):
                cols[fn] = cells_digest(np.fromfile(os.path.join(tmp, fn), dtype=np.int32).reshape(-1, 2))
        single = oracle.run_ref(s0, s1, ["--stage-1", "--no-flush", "--no-block-pruning"])
        out["chain"] = {"name": CHAIN["name"], "seq": CHAIN["seq"], "parts": CHAIN["parts"], "m": len(s0), "n": len(s1),
                        "band_bests": bests, "boundary_columns": cols, "single_best": list(single["best"])}
        print("chain", bests, single["best"], list(cols))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  BASELINE config C1 (1 000 000 x 1 000 000 unrelated, local SW) once through the REFERENCE's own CPU
path -- the "single large pinned result" of SURVEY.md 8(c) -- written to tests/golden/c1_reference.json.

MASA-Core itself (oracle/_ref/ref_driver: its sources compiled where they lie) runs as a chain of P column bands, one
process per band, exactly as `--split=P --part=k` does (libmasa.cpp:497-535): band k hands its last column to band k+1
through `--flush-column=file://... / --load-column=file://...` (FileCellsWriter / FileCellsReader, plain fopen + fwrite
/ fread), and the best score travels from node to node through AlignerPool's message files (sw_stage1.cpp:421-464).  Here
the "files" are named pipes with a copier in between that keeps what passes: every boundary column of the chain is then
a reference-computed value the engine can be compared with, and the bands run side by side instead of one after another.
Each band also flushes special rows (--disk-size), read back from its own special-rows area.

Build container only (about 15-30 minutes on 8 cores); tests and the GPU box read the committed JSON.
    python oracle/make_golden_c1.py [--size 1000000] [--parts 8] [--out tests/golden/c1_reference.json]
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


HEAD_CELLS = 16385     # digest of the first cells of every boundary column as well (rows 0..16384): what the C restatement of the
#                        oracle can recompute in seconds (tests/test_oracle_golden.py) -- the full columns take the GPU


def cells_digest(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return {"len": int(a.shape[0]), "sha256": hashlib.sha256(a.tobytes()).hexdigest(),
            "head": a[:4].tolist(), "tail": a[-4:].tolist(), "max_h": int(a[:, 0].max()),
            "head_cells": min(HEAD_CELLS, int(a.shape[0])), "head_sha256": hashlib.sha256(a[:HEAD_CELLS].tobytes()).hexdigest()}


def copier(src, dst, keep):
    """what `tee` would do between two named pipes: band k's FileCellsWriter -> band k+1's FileCellsReader, and a copy"""
    with open(src, "rb") as fi, open(dst, "wb") as fo, open(keep, "wb") as fk:
        while True:
            b = fi.read(1 << 16)
            if not b:
                break
            fo.write(b)
            fo.flush()
            fk.write(b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1000000)
    ap.add_argument("--parts", type=int, default=8)
    ap.add_argument("--cfg", type=int, default=1)
    ap.add_argument("--disk-size", default="80M", help="per band: its special rows are 8*(n/parts) bytes each")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "c1_reference.json"))
    args = ap.parse_args()

    pkg = graft.load_package()
    oracle = graft.load_oracle()
    from oracle import binding
    assert oracle.have_ref(), "build oracle/_ref first (oracle/build_ref.sh)"
    m = n = args.size
    P = args.parts
    s0, s1 = pkg.seqgen.unrelated_pair(m, n, cfg=args.cfg)
    tmp = tempfile.mkdtemp(prefix="masa_c1_")
    t0 = time.time()
    try:
        f0, f1 = os.path.join(tmp, "s0.fasta"), os.path.join(tmp, "s1.fasta")
        binding._write_fasta(f0, s0, "s0")
        binding._write_fasta(f1, s1, "s1")
        shared = os.path.join(tmp, "shared")
        os.makedirs(shared)
        threads, procs, logs = [], [], []
        for k in range(1, P):
            for nm in ("out", "in"):
                os.mkfifo(os.path.join(tmp, "col%d.%s" % (k, nm)))
            th = threading.Thread(target=copier, args=(os.path.join(tmp, "col%d.out" % k), os.path.join(tmp, "col%d.in" % k),
                                                       os.path.join(tmp, "col%d.bin" % k)), daemon=True)
            th.start()
            threads.append(th)
        for k in range(1, P + 1):
            work = os.path.join(tmp, "part%d" % k, "work")
            os.makedirs(os.path.dirname(work))
            cmd = [binding.REF_DRIVER, "--work-dir=" + work, "--shared-dir=" + shared, "--stage-1", "--disk-size=" + args.disk_size,
                   "--split=%d" % P, "--part=%d" % k]
            if k > 1:
                cmd.append("--load-column=file://" + os.path.join(tmp, "col%d.in" % (k - 1)))
            if k < P:
                cmd.append("--flush-column=file://" + os.path.join(tmp, "col%d.out" % k))
            cmd += [f0, f1]
            log = open(os.path.join(tmp, "part%d.log" % k), "wb")
            logs.append(log)
            procs.append(subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT, cwd=os.path.dirname(work)))
        bad = 0
        for k, p in enumerate(procs, 1):
            rc = p.wait()
            print("part", k, "rc", rc, "%.0f s" % (time.time() - t0), flush=True)
            bad |= rc
        for th in threads:
            th.join(60)
        for log in logs:
            log.close()
        if bad:
            for k in range(1, P + 1):
                sys.stderr.write(open(os.path.join(tmp, "part%d.log" % k), errors="replace").read()[-2000:])
            raise SystemExit("a band of the reference chain failed")
        lim = [int((n * k) // P) for k in range(P + 1)]          # libmasa.cpp:497-535: band k = columns (lim[k-1], lim[k]]
        rec = {"generator": "oracle/make_golden_c1.py", "reference": "masa-cudalign-4.0.2.1028 MASA-Core CPU path, --split=%d chain of "
               "processes, boundary columns through named pipes" % P,
               "seq": {"kind": "unrelated", "m": m, "n": n, "cfg": args.cfg}, "m": m, "n": n, "parts": P, "band_limits": lim,
               "seq0_sha256": hashlib.sha256(s0.tobytes()).hexdigest(), "seq1_sha256": hashlib.sha256(s1.tobytes()).hexdigest(),
               "seconds": None, "band_bests": [], "boundary_columns": {}, "special_rows": {}}
        rows = {}
        for k in range(1, P + 1):
            ref = binding.read_ref_work(os.path.join(tmp, "part%d" % k, "work"))
            rec["band_bests"].append(list(ref["best"]))
            for (d, i), a in sorted(ref["special_rows"].items()):
                rows.setdefault(i, {})[k] = a
        rec["best"] = rec["band_bests"][-1]                     # the last node's crosspoint carries the chain's best
        for k in range(1, P):
            col = np.fromfile(os.path.join(tmp, "col%d.bin" % k), dtype=np.int32).reshape(-1, 2)
            rec["boundary_columns"][str(lim[k])] = cells_digest(col)       # (H,E) of column lim[k], rows 0..m
        for i in sorted(rows):
            if len(rows[i]) != P:
                continue
            # each band's row holds its own columns plus the cell left of them: per band digests, in band order
            rec["special_rows"][str(i)] = [cells_digest(rows[i][k]) for k in range(1, P + 1)]
        rec["seconds"] = time.time() - t0
        with open(args.out, "w") as f:
            json.dump(rec, f, indent=1)
        print("best", rec["best"], "bands", rec["band_bests"], "columns", list(rec["boundary_columns"]), "rows", list(rec["special_rows"]),
              "%.0f s" % rec["seconds"])
        print("wrote", args.out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()

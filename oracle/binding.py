"""ctypes binding of sw_oracle.c + runner for oracle/_ref/ref_driver (test infrastructure)."""
import ctypes as C
import os
import shutil
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libsworacle.so")
REF_DRIVER = os.path.join(HERE, "_ref", "ref_driver")

INF = 999999999
NEEDLEMAN_WUNSCH, SMITH_WATERMAN = 0, 1
INIT_WITH_ZEROES, INIT_WITH_GAPS, INIT_WITH_CUSTOM_DATA, INIT_WITH_GAPS_OPENED = 0, 1, 2, 3
BEST_NOWHERE, BEST_ANYWHERE, BEST_LAST_CELL, BEST_LAST_ROW, BEST_LAST_COL, BEST_LAST_ROW_OR_COL = 0, 1, 2, 3, 4, 5

CELL = np.dtype([("h", "<i4"), ("f", "<i4")])


class OcScore(C.Structure):
    _fields_ = [("i", C.c_int), ("j", C.c_int), ("score", C.c_int)]


class OcParams(C.Structure):
    _fields_ = [
        ("seq0", C.c_void_p), ("m", C.c_int),
        ("seq1", C.c_void_p), ("n", C.c_int),
        ("recurrence", C.c_int),
        ("first_row_type", C.c_int), ("first_col_type", C.c_int),
        ("row_start_offset", C.c_int), ("col_start_offset", C.c_int),
        ("custom_first_row", C.c_void_p), ("custom_first_col", C.c_void_p),
        ("block_h", C.c_int), ("block_w", C.c_int),
        ("special_row_interval", C.c_int),
        ("want_last_row", C.c_int), ("want_last_col", C.c_int),
        ("pruning", C.c_int), ("max_i", C.c_int), ("max_j", C.c_int),
        ("best_mode", C.c_int),
    ]


class OcResult(C.Structure):
    _fields_ = [
        ("best", OcScore),
        ("blocks_total", C.c_longlong), ("blocks_pruned", C.c_longlong),
        ("n_special_rows", C.c_int),
        ("special_row_ids", C.POINTER(C.c_int)),
        ("special_rows", C.c_void_p),
        ("last_row", C.c_void_p),
        ("last_col", C.c_void_p),
        ("block_scores", C.c_void_p), ("grid_w", C.c_int), ("grid_h", C.c_int),
    ]


_lib = None


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < max(os.path.getmtime(os.path.join(HERE, f)) for f in ("sw_oracle.c", "stage4_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "_build/libsworacle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.oracle_stage1.argtypes = [C.POINTER(OcParams), C.POINTER(OcResult)]
        _lib.oracle_stage1.restype = C.c_int
        _lib.oracle_stage1_mt.argtypes = [C.POINTER(OcParams), C.POINTER(OcResult), C.c_int]
        _lib.oracle_stage1_mt.restype = C.c_int
        _lib.oracle_free_result.argtypes = [C.POINTER(OcResult)]
        _lib.oracle_process_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.oracle_process_block.restype = OcScore
        _lib.oracle_initial_cells.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int]
        _lib.oracle_match_column.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                             C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _lib.oracle_match_column.restype = C.c_int
    return _lib


def _u8(a):
    a = np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a,
                             dtype=np.uint8)
    return a


def stage4(seq0, seq1, crosspoints, max_size=16):
    """MASA-Core stage 4 (stage4_oracle.c): refine [(type, i, j, score), ...] down to partitions <= max_size"""
    s0, s1 = _u8(seq0), _u8(seq1)
    cp = np.ascontiguousarray(crosspoints, dtype=np.int32).reshape(-1, 4)
    out = C.c_void_p()
    steps = C.c_int()
    fn = lib().oc_stage4
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    fn.restype = C.c_int
    n = fn(s0.ctypes.data, s1.ctypes.data, cp.ctypes.data, len(cp), max_size, C.byref(out), C.byref(steps))
    if n < 0:
        raise RuntimeError("oc_stage4 failed: %d" % n)
    buf = (C.c_int32 * (n * 4)).from_address(out.value)
    res = np.frombuffer(buf, dtype=np.int32).reshape(n, 4).copy()
    lib().oc_stage4_free.argtypes = [C.c_void_p]
    lib().oc_stage4_free(out)
    return [tuple(int(x) for x in r) for r in res], steps.value


def process_block(seq0, seq1, row, col, i0, j0, i1, j1, recurrence=SMITH_WATERMAN):
    """In-place CPUBlockProcessor::processBlock restatement. row/col: int32 arrays [k,2]."""
    s0, s1 = _u8(seq0), _u8(seq1)
    assert row.dtype == np.int32 and col.dtype == np.int32 and row.flags.c_contiguous and col.flags.c_contiguous
    r = lib().oracle_process_block(s0.ctypes.data, s1.ctypes.data, row.ctypes.data, col.ctypes.data,
                                   i0, j0, i1, j1, recurrence)
    return (r.i, r.j, r.score)


def initial_cells(init_type, position, length):
    buf = np.empty((length, 2), dtype=np.int32)
    lib().oracle_initial_cells(init_type, position, buf.ctypes.data, length)
    return buf


def match_column(buffer, base, goal):
    k, s, t = C.c_int(), C.c_int(), C.c_int()
    buffer = np.ascontiguousarray(buffer, dtype=np.int32)
    base = np.ascontiguousarray(base, dtype=np.int32)
    rc = lib().oracle_match_column(buffer.ctypes.data, base.ctypes.data, len(buffer), goal,
                                   C.byref(k), C.byref(s), C.byref(t))
    return rc, k.value, s.value, t.value


def stage1(seq0, seq1, recurrence=SMITH_WATERMAN, first_row_type=INIT_WITH_ZEROES,
           first_col_type=INIT_WITH_ZEROES, row_start_offset=0, col_start_offset=0,
           custom_first_row=None, custom_first_col=None, block_h=1024, block_w=1024,
           special_row_interval=0, want_last_row=False, want_last_col=False, pruning=False,
           max_i=0, max_j=0, best_mode=BEST_ANYWHERE, threads=0):
    """Run the restated Stage-1 pass.  Returns a dict (numpy copies, nothing borrowed)."""
    s0, s1 = _u8(seq0), _u8(seq1)
    p = OcParams()
    p.seq0, p.m, p.seq1, p.n = s0.ctypes.data, len(s0), s1.ctypes.data, len(s1)
    p.recurrence = recurrence
    p.first_row_type, p.first_col_type = first_row_type, first_col_type
    p.row_start_offset, p.col_start_offset = row_start_offset, col_start_offset
    keep = []
    if custom_first_row is not None:
        a = np.ascontiguousarray(custom_first_row, dtype=np.int32); keep.append(a)
        assert a.shape == (len(s1) + 1, 2)
        p.custom_first_row = a.ctypes.data
    if custom_first_col is not None:
        a = np.ascontiguousarray(custom_first_col, dtype=np.int32); keep.append(a)
        assert a.shape == (len(s0) + 1, 2)
        p.custom_first_col = a.ctypes.data
    p.block_h, p.block_w = block_h, block_w
    p.special_row_interval = special_row_interval
    p.want_last_row, p.want_last_col = int(want_last_row), int(want_last_col)
    p.pruning, p.max_i, p.max_j = int(pruning), max_i, max_j
    p.best_mode = best_mode
    r = OcResult()
    if threads and threads > 0:
        rc = lib().oracle_stage1_mt(C.byref(p), C.byref(r), threads)
    else:
        rc = lib().oracle_stage1(C.byref(p), C.byref(r))
    if rc != 0:
        raise RuntimeError("oracle_stage1 failed: %d" % rc)
    n, m = len(s1), len(s0)
    out = {
        "best": (r.best.i, r.best.j, r.best.score),
        "blocks_total": r.blocks_total, "blocks_pruned": r.blocks_pruned,
        "special_row_ids": [r.special_row_ids[k] for k in range(r.n_special_rows)],
        "special_rows": None, "last_row": None, "last_col": None,
    }
    if r.n_special_rows:
        buf = (C.c_int32 * (r.n_special_rows * (n + 1) * 2)).from_address(r.special_rows)
        out["special_rows"] = np.frombuffer(buf, dtype=np.int32).reshape(r.n_special_rows, n + 1, 2).copy()
    if r.last_row:
        buf = (C.c_int32 * ((n + 1) * 2)).from_address(r.last_row)
        out["last_row"] = np.frombuffer(buf, dtype=np.int32).reshape(n + 1, 2).copy()
    if r.last_col:
        buf = (C.c_int32 * ((m + 1) * 2)).from_address(r.last_col)
        out["last_col"] = np.frombuffer(buf, dtype=np.int32).reshape(m + 1, 2).copy()
    if r.block_scores:      # {(bx, by): (i, j, score)} 0-based cells, as dispatchScore(score, bx, by) receives them
        buf = (C.c_int32 * (r.grid_w * r.grid_h * 3)).from_address(r.block_scores)
        a = np.frombuffer(buf, dtype=np.int32).reshape(r.grid_w, r.grid_h, 3)
        out["block_scores"] = {(bx, by): tuple(int(x) for x in a[bx, by]) for bx in range(r.grid_w) for by in range(r.grid_h)}
        out["grid"] = (r.grid_h, r.grid_w)
    lib().oracle_free_result(C.byref(r))
    return out


# --------------------------------------------------------------------------- #
# oracle/_ref : the reference's own MASA-Core CPU path (only where it was built)
# --------------------------------------------------------------------------- #
def have_ref():
    return os.path.exists(REF_DRIVER)


def _write_fasta(path, seq, name):
    b = _u8(seq).tobytes()
    with open(path, "wb") as f:
        f.write(b">" + name.encode() + b"\n")
        for i in range(0, len(b), 70):
            f.write(b[i:i + 70] + b"\n")


def run_ref(seq0, seq1, args=(), workdir=None, keep=False, timeout=300):
    """Run oracle/_ref/ref_driver on the pair; returns dict with best + special rows read back
    from the reference's own on-disk formats."""
    assert have_ref(), "oracle/_ref/ref_driver not built (needs /root/reference)"
    tmp = workdir or tempfile.mkdtemp(prefix="masa_ref_")
    try:
        f0, f1 = os.path.join(tmp, "s0.fasta"), os.path.join(tmp, "s1.fasta")
        _write_fasta(f0, seq0, "s0")
        _write_fasta(f1, seq1, "s1")
        work = os.path.join(tmp, "work")
        cmd = [REF_DRIVER, "--work-dir=" + work] + list(args) + [f0, f1]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, cwd=tmp)
        if p.returncode != 0:
            raise RuntimeError("ref_driver failed:\n" + p.stdout.decode(errors="replace")[-4000:])
        return read_ref_work(work, log=p.stdout.decode(errors="replace"))
    finally:
        if not keep and workdir is None:
            shutil.rmtree(tmp, ignore_errors=True)


def read_ref_work(work, log=""):
    out = {"log": log, "best": None, "special_rows": {}, "sra_dirs": []}
    cp = os.path.join(work, "crosspoints", "crosspoint_01.00")
    if os.path.exists(cp):
        out["crosspoint_txt"] = open(cp).read()
        lines = open(cp).read().split()
        # START / type,i,j,score / END   (CrosspointsFile.cpp:99-150)
        t, i, j, s = [int(x) for x in lines[1].split(",")]
        out["best"] = (i, j, s)
    sra = os.path.join(work, "special_rows", "stage.01.00")
    if os.path.isdir(sra):
        out["sra_listing"] = {}
        for d in sorted(os.listdir(sra)):
            full = os.path.join(sra, d)
            out["sra_dirs"].append(d)
            out["sra_listing"][d] = {fn: os.path.getsize(os.path.join(full, fn)) for fn in sorted(os.listdir(full))}
            for fn in sorted(os.listdir(full)):
                if len(fn) == 8 and all(c in "0123456789ABCDEF" for c in fn):
                    out["special_rows"][(d, int(fn, 16))] = \
                        np.fromfile(os.path.join(full, fn), dtype=np.int32).reshape(-1, 2)
    # per-stage statistics files (Pruned Blocks of stage 1; the product adapter's own section)
    out["statistics"] = {}
    for fn in sorted(os.listdir(work)) if os.path.isdir(work) else []:
        if fn.startswith("statistics_"):
            txt = open(os.path.join(work, fn), errors="replace").read()
            out["statistics"][fn] = txt
            for ln in txt.splitlines():
                if ln.startswith("Pruned Blocks:") and fn == "statistics_01.00":
                    a, b = ln.split(":")[1].split("/")
                    out["pruned_blocks"] = [int(a), int(b)]
    if os.path.exists(os.path.join(work, "status")):
        out["status_txt"] = open(os.path.join(work, "status")).read()
    for fn, key in (("alignment.00.txt", "alignment_txt"), ("alignment.00.bin", "alignment_bin")):
        pth = os.path.join(work, fn)
        if os.path.exists(pth):
            out[key] = open(pth, "rb").read()
    pth = os.path.join(work, "pruning_dump.txt")         # --dump-blocks: BlocksFile.cpp:44-64 (binary despite its name)
    if os.path.exists(pth):
        raw = np.fromfile(pth, dtype=np.int32)
        gh, gw = int(raw[0]), int(raw[1])
        grid = np.full(gh * gw, -2 ** 31, dtype=np.int64)
        grid[:len(raw) - 2] = raw[2:2 + gh * gw]      # blocks never written stay holes (the file is sparse up to the last one)
        out["blocks_file"] = open(pth, "rb").read()
        out["blocks"] = grid.reshape(gh, gw)
    for st in range(2, 5):
        pth = os.path.join(work, "crosspoints", "crosspoint_%02d.00" % st)
        if os.path.exists(pth):
            out["crosspoints_%d_txt" % st] = open(pth, "rb").read()
            pts = []
            for ln in open(pth).read().split():
                if "," in ln:
                    pts.append(tuple(int(x) for x in ln.split(",")))
            out["crosspoints_%d" % st] = pts
    return out

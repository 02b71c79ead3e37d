"""ctypes binding of the C ABI (include/mi355sw.h) + Python mirror of the IAligner surface.

Method names follow MASA-Core's IAligner (M/libmasa/IAligner.hpp:159-377) so the parity tests read
like a MASA extension would be driven: getCapabilities / setSequences / alignPartition / ...
There is no CPU fallback: if libmi355sw.so is missing or no gfx950 GPU is present, construction
raises AlignerError.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# MI355SW_LIB: alternative build of the same library (kernel A/B experiments, tools/gpu_perf.py)
LIB_PATH = os.environ.get("MI355SW_LIB") or os.path.join(HERE, "libmi355sw.so")
INCLUDE_PATH = os.path.join(os.path.dirname(HERE), "include", "mi355sw.h")

INF = 999999999
NEEDLEMAN_WUNSCH, SMITH_WATERMAN = 0, 1
INIT_WITH_ZEROES, INIT_WITH_GAPS, INIT_WITH_CUSTOM_DATA, INIT_WITH_GAPS_OPENED = 0, 1, 2, 3

ERRORS = {-1: "EINVAL", -2: "EHIP", -3: "ENOGPU", -4: "ENOMEM", -5: "ETIMEOUT", -6: "ESTATE", -7: "EOVERFLOW16",
          -8: "ETRACEBACK", -9: "ETOOLARGE", -10: "EBOUND"}

# mi355sw_config.flags / .verbosity (include/mi355sw.h)
F_FORCE_GENERIC_COMPARE, F_FORCE_INT32, F_NO_DIAGONAL_SEED, F_NO_SEED_PASS, F_NO_PRUNE_PROBE = 1, 2, 4, 8, 16
F_TWO_PHASE, F_NO_MIXED, F_NO_SHARED_BEST, F_NO_BATCH, F_NO_HOST_COUNTER, F_NO_WINDOW, F_STAIRCASE_SEED = 32, 64, 128, 256, 512, 1024, 2048
F_GENERATE_GAP_COLUMNS = 4096      # (ABI 7's opt-in; the default since ABI 8, accepted and ignored)
F_DETERMINISTIC_PRUNE = 16384      # reproducible special rows under block pruning
F_NO_GOAL_SWEEP_HEIGHTS = 32768
F_STREAM_GAP_COLUMNS = 8192        # gap-initialised first columns taken from the manager's stream instead of made on the device
V_MESSAGES, V_JOBS, V_SEED_TILES, V_BATCH, V_DEBUG_WORDS = 1, 2, 4, 8, 16

# The C library reads no environment variable (ABI 7): the MI355SW_* switches live HERE, in the Python front, and are
# translated into mi355sw_config fields -- when an engine is created and again before every call that starts work
# (MI355Aligner._sync_config), so that a test or a tool may flip one between two calls on the same engine.
_ENV_FLAGS = {"MI355SW_NO_DIAGONAL_SEED": F_NO_DIAGONAL_SEED, "MI355SW_NOSEED": F_NO_SEED_PASS, "MI355SW_NO_PRUNE_PROBE": F_NO_PRUNE_PROBE,
              "MI355SW_TWO_PHASE": F_TWO_PHASE, "MI355SW_NO_MIXED": F_NO_MIXED, "MI355SW_NO_SHARED_BEST": F_NO_SHARED_BEST,
              "MI355SW_NO_BATCH": F_NO_BATCH, "MI355SW_NOHOST": F_NO_HOST_COUNTER, "MI355SW_NO_WINDOW": F_NO_WINDOW, "MI355SW_STAIRCASE_SEED": F_STAIRCASE_SEED,
              "MI355SW_STREAM_GAP_COLUMNS": F_STREAM_GAP_COLUMNS, "MI355SW_DETERMINISTIC_PRUNE": F_DETERMINISTIC_PRUNE,
              "MI355SW_NO_GOAL_SWEEP_HEIGHTS": F_NO_GOAL_SWEEP_HEIGHTS}
_ENV_VERBOSITY = {"MI355SW_VERBOSE": V_MESSAGES, "MI355SW_VERBOSE_JOBS": V_JOBS, "MI355SW_VERBOSE_TILES": V_SEED_TILES,
                  "MI355SW_BATCH_DEBUG": V_BATCH, "MI355SW_DEBUG": V_DEBUG_WORDS}


def env_switches():
    """(flags, verbosity, wait_seconds, fault_overflow_strip_plus1, stream_priority, trace_path) as the environment has them now"""
    flags = 0
    for name, bit in _ENV_FLAGS.items():
        if os.environ.get(name):
            flags |= bit
    verbosity = 0
    for name, bit in _ENV_VERBOSITY.items():
        if os.environ.get(name):
            verbosity |= bit
    wait_s = float(os.environ.get("MI355SW_WAIT_S") or 0.0)
    fault = os.environ.get("MI355SW_FAULT_OVERFLOW_STRIP")
    prio = {"low": 2, "normal": 1}.get(os.environ.get("MI355SW_STREAM_PRIO", ""), 0)
    return flags, verbosity, max(wait_s, 0.0), (int(fault) + 1) if fault not in (None, "") else 0, prio, os.environ.get("MI355SW_TRACE") or None


class AlignerError(RuntimeError):
    pass


class Cell(C.Structure):
    _fields_ = [("h", C.c_int32), ("f", C.c_int32)]


class Score(C.Structure):
    _fields_ = [("i", C.c_int32), ("j", C.c_int32), ("score", C.c_int32)]


class ScoreParams(C.Structure):
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("gap_open", C.c_int32), ("gap_ext", C.c_int32)]


class MatchResult(C.Structure):
    _fields_ = [("found", C.c_int32), ("k", C.c_int32), ("score", C.c_int32), ("type", C.c_int32)]


class Partition(C.Structure):
    """M/libmasa/Partition.hpp: half-open [i0,i1) x [j0,j1), sequence-relative."""
    _fields_ = [("i0", C.c_int32), ("j0", C.c_int32), ("i1", C.c_int32), ("j1", C.c_int32)]

    def __init__(self, i0=0, j0=0, i1=0, j1=0):
        super().__init__(i0, j0, i1, j1)

    def getHeight(self):
        return self.i1 - self.i0

    def getWidth(self):
        return self.j1 - self.j0


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("rows_per_lane", C.c_int32), ("waves", C.c_int32),
                ("flags", C.c_int32), ("max_special_bytes", C.c_int64), ("block_score_columns", C.c_int32),
                ("verbosity", C.c_int32), ("wait_seconds", C.c_double), ("fault_overflow_strip_plus1", C.c_int32),
                ("stream_priority", C.c_int32), ("trace_path", C.c_char_p), ("batch_rows_per_lane", C.c_int32), ("reserved32_", C.c_int32),
                ("reserved_", C.c_int64 * 3)]


class Capabilities(C.Structure):
    _fields_ = [(k, C.c_int32) for k in (
        "dispatch_last_cell", "dispatch_last_row", "dispatch_last_column",
        "dispatch_special_row", "dispatch_special_column",
        "dispatch_scores", "dispatch_block_scores", "dispatch_best_score",
        "customize_first_row", "customize_first_column",
        "process_partition", "variable_penalties", "block_pruning",
        "needleman_wunsch", "smith_waterman", "fork_processes",
        "maximum_seq0_len", "maximum_seq1_len")]


class Stats(C.Structure):
    _fields_ = [("cells", C.c_int64), ("processed_cells", C.c_int64), ("kernel_ms", C.c_double),
                ("total_ms", C.c_double), ("kernel_launches", C.c_int32), ("strips", C.c_int32),
                ("strip_rows", C.c_int32), ("waves", C.c_int32), ("profile_kernel", C.c_int32),
                ("algorithmic_bytes", C.c_int64), ("pruned_cells", C.c_int64), ("wait_ms", C.c_double),
                ("strips_first", C.c_int32), ("strip_rows_second", C.c_int32), ("restarts", C.c_int32), ("reserved_", C.c_int32),
                ("seed_ms", C.c_double), ("kernel", C.c_char * 64)]


class StreamParams(C.Structure):
    _fields_ = [("recurrence_type", C.c_int32),
                ("first_row_init_type", C.c_int32), ("first_row_start_offset", C.c_int32),
                ("first_row", C.c_void_p),
                ("first_column_init_type", C.c_int32), ("first_column_start_offset", C.c_int32),
                ("stream_first_column", C.c_int32),
                ("first_column", C.c_void_p),
                ("want_last_column", C.c_int32), ("want_last_row", C.c_int32),
                ("special_row_interval", C.c_int32), ("track_best", C.c_int32), ("force_int32", C.c_int32),
                ("prune_blocks", C.c_int32), ("prune_rows", C.c_int32), ("prune_cols", C.c_int32),
                ("first_column_port", C.c_int32), ("last_column_port", C.c_int32),
                ("first_column_resume_rows", C.c_int32), ("share_best", C.c_int32),
                ("have_initial_bound", C.c_int32), ("initial_bound", C.c_int32)]


class Stage4Stats(C.Structure):
    _fields_ = [("steps", C.c_int32), ("kernel_ms", C.c_double), ("dp_cells", C.c_int64), ("partitions", C.c_int64)]


class Stage5Totals(C.Structure):
    _fields_ = [("score", C.c_int64), ("matches", C.c_int64), ("mismatches", C.c_int64), ("gap_open", C.c_int64),
                ("gap_extensions", C.c_int64)]


class PortHandle(C.Structure):
    """mi355sw_port_handle: what the owner of a column port sends to the band on its left (hipIpc handle + size)."""
    _fields_ = [("ipc", C.c_ubyte * 64), ("bytes", C.c_int64), ("rows", C.c_int32), ("device", C.c_int32)]

    def tobytes(self):
        return bytes(memoryview(self))

    @classmethod
    def frombytes(cls, b):
        return cls.from_buffer_copy(bytes(b))


_VP = C.c_void_p
CB_INT = C.CFUNCTYPE(C.c_int32, _VP)
CB_PART = C.CFUNCTYPE(None, _VP, C.POINTER(Partition))
CB_RECV = C.CFUNCTYPE(None, _VP, C.POINTER(Cell), C.c_int32)
CB_DISP = C.CFUNCTYPE(None, _VP, C.c_int32, C.POINTER(Cell), C.c_int32)
CB_SCORE = C.CFUNCTYPE(None, _VP, Score, C.c_int32, C.c_int32)
CB_VALUE = C.CFUNCTYPE(None, _VP, C.c_int32, C.c_int32, C.c_int32)


class ManagerTable(C.Structure):
    _fields_ = [("get_recurrence_type", CB_INT), ("get_special_row_interval", CB_INT),
                ("get_first_column_init_type", CB_INT), ("get_first_row_init_type", CB_INT),
                ("get_super_partition", CB_PART),
                ("receive_first_row", CB_RECV), ("receive_first_column", CB_RECV),
                ("dispatch_column", CB_DISP), ("dispatch_row", CB_DISP), ("dispatch_score", CB_SCORE),
                ("must_continue", CB_INT), ("must_dispatch_last_cell", CB_INT),
                ("must_dispatch_last_row", CB_INT), ("must_dispatch_last_column", CB_INT),
                ("must_dispatch_special_rows", CB_INT), ("must_dispatch_scores", CB_INT),
                ("must_prune_blocks", CB_INT), ("dispatch_strip_value", CB_VALUE)]


# every symbol include/mi355sw.h declares (tests check the .so exports all of them)
ABI_SYMBOLS = [
    "mi355sw_create", "mi355sw_configure", "mi355sw_destroy", "mi355sw_last_error", "mi355sw_abi_version", "mi355sw_build_id",
    "mi355sw_get_capabilities", "mi355sw_get_score_parameters", "mi355sw_set_rows_per_lane",
    "mi355sw_set_sequences", "mi355sw_unset_sequences", "mi355sw_align_partition", "mi355sw_align_partitions",
    "mi355sw_process_block", "mi355sw_match_last_column", "mi355sw_progress",
    "mi355sw_processed_cells", "mi355sw_get_stats",
    "mi355sw_stream_begin", "mi355sw_seed_bound", "mi355sw_stream_feed_column", "mi355sw_stream_poll",
    "mi355sw_stream_read_column", "mi355sw_stream_read_special_row", "mi355sw_stream_read_last_row",
    "mi355sw_stream_abort", "mi355sw_stream_end", "mi355sw_stream_strip_scores",
    "mi355sw_stream_best_hint", "mi355sw_stream_running_best",
    "mi355sw_port_create", "mi355sw_port_open", "mi355sw_port_attach", "mi355sw_port_reset", "mi355sw_port_rows_ready", "mi355sw_port_read",
    "mi355sw_port_local_pointers", "mi355sw_port_close", "mi355sw_stage4", "mi355sw_free", "mi355sw_stage5", "mi355sw_stage6_text",
    "mi355sw_crosspoints_text", "mi355sw_device_count", "mi355sw_device_info",
]

_lib = None


def build_library(force=False):
    """Compile csrc/ for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src = os.path.join(HERE, "csrc")
    deps = [os.path.join(src, f) for f in sorted(os.listdir(src))
            if f.endswith((".cpp", ".hip", ".inc", ".h", ".py", ".sh")) or f == "Makefile"] + [INCLUDE_PATH]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    subprocess.check_call(["make", "-C", src], stdout=subprocess.DEVNULL)
    return LIB_PATH


def library_build_id():
    """identity of the device code the LOADED library was built from (mi355sw_build_id)"""
    return load_library().mi355sw_build_id().decode()


def source_build_id():
    """the same hash over the sources in the tree now (csrc/build_id.py); differs from library_build_id() when the
    library is stale"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mi355sw_build_id", os.path.join(HERE, "csrc", "build_id.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.kernel_build_id()


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AlignerError("libmi355sw.so is not built (run __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    H = C.c_void_p
    lib.mi355sw_create.argtypes = [C.POINTER(Config), C.POINTER(H)]
    lib.mi355sw_configure.argtypes = [H, C.POINTER(Config)]
    lib.mi355sw_destroy.argtypes = [H]
    lib.mi355sw_destroy.restype = None
    lib.mi355sw_last_error.argtypes = [H]
    lib.mi355sw_last_error.restype = C.c_char_p
    lib.mi355sw_build_id.restype = C.c_char_p
    lib.mi355sw_get_capabilities.argtypes = [H, C.POINTER(Capabilities)]
    lib.mi355sw_get_score_parameters.argtypes = [H, C.POINTER(ScoreParams)]
    lib.mi355sw_set_rows_per_lane.argtypes = [H, C.c_int32]
    lib.mi355sw_set_sequences.argtypes = [H, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    lib.mi355sw_unset_sequences.argtypes = [H]
    lib.mi355sw_align_partition.argtypes = [H, C.POINTER(Partition), C.POINTER(ManagerTable), C.c_void_p]
    lib.mi355sw_align_partitions.argtypes = [H, C.c_int32, C.POINTER(Partition), C.POINTER(C.POINTER(ManagerTable)), C.POINTER(C.c_void_p)]
    lib.mi355sw_process_block.argtypes = [H, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, C.POINTER(Score)]
    lib.mi355sw_match_last_column.argtypes = [H, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(MatchResult)]
    lib.mi355sw_progress.argtypes = [H, C.c_char_p, C.c_size_t]
    lib.mi355sw_processed_cells.argtypes = [H]
    lib.mi355sw_processed_cells.restype = C.c_longlong
    lib.mi355sw_get_stats.argtypes = [H, C.POINTER(Stats)]
    lib.mi355sw_stream_begin.argtypes = [H, C.POINTER(Partition), C.POINTER(StreamParams)]
    lib.mi355sw_seed_bound.argtypes = [H, C.POINTER(Partition), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.mi355sw_stream_feed_column.argtypes = [H, C.c_int32, C.c_void_p, C.c_int32]
    lib.mi355sw_stream_poll.argtypes = [H, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.mi355sw_stream_read_column.argtypes = [H, C.c_int32, C.c_void_p, C.c_int32]
    lib.mi355sw_stream_read_special_row.argtypes = [H, C.c_int32, C.POINTER(C.c_int32), C.c_void_p, C.c_int32, C.c_int32]
    lib.mi355sw_stream_read_last_row.argtypes = [H, C.c_void_p, C.c_int32, C.c_int32]
    lib.mi355sw_stream_abort.argtypes = [H]
    lib.mi355sw_stream_best_hint.argtypes = [H, C.c_int32]
    lib.mi355sw_stream_running_best.argtypes = [H, C.POINTER(C.c_int32)]
    lib.mi355sw_stream_end.argtypes = [H, C.POINTER(Score), C.POINTER(C.c_int32)]
    lib.mi355sw_stream_strip_scores.argtypes = [H, C.c_void_p, C.c_int32]
    lib.mi355sw_port_create.argtypes = [H, C.c_int32, C.POINTER(PortHandle)]
    lib.mi355sw_port_open.argtypes = [H, C.POINTER(PortHandle)]
    lib.mi355sw_port_attach.argtypes = [H, H]
    lib.mi355sw_port_reset.argtypes = [H]
    lib.mi355sw_port_rows_ready.argtypes = [H, C.POINTER(C.c_int32)]
    lib.mi355sw_port_read.argtypes = [H, C.c_int32, C.c_void_p, C.c_int32]
    lib.mi355sw_port_local_pointers.argtypes = [H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    lib.mi355sw_port_close.argtypes = [H]
    lib.mi355sw_stage4.argtypes = [H, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                   C.POINTER(Stage4Stats)]
    lib.mi355sw_free.argtypes = [C.c_void_p]
    lib.mi355sw_free.restype = None
    lib.mi355sw_stage5.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                   C.POINTER(Stage5Totals), C.POINTER(C.c_int32)]
    lib.mi355sw_stage6_text.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_int64), C.POINTER(Stage5Totals)]
    lib.mi355sw_crosspoints_text.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    lib.mi355sw_device_info.argtypes = [C.c_int32, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int64)]
    _lib = lib
    return lib


def stage5_events(data0, data1, crosspoints):
    """mi355sw_stage5 (host code, no GPU): gap events of the exact alignment through `crosspoints` [(type, i, j, score)].
    Returns (rows of the events for sequence 0's gap list, columns of those for sequence 1's, totals dict)."""
    lib = load_library()
    d0, d1 = _as_u8(data0), _as_u8(data1)
    cp = np.ascontiguousarray(crosspoints, dtype=np.int32).reshape(-1, 4)
    g0, g1, n0, n1 = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_int64()
    tot, bad = Stage5Totals(), C.c_int32(-1)
    rc = lib.mi355sw_stage5(d0.ctypes.data, len(d0), d1.ctypes.data, len(d1), cp.ctypes.data, len(cp), C.byref(g0), C.byref(n0),
                            C.byref(g1), C.byref(n1), C.byref(tot), C.byref(bad))
    if rc != 0:
        raise AlignerError("stage5: %s at partition %d (crosspoints %s -> %s)" % (
            ERRORS.get(rc, rc), bad.value, tuple(cp[bad.value]) if 0 <= bad.value < len(cp) else None,
            tuple(cp[bad.value + 1]) if 0 <= bad.value + 1 < len(cp) else None))
    try:
        a0 = np.frombuffer((C.c_int32 * max(n0.value, 1)).from_address(g0.value), dtype=np.int32)[:n0.value].copy()
        a1 = np.frombuffer((C.c_int32 * max(n1.value, 1)).from_address(g1.value), dtype=np.int32)[:n1.value].copy()
    finally:
        lib.mi355sw_free(g0)
        lib.mi355sw_free(g1)
    return a0, a1, {k: int(getattr(tot, k)) for k, _ in Stage5Totals._fields_}


def stage6_body(forward0, forward1, start, end, gaps0, gaps1, raw_score):
    """mi355sw_stage6_text (host code, no GPU): the bytes of alignment.NN.txt after its three header lines."""
    lib = load_library()
    d0, d1 = _as_u8(forward0), _as_u8(forward1)
    g0 = np.ascontiguousarray(gaps0, dtype=np.int32).reshape(-1, 2)
    g1 = np.ascontiguousarray(gaps1, dtype=np.int32).reshape(-1, 2)
    text, n, tot = C.c_void_p(), C.c_int64(), Stage5Totals()
    rc = lib.mi355sw_stage6_text(d0.ctypes.data, len(d0), d1.ctypes.data, len(d1), int(start[0]), int(start[1]), int(end[0]), int(end[1]),
                                 g0.ctypes.data, len(g0), g1.ctypes.data, len(g1), int(raw_score), C.byref(text), C.byref(n), C.byref(tot))
    if rc == -8:
        raise RuntimeError("Stage6 error: Alignment score is different (%d != %d)" % (tot.score, raw_score))
    if rc != 0:
        raise AlignerError("stage6: %s" % ERRORS.get(rc, rc))
    try:
        return C.string_at(text.value, n.value)
    finally:
        lib.mi355sw_free(text)


def crosspoints_text(points):
    """mi355sw_crosspoints_text (host code, no GPU): the bytes of a crosspoint file for an (N, 4) array of (type, i, j, score)"""
    lib = load_library()
    cp = np.ascontiguousarray(points, dtype=np.int32).reshape(-1, 4)
    text, n = C.c_void_p(), C.c_int64()
    rc = lib.mi355sw_crosspoints_text(cp.ctypes.data, len(cp), C.byref(text), C.byref(n))
    if rc != 0:
        raise AlignerError("crosspoints_text: %s" % ERRORS.get(rc, rc))
    try:
        return C.string_at(text.value, n.value)
    finally:
        lib.mi355sw_free(text)


def _as_u8(seq):
    if isinstance(seq, (bytes, bytearray)):
        return np.frombuffer(bytes(seq), dtype=np.uint8)
    if isinstance(seq, str):
        return np.frombuffer(seq.encode("ascii"), dtype=np.uint8)
    return np.ascontiguousarray(seq, dtype=np.uint8)


def _cells(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    assert a.ndim == 2 and a.shape[1] == 2
    return a


class MI355Aligner:
    """Python mirror of the IAligner a MASA extension implements (M/libmasa/IAligner.hpp)."""

    def __init__(self, device=-1, rows_per_lane=0, waves=0, flags=0, max_special_bytes=0, block_score_columns=0, verbosity=0,
                 wait_seconds=0.0):
        self._lib = load_library()
        self._h = C.c_void_p()
        # what the caller asked for; the environment's switches are OR-ed in by _make_config
        self._opts = dict(device=int(device), rows_per_lane=int(rows_per_lane), waves=int(waves), flags=int(flags),
                          max_special_bytes=int(max_special_bytes), block_score_columns=int(block_score_columns),
                          verbosity=int(verbosity), wait_seconds=float(wait_seconds))
        cfg = self._make_config()
        self._cfg_key = self._config_key(cfg)
        self._rows_per_lane = int(rows_per_lane)
        rc = self._lib.mi355sw_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise AlignerError("mi355sw_create failed: %s (the engine needs a gfx950 GPU; no CPU fallback)"
                               % ERRORS.get(rc, rc))
        self._seqs = None

    # -- plumbing -------------------------------------------------------------------------
    def _make_config(self):
        o = self._opts
        eflags, everb, ewait, efault, eprio, etrace = env_switches()
        cfg = Config()
        cfg.device, cfg.rows_per_lane, cfg.waves = o["device"], o["rows_per_lane"], o["waves"]
        cfg.flags = o["flags"] | eflags
        cfg.max_special_bytes, cfg.block_score_columns = o["max_special_bytes"], o["block_score_columns"]
        cfg.verbosity = o["verbosity"] | everb
        cfg.wait_seconds = o["wait_seconds"] if o["wait_seconds"] > 0 else ewait
        cfg.fault_overflow_strip_plus1 = o.get("fault_overflow_strip_plus1", 0) or efault
        cfg.stream_priority = eprio
        cfg.trace_path = etrace.encode() if etrace else None
        cfg.batch_rows_per_lane = o.get("batch_rows_per_lane", 0)
        return cfg

    @staticmethod
    def _config_key(cfg):
        return (cfg.rows_per_lane, cfg.waves, cfg.flags, cfg.max_special_bytes, cfg.block_score_columns, cfg.verbosity, cfg.wait_seconds,
                cfg.fault_overflow_strip_plus1, cfg.trace_path, cfg.batch_rows_per_lane)

    def _sync_config(self):
        """hand the library the switches as they are NOW (constructor arguments, configure(), the MI355SW_* environment)"""
        cfg = self._make_config()
        key = self._config_key(cfg)
        if key != self._cfg_key:
            self._check(self._lib.mi355sw_configure(self._h, C.byref(cfg)), "configure")
            self._cfg_key = key

    def configure(self, **kw):
        """change switches of a live engine: flags, verbosity, wait_seconds, waves, rows_per_lane, max_special_bytes,
        block_score_columns, fault_overflow_strip_plus1, batch_rows_per_lane (mi355sw_configure; no stream may be active)"""
        for k in kw:
            if k not in ("flags", "verbosity", "wait_seconds", "waves", "rows_per_lane", "max_special_bytes", "block_score_columns",
                         "fault_overflow_strip_plus1", "batch_rows_per_lane"):
                raise TypeError("configure: unknown option %r" % k)
        self._opts.update(kw)
        if "rows_per_lane" in kw:
            self._rows_per_lane = int(kw["rows_per_lane"])
        self._sync_config()

    def setFlag(self, bit, on=True, defer=False):
        """one MI355SW_F_* bit on or off for the calls that follow.  defer: only noted here -- the library gets it with the
        next call that starts work (_sync_config).  For clean-up paths: mi355sw_configure refuses while a stream is active,
        and an error raised from a `finally` would replace the one that is on its way out."""
        f = self._opts["flags"]
        f = (f | bit) if on else (f & ~bit)
        if defer:
            self._opts["flags"] = f
        else:
            self.configure(flags=f)

    def getFlags(self):
        return self._opts["flags"]

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.mi355sw_last_error(self._h)
            raise AlignerError("%s: %s %s" % (what, ERRORS.get(rc, rc), msg.decode() if msg else ""))

    def close(self):
        if self._h:
            self._lib.mi355sw_destroy(self._h)
            self._h = C.c_void_p()

    finalize = close  # IAligner::finalize

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- IAligner -------------------------------------------------------------------------
    def getCapabilities(self):
        c = Capabilities()
        self._check(self._lib.mi355sw_get_capabilities(self._h, C.byref(c)), "getCapabilities")
        return {k: getattr(c, k) for k, _ in Capabilities._fields_}

    def setRowsPerLane(self, rows_per_lane):
        """strip height (64 * rows_per_lane rows) of the partitions that follow; 0 = the engine's cost model"""
        self._check(self._lib.mi355sw_set_rows_per_lane(self._h, int(rows_per_lane)), "setRowsPerLane")
        self._rows_per_lane = int(rows_per_lane)
        self._opts["rows_per_lane"] = int(rows_per_lane)
        # (the library has this one value already; anything else that changed meanwhile is still to be handed over by _sync_config)
        self._cfg_key = (int(rows_per_lane),) + tuple(self._cfg_key[1:])

    def getRowsPerLane(self):
        return getattr(self, "_rows_per_lane", None)

    def getScoreParameters(self):
        p = ScoreParams()
        self._check(self._lib.mi355sw_get_score_parameters(self._h, C.byref(p)), "getScoreParameters")
        return {"match": p.match, "mismatch": p.mismatch, "gap_open": p.gap_open, "gap_ext": p.gap_ext}

    def setSequences(self, seq0, seq1, seq0_len=None, seq1_len=None):
        self._sync_config()
        s0, s1 = _as_u8(seq0), _as_u8(seq1)
        l0 = len(s0) if seq0_len is None else seq0_len
        l1 = len(s1) if seq1_len is None else seq1_len
        self._check(self._lib.mi355sw_set_sequences(self._h, s0.ctypes.data, s1.ctypes.data, l0, l1), "setSequences")
        self._seqs = (l0, l1)

    def unsetSequences(self):
        self._check(self._lib.mi355sw_unset_sequences(self._h), "unsetSequences")
        self._seqs = None

    def alignPartition(self, partition, manager):
        """manager: object with the IManager methods (see manager.Stage1Manager)."""
        self._sync_config()
        table, keep = make_manager_table(manager)
        self._check(self._lib.mi355sw_align_partition(self._h, C.byref(partition), C.byref(table), None),
                    "alignPartition")
        if getattr(manager, "_callback_error", None):
            raise manager._callback_error
        del keep

    batch_rows_per_lane_choices = (4, 8, 16)      # mi355sw_config.batch_rows_per_lane

    def alignPartitions(self, partitions, managers, rows_per_lane=None):
        """mi355sw_align_partitions: independent partitions side by side in one kernel launch; managers[k] is served
        exactly as alignPartition(partitions[k], managers[k]) would serve it, the calls of different managers interleave.
        rows_per_lane: strip height of THIS launch (mi355sw_config.batch_rows_per_lane: 4, 8 or 16 = 256 / 512 / 1024 rows)"""
        n = len(partitions)
        assert n == len(managers)
        if n == 0:
            return
        if rows_per_lane is not None:
            saved = self._opts.get("batch_rows_per_lane", 0)
            self._opts["batch_rows_per_lane"] = int(rows_per_lane)
            try:
                return self.alignPartitions(partitions, managers)
            finally:
                self._opts["batch_rows_per_lane"] = saved
        self._sync_config()
        parts = (Partition * n)(*[Partition(p.i0, p.j0, p.i1, p.j1) for p in partitions])
        tables, keeps = zip(*[make_manager_table(m) for m in managers])
        ptrs = (C.POINTER(ManagerTable) * n)(*[C.pointer(t) for t in tables])
        rc = self._lib.mi355sw_align_partitions(self._h, n, parts, ptrs, None)
        for m in managers:
            if getattr(m, "_callback_error", None):
                raise m._callback_error
        self._check(rc, "alignPartitions")
        del keeps

    def processBlock(self, row, col, i0, j0, i1, j1, recurrence_type):
        """AbstractBlockProcessor::processBlock: row (n,2), col (m+1,2) int32, updated in place."""
        assert row.dtype == np.int32 and col.dtype == np.int32 and row.flags.c_contiguous and col.flags.c_contiguous
        self._sync_config()
        s = Score()
        self._check(self._lib.mi355sw_process_block(self._h, row.ctypes.data, col.ctypes.data, i0, j0, i1, j1,
                                                    recurrence_type, C.byref(s)), "processBlock")
        return (s.i, s.j, s.score)

    def matchLastColumn(self, buffer, base, goal_score):
        b, a = _cells(buffer), _cells(base)
        r = MatchResult()
        self._check(self._lib.mi355sw_match_last_column(self._h, b.ctypes.data, a.ctypes.data, len(b), goal_score,
                                                        C.byref(r)), "matchLastColumn")
        return {"found": bool(r.found), "k": r.k, "score": r.score, "type": r.type}

    def getProgressString(self):
        buf = C.create_string_buffer(256)
        self._lib.mi355sw_progress(self._h, buf, 256)
        return buf.value.decode()

    def getProcessedCells(self):
        return self._lib.mi355sw_processed_cells(self._h)

    def getStatistics(self):
        s = Stats()
        self._check(self._lib.mi355sw_get_stats(self._h, C.byref(s)), "getStatistics")
        out = {k: getattr(s, k) for k, _ in Stats._fields_}
        out["kernel"] = out["kernel"].decode()
        return out

    # -- streaming form (column-band driver) ---------------------------------------------------
    def seedBound(self, partition, recurrence_type=SMITH_WATERMAN):
        """mi355sw_seed_bound: the diagonal seed pass over `partition` -- the WHOLE matrix a chain of bands divides among itself
        -- on this engine's GPU; returns the value for streamBegin(initial_bound=...) of every band, or None when there is
        none (an unrelated pair, a small matrix)."""
        self._sync_config()
        have, bound = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.mi355sw_seed_bound(self._h, C.byref(partition), int(recurrence_type), C.byref(have), C.byref(bound)), "seedBound")
        return int(bound.value) if have.value else None

    def streamBegin(self, partition, recurrence_type=SMITH_WATERMAN, first_row_init_type=INIT_WITH_ZEROES,
                    first_row_start_offset=0, first_row=None, first_column_init_type=INIT_WITH_ZEROES,
                    first_column_start_offset=0, stream_first_column=False, first_column=None,
                    want_last_column=False, want_last_row=False, special_row_interval=0, track_best=True,
                    force_int32=False, prune_blocks=False, prune_rows=0, prune_cols=0,
                    first_column_port=False, last_column_port=False, first_column_resume_rows=0, share_best=False,
                    initial_bound=None):
        self._sync_config()
        sp = StreamParams()
        sp.recurrence_type = recurrence_type
        sp.first_row_init_type, sp.first_row_start_offset = first_row_init_type, first_row_start_offset
        keep = []
        if first_row is not None:
            a = _cells(first_row); keep.append(a); sp.first_row = a.ctypes.data
        sp.first_column_init_type, sp.first_column_start_offset = first_column_init_type, first_column_start_offset
        sp.stream_first_column = int(stream_first_column)
        if first_column is not None:
            a = _cells(first_column); keep.append(a); sp.first_column = a.ctypes.data
        sp.want_last_column, sp.want_last_row = int(want_last_column), int(want_last_row)
        sp.special_row_interval, sp.track_best = special_row_interval, int(track_best)
        sp.force_int32 = int(force_int32)
        sp.prune_blocks, sp.prune_rows, sp.prune_cols = int(prune_blocks), int(prune_rows), int(prune_cols)
        sp.first_column_port, sp.last_column_port = int(first_column_port), int(last_column_port)
        sp.first_column_resume_rows = int(first_column_resume_rows)
        sp.share_best = int(share_best)
        if initial_bound is not None:
            sp.have_initial_bound, sp.initial_bound = 1, int(initial_bound)
        self._check(self._lib.mi355sw_stream_begin(self._h, C.byref(partition), C.byref(sp)), "streamBegin")
        self._stream_part = partition

    def streamFeedColumn(self, row, cells):
        a = _cells(cells)
        self._check(self._lib.mi355sw_stream_feed_column(self._h, row, a.ctypes.data, len(a)), "streamFeedColumn")

    def streamPoll(self):
        rows, fin = C.c_int32(), C.c_int32()
        self._check(self._lib.mi355sw_stream_poll(self._h, C.byref(rows), C.byref(fin)), "streamPoll")
        return rows.value, bool(fin.value)

    def streamReadColumn(self, row, length):
        out = np.empty((length, 2), dtype=np.int32)
        self._check(self._lib.mi355sw_stream_read_column(self._h, row, out.ctypes.data, length), "streamReadColumn")
        return out

    def streamReadSpecialRow(self, k, col=0, length=None):
        n = self._stream_part.getWidth()
        length = n - col if length is None else length
        out = np.empty((length, 2), dtype=np.int32)
        dp = C.c_int32()
        self._check(self._lib.mi355sw_stream_read_special_row(self._h, k, C.byref(dp), out.ctypes.data, col, length),
                    "streamReadSpecialRow")
        return dp.value, out

    def streamReadLastRow(self, col=0, length=None):
        n = self._stream_part.getWidth()
        length = n - col if length is None else length
        out = np.empty((length, 2), dtype=np.int32)
        self._check(self._lib.mi355sw_stream_read_last_row(self._h, out.ctypes.data, col, length), "streamReadLastRow")
        return out

    def streamAbort(self):
        self._check(self._lib.mi355sw_stream_abort(self._h), "streamAbort")

    def streamBestHint(self, score):
        """a score some alignment of the super-partition reaches (found elsewhere): lower bound for block pruning"""
        self._check(self._lib.mi355sw_stream_best_hint(self._h, int(score)), "streamBestHint")

    def streamRunningBest(self):
        """best score known to the running kernel at its last strip hand-over (-INF before the first)"""
        v = C.c_int32()
        self._check(self._lib.mi355sw_stream_running_best(self._h, C.byref(v)), "streamRunningBest")
        return v.value

    def streamEnd(self):
        s, nsp = Score(), C.c_int32()
        self._check(self._lib.mi355sw_stream_end(self._h, C.byref(s), C.byref(nsp)), "streamEnd")
        return (s.i, s.j, s.score), nsp.value

    def streamStripScores(self, max_count=1 << 22):
        buf = np.empty((max_count, 3), dtype=np.int32)
        cnt = self._lib.mi355sw_stream_strip_scores(self._h, buf.ctypes.data, max_count)
        return buf[:cnt].copy()

    # -- stage 4 (Myers-Miller refinement of the stage-3 crosspoints, M/stage4/sw_stage4.cpp) ------------------
    def stage4(self, crosspoints, max_partition_size=16, as_array=False):
        """crosspoints: [(type, i, j, score), ...] as in crosspoint_03.NN; returns (refined list, stats dict); as_array: the
        list as an (N, 4) int32 array (millions of points at sizes like C3: tuples of Python ints cost seconds)"""
        cp = np.ascontiguousarray(crosspoints, dtype=np.int32).reshape(-1, 4)
        out, n, st = C.c_void_p(), C.c_int32(), Stage4Stats()
        self._check(self._lib.mi355sw_stage4(self._h, cp.ctypes.data, len(cp), max_partition_size, C.byref(out), C.byref(n),
                                             C.byref(st)), "stage4")
        if n.value > 0 and out.value:
            buf = (C.c_int32 * (n.value * 4)).from_address(out.value)
            res = np.frombuffer(buf, dtype=np.int32).reshape(n.value, 4).copy()
        else:
            res = np.empty((0, 4), dtype=np.int32)
        if out.value:
            self._lib.mi355sw_free(out)
        stats = {k: getattr(st, k) for k, _ in Stage4Stats._fields_}
        return (res if as_array else [tuple(r) for r in res.tolist()]), stats

    # -- column ports (boundary column GPU to GPU over xGMI) ----------------------------------------
    def portCreate(self, rows):
        """inbound port of this band: returns the PortHandle to send to the band on the left"""
        ph = PortHandle()
        self._check(self._lib.mi355sw_port_create(self._h, rows, C.byref(ph)), "portCreate")
        return ph

    def portOpen(self, port_handle):
        """map the right neighbour's inbound port (another process) as this band's outbound port"""
        self._check(self._lib.mi355sw_port_open(self._h, C.byref(port_handle)), "portOpen")

    def portAttach(self, downstream):
        """same process: `downstream`'s inbound port becomes this band's outbound port"""
        self._check(self._lib.mi355sw_port_attach(self._h, downstream._h), "portAttach")

    def portReset(self):
        self._check(self._lib.mi355sw_port_reset(self._h), "portReset")

    def portRowsReady(self):
        r = C.c_int32()
        self._check(self._lib.mi355sw_port_rows_ready(self._h, C.byref(r)), "portRowsReady")
        return r.value

    def portRead(self, row, length):
        out = np.empty((length, 2), dtype=np.int32)
        self._check(self._lib.mi355sw_port_read(self._h, row, out.ctypes.data, length), "portRead")
        return out

    def portLocalPointers(self):
        cells, counter = C.c_void_p(), C.c_void_p()
        self._check(self._lib.mi355sw_port_local_pointers(self._h, C.byref(cells), C.byref(counter)), "portLocalPointers")
        return cells.value, counter.value

    def portClose(self):
        self._check(self._lib.mi355sw_port_close(self._h), "portClose")


def make_manager_table(mgr):
    """Wrap a Python IManager-like object into the C callback table (IManager.hpp:98-313)."""
    mgr._callback_error = None

    def guard(fn, default=0):
        def w(*a):
            try:
                r = fn(*a)
                return default if r is None else r
            except BaseException as e:  # never let an exception cross the C boundary
                if mgr._callback_error is None:
                    mgr._callback_error = e
                return default
        return w

    def recv(method):
        def f(_u, buf, length):
            arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_int32)), shape=(length, 2))
            method(arr, length)
        return guard(f, None)

    def disp(method):
        def f(_u, pos, buf, length):
            arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_int32)), shape=(length, 2))
            method(pos, arr, length)
        return guard(f, None)

    def superp(_u, out):
        p = mgr.getSuperPartition()
        out[0].i0, out[0].j0, out[0].i1, out[0].j1 = p.i0, p.j0, p.i1, p.j1

    def score(_u, s, bx, by):
        mgr.dispatchScore((s.i, s.j, s.score), bx, by)

    keep = dict(
        get_recurrence_type=CB_INT(guard(lambda u: mgr.getRecurrenceType())),
        get_special_row_interval=CB_INT(guard(lambda u: mgr.getSpecialRowInterval())),
        get_first_column_init_type=CB_INT(guard(lambda u: mgr.getFirstColumnInitType())),
        get_first_row_init_type=CB_INT(guard(lambda u: mgr.getFirstRowInitType())),
        get_super_partition=CB_PART(guard(superp, None)),
        receive_first_row=CB_RECV(recv(mgr.receiveFirstRow)),
        receive_first_column=CB_RECV(recv(mgr.receiveFirstColumn)),
        dispatch_column=CB_DISP(disp(mgr.dispatchColumn)),
        dispatch_row=CB_DISP(disp(mgr.dispatchRow)),
        dispatch_score=CB_SCORE(guard(score, None)),
        must_continue=CB_INT(guard(lambda u: int(mgr.mustContinue()), 0)),
        must_dispatch_last_cell=CB_INT(guard(lambda u: int(mgr.mustDispatchLastCell()))),
        must_dispatch_last_row=CB_INT(guard(lambda u: int(mgr.mustDispatchLastRow()))),
        must_dispatch_last_column=CB_INT(guard(lambda u: int(mgr.mustDispatchLastColumn()))),
        must_dispatch_special_rows=CB_INT(guard(lambda u: int(mgr.mustDispatchSpecialRows()))),
        must_dispatch_scores=CB_INT(guard(lambda u: int(mgr.mustDispatchScores()))),
        must_prune_blocks=CB_INT(guard(lambda u: int(mgr.mustPruneBlocks()))),
    )
    if hasattr(mgr, "dispatchStripValue"):      # optional: best VALUE per strip of a two-phase run (see mi355sw.h)
        keep["dispatch_strip_value"] = CB_VALUE(guard(lambda u, lo, hi, sc: mgr.dispatchStripValue(lo, hi, sc), None))
    table = ManagerTable(**keep)
    return table, keep

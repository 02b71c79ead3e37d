"""CPU: the C-ABI library builds, loads and exports every symbol include/mi355sw.h declares; without a
GPU the engine refuses to start (no CPU fallback)."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as graft


def header_functions():
    src = open(os.path.join(graft.ROOT, "include", "mi355sw.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355sw_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(pkg):
    pkg.build_library()
    lib = ctypes.CDLL(pkg.LIB_PATH)
    names = header_functions()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(pkg.engine.ABI_SYMBOLS) == names
    lib.mi355sw_abi_version.restype = ctypes.c_int
    assert lib.mi355sw_abi_version() == 8


def test_struct_layouts_match_reference_types(pkg):
    # cell_t is 8 bytes (libmasaTypes.hpp:35-41), score_t 12 bytes (:88-95)
    assert ctypes.sizeof(pkg.engine.Cell) == 8
    assert ctypes.sizeof(pkg.engine.Score) == 12
    assert ctypes.sizeof(pkg.engine.Partition) == 16
    assert pkg.INF == 999999999


def test_no_cpu_fallback(pkg):
    """On a box without a gfx950 GPU construction must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.AlignerError):
        pkg.MI355Aligner(device=0)


def test_product_does_not_import_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkgdir = graft.PKG_DIR
    for root, _, files in os.walk(pkgdir):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(root, fn), errors="replace").read()
                assert "sw_oracle" not in txt and "import oracle" not in txt and "load_oracle" not in txt, fn
    hdr = open(os.path.join(graft.ROOT, "include", "mi355sw.h")).read()
    assert "oracle" not in hdr


def test_every_struct_of_the_python_front_has_the_headers_layout(pkg, tmp_path):
    """the ctypes mirrors of engine.py against include/mi355sw.h itself: a C program compiled here prints sizeof and the offset
    of every field of the structs that cross the boundary (ABI 7 grew mi355sw_config; a field out of place would be read as
    another switch)"""
    import subprocess
    eng = pkg.engine
    structs = {"mi355sw_config": eng.Config, "mi355sw_stats": eng.Stats, "mi355sw_stream_params": eng.StreamParams,
               "mi355sw_capabilities": eng.Capabilities, "mi355sw_manager": eng.ManagerTable, "mi355sw_port_handle": eng.PortHandle,
               "mi355sw_stage4_stats": eng.Stage4Stats, "mi355sw_stage5_totals": eng.Stage5Totals, "mi355sw_match_result": eng.MatchResult,
               "mi355sw_score_params": eng.ScoreParams}
    lines = []
    for cname, ct in structs.items():
        lines.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for fname, _ in ct._fields_:
            lines.append('printf(" %%zu", offsetof(%s, %s));' % (cname, fname))
        lines.append('printf("\\n");')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mi355sw.h"\nint main(void) {\n' + "\n".join(lines) + "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(graft.ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().splitlines()
    assert len(out) == len(structs)
    for line in out:
        parts = line.split()
        ct = structs[parts[0]]
        assert int(parts[1]) == ctypes.sizeof(ct), parts[0]
        assert [int(x) for x in parts[2:]] == [getattr(ct, f).offset for f, _ in ct._fields_], parts[0]


def test_the_environment_switches_live_in_the_python_front(pkg, monkeypatch):
    """ABI 7: the library reads no environment variable -- engine.env_switches maps the MI355SW_* names onto mi355sw_config
    fields, and the C sources contain no getenv"""
    eng = pkg.engine
    for name in list(eng._ENV_FLAGS) + list(eng._ENV_VERBOSITY) + ["MI355SW_WAIT_S", "MI355SW_FAULT_OVERFLOW_STRIP", "MI355SW_TRACE", "MI355SW_STREAM_PRIO"]:
        monkeypatch.delenv(name, raising=False)
    assert eng.env_switches() == (0, 0, 0.0, 0, 0, None)
    monkeypatch.setenv("MI355SW_TWO_PHASE", "1")
    monkeypatch.setenv("MI355SW_NO_WINDOW", "1")
    monkeypatch.setenv("MI355SW_VERBOSE", "1")
    monkeypatch.setenv("MI355SW_WAIT_S", "2.5")
    monkeypatch.setenv("MI355SW_FAULT_OVERFLOW_STRIP", "0")
    monkeypatch.setenv("MI355SW_TRACE", "/tmp/x.bin")
    flags, verb, wait_s, fault, prio, trace = eng.env_switches()
    assert flags == eng.F_TWO_PHASE | eng.F_NO_WINDOW and verb == eng.V_MESSAGES and wait_s == 2.5 and fault == 1 and trace == "/tmp/x.bin"
    csrc = os.path.join(graft.PKG_DIR, "csrc")
    for fn in os.listdir(csrc):
        if fn.endswith((".cpp", ".hip", ".inc", ".h")):
            assert "getenv" not in open(os.path.join(csrc, fn), errors="replace").read(), fn
    for fn in os.listdir(os.path.join(graft.PKG_DIR, "host")):
        assert "getenv" not in open(os.path.join(graft.PKG_DIR, "host", fn), errors="replace").read(), fn


def test_a_chain_that_ends_below_its_bound_is_refused(pkg):
    from masa_cudalign_amd.bands import check_chain_bound
    check_chain_bound(100, None)
    check_chain_bound(100, 100)
    with pytest.raises(pkg.AlignerError, match="EBOUND"):
        check_chain_bound(99, 100)

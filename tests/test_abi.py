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
    assert lib.mi355sw_abi_version() == 7


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

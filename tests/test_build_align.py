"""CPU: the kernel build's instruction-alignment step (csrc/hipcc_aligned.sh + align8.py, docs/NOTEBOOK_r1-r3.md section 5 (4)).
A small HIP file goes through the same script as the product kernels; the assembled code object must obey the
pass's rule (no run of six or more 8-byte instructions at an address = 4 mod 8) and the host object must still
carry the kernel stub and the device bundle."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "masa-cudalign_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"

SRC = r"""
#include <hip/hip_runtime.h>
typedef short s2 __attribute__((ext_vector_type(2)));
__global__ void probe(int* out, const int* in, int n) {
    int x = in[threadIdx.x], y = in[threadIdx.x + 64], acc = 0;
    for (int k = 0; k < n; k++) {
        s2 a = __builtin_bit_cast(s2, x), b = __builtin_bit_cast(s2, y);
        a = __builtin_elementwise_max(a, b) + b;
        b = __builtin_elementwise_add_sat(b, a);
        x = __builtin_bit_cast(int, a) ^ k; y = __builtin_bit_cast(int, b) + (x >> 3);
        acc += (x & 0xff) > 17 ? x : y;
    }
    out[threadIdx.x] = acc + x + y;
}
"""


def _misaligned_runs(objdump_text):
    runs, cur = [], 0
    for ln in objdump_text.splitlines():
        m = re.match(r"^\s+(\S+)\s.*//\s*([0-9A-Fa-f]+):\s*((?:[0-9A-Fa-f]{8}\s*)+)(?:<.*>)?\s*$", ln)
        if not m:
            continue
        size = 4 * len(m.group(3).split())
        addr = int(m.group(2), 16)
        if size == 8 and addr % 8 == 4:
            cur += 1
        else:
            if cur:
                runs.append(cur)
            cur = 0
    if cur:
        runs.append(cur)
    return runs


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_aligned_build_obeys_its_rule_and_keeps_the_kernel():
    tmp = tempfile.mkdtemp(prefix="align8_")
    try:
        src = os.path.join(tmp, "probe.hip")
        open(src, "w").write(SRC)
        obj = os.path.join(tmp, "probe.o")
        env = dict(os.environ, ALIGN8_KEEP="1")
        p = subprocess.run([os.path.join(CSRC, "hipcc_aligned.sh"), src, obj, "-O3", "-std=c++17", "-fPIC"],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
        assert b"align8:" in p.stdout
        co = os.path.join(tmp, "probe.al", "dev.co")
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], stdout=subprocess.PIPE, check=True).stdout.decode()
        assert "<_Z5probePiPKii>:" in dis
        runs = _misaligned_runs(dis)
        assert all(r < 6 for r in runs), runs
        syms = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-t", obj], stdout=subprocess.PIPE, check=True).stdout.decode()
        assert "probe" in syms            # the host stub of the kernel
        assert "__hip_fatbin" in syms     # the device bundle is embedded
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_align8_refuses_a_mismatching_object(tmp_path):
    """the sizes come from the assembled object: if it does not belong to the assembly the pass must stop"""
    s = tmp_path / "a.s"
    s.write_text("\t.text\nfoo:\n\ts_nop 0\n\tv_mov_b32_e32 v0, v1\n\ts_endpgm\n")
    d = tmp_path / "a.objdump"
    d.write_text("0000000000000000 <foo>:\n\ts_nop 0      // 000000000000: BF800000\n\ts_endpgm      // 000000000004: BF810000\n")
    p = subprocess.run(["python3", os.path.join(CSRC, "align8.py"), str(s), str(d), str(tmp_path / "o.s")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode != 0 and b"align8" in p.stdout


def test_only_wait_states_between_plain_packed_instructions_are_removed():
    """csrc/strip_pk_nops.py: an `s_nop 0` goes only when BOTH neighbours are plain v_pk_* instructions; wait states in
    front of DPP moves, after other instructions, longer ones, and ones next to a label stay"""
    import sys
    tmp = tempfile.mkdtemp(prefix="stripnops_")
    try:
        src, dst = os.path.join(tmp, "in.s"), os.path.join(tmp, "out.s")
        open(src, "w").write("\n".join([
            "\tv_pk_max_i16 v2, v2, v211",
            "\ts_nop 0",                                   # 1: removed
            "\tv_pk_max_i16 v2, v2, v1",
            "\ts_nop 0",                                   # 2: removed (comment lines in between do not matter)
            "\t; a comment",
            "\tv_pk_add_u16 v37, v2, -3 op_sel_hi:[1,0]",
            "\ts_nop 0",                                   # 3: kept, a DPP move follows
            "\tv_mov_b32_dpp v8, v37 wave_shr:1 row_mask:0xf bank_mask:0xf",
            "\tv_pk_add_u16 v3, v3, v4",
            "\ts_nop 1",                                   # 4: kept, two wait states are somebody else's hazard
            "\tv_pk_add_u16 v5, v3, v4",
            "\tv_add_u32_e32 v6, v5, v4",
            "\ts_nop 0",                                   # 5: kept, the producer is not a packed instruction
            "\tv_pk_add_u16 v7, v6, v4",
            "\ts_nop 0",                                   # 6: kept, a label (branch target) follows
            ".LBB0_1:",
            "\tv_pk_add_u16 v9, v7, v4",
            ""]))
        subprocess.check_call([sys.executable, os.path.join(CSRC, "strip_pk_nops.py"), src, dst], stderr=subprocess.DEVNULL)
        out = open(dst).read()
        assert out.count("s_nop 0") == 3 and out.count("s_nop 1") == 1
        assert "v_pk_max_i16 v2, v2, v211\n\tv_pk_max_i16 v2, v2, v1\n\t; a comment\n\tv_pk_add_u16 v37" in out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_library_carries_the_identity_of_its_sources(pkg):
    """mi355sw_build_id() = csrc/build_id.py over the sources in the tree (a stale prebuilt library shows here)"""
    pkg.build_library()
    assert pkg.engine.library_build_id() == pkg.engine.source_build_id()
    assert len(pkg.engine.library_build_id()) == 16

"""Where a strip's lifetime goes in a pruned run: fast-forward runs, skipped chunks the chunk body takes, computed chunks.
Needs a library whose pruning kernels carry the instrumentation of tools/pk16_trace_band.patch (kept out of the tree: it
would change the build id of the measured library):
    git apply tools/pk16_trace_band.patch
    (cd masa-cudalign_amd/csrc && ./hipcc_aligned.sh sw_kernel_pk16_f.hip _var/f_tb.o -O3 -std=c++17 -fPIC -w -mllvm -amdgpu-sched-strategy=max-ilp -DPK16_TRACE_BAND \
      && hipcc --offload-arch=gfx950 -shared $(ls _obj/*.o | grep -v sw_kernel_pk16_f.o) _var/f_tb.o -o ../libvar_tb.so)
    git checkout masa-cudalign_amd/csrc/sw_kernel_pk16.inc
    MI355SW_LIB=$PWD/masa-cudalign_amd/libvar_tb.so MI355SW_TRACE=/tmp/tb.bin python tools/seed_probe.py 16000000 14650000 sw 5 nobase
    python tools/trace_band_split.py /tmp/tb.bin"""
import sys, numpy as np
t = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 4)
S = int((t[:, 0] != 0).sum())
t = t[:S]
life = (t[:, 1] - t[:, 0]) / 1e5                       # ms
ff = (t[:, 2] & 0xffffffff) / 1e5
bs = ((t[:, 2] >> 32) & 0xffffffff) / 1e5
comp = (t[:, 3] & 0xffffffff) / 1e5
nbs = (t[:, 3] >> 32) & 0xffffffff
for lo, hi in ((0, S // 8), (S // 8, S // 2), (S // 2, 7 * S // 8), (7 * S // 8, S)):
    sl = slice(lo, hi)
    print("strips %5d..%5d: lifetime %.0f ms = fast-forward runs %.0f + skipped chunks in the body %.0f (%d of them, %.1f us each) + computed chunks %.0f + rest %.0f" % (
        lo, hi, life[sl].mean(), ff[sl].mean(), bs[sl].mean(), nbs[sl].mean(), 1e3 * bs[sl].sum() / max(1, nbs[sl].sum()), comp[sl].mean(),
        (life[sl] - ff[sl] - bs[sl] - comp[sl]).mean()))
print("all: lifetime %.0f s-wavefront, ff %.0f, body-skip %.0f, computed %.0f" % (life.sum() / 1e3, ff.sum() / 1e3, bs.sum() / 1e3, comp.sum() / 1e3))

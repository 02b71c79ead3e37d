"""Where a strip's lifetime goes in a pruned run: fast-forward runs, skipped chunks the chunk body takes, computed chunks.
Needs a library whose R' = 16 pruning kernels were built with -DPK16_TRACE_BAND (a build of its own: the instrumentation
switches the hot loop off):
    (cd masa-cudalign_amd/csrc && mkdir -p _var && ./hipcc_aligned.sh sw_kernel_pk16_f.hip _var/f_tb.o -O3 -std=c++17 -fPIC -w -mllvm -amdgpu-sched-strategy=max-ilp -DPK16_TRACE_BAND \
      && hipcc --offload-arch=gfx950 -shared $(ls _obj/*.o | grep -v sw_kernel_pk16_f.o) _var/f_tb.o -o ../libvar_tb.so)
    MI355SW_LIB=$PWD/masa-cudalign_amd/libvar_tb.so MI355SW_TRACE=/tmp/tb.bin PROBE_WINDOW_ONLY=1 python tools/window_probe.py 16000000 14650000 32
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

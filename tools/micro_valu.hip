// micro-benchmark: int32 VALU issue rate on gfx950 for the op mix of the SW cell (1/2/4 waves per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(64) k(int* out, int iters, long long* cycles) {
    int lane = threadIdx.x;
    int a = lane, b = lane * 3, c = lane ^ 5, d = 7 - lane, e = lane + 11, f = lane - 3, g = 2 * lane, h = 9;
    int p = 0x22262222 + lane;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (MODE == 0) {          // add/max dependent chains x8 independent
                a = max(a + -2, b); b = max(b + -2, c); c = max(c + -2, d); d = max(d + -2, e);
                e = max(e + -2, f); f = max(f + -2, g); g = max(g + -2, h); h = max(h + -2, a);
            } else if (MODE == 1) {   // max3
                a = max(max(a, b), c) - 1; b = max(max(b, c), d) - 1; c = max(max(c, d), e) - 1; d = max(max(d, e), f) - 1;
                e = max(max(e, f), g) - 1; f = max(max(f, g), h) - 1; g = max(max(g, h), a) - 1; h = max(max(h, a), b) - 1;
            } else if (MODE == 2) {   // bfe + add
                a += __builtin_amdgcn_sbfe(p, b & 28, 4); b += __builtin_amdgcn_sbfe(p, c & 28, 4);
                c += __builtin_amdgcn_sbfe(p, d & 28, 4); d += __builtin_amdgcn_sbfe(p, e & 28, 4);
                e += __builtin_amdgcn_sbfe(p, f & 28, 4); f += __builtin_amdgcn_sbfe(p, g & 28, 4);
                g += __builtin_amdgcn_sbfe(p, h & 28, 4); h += __builtin_amdgcn_sbfe(p, a & 28, 4);
            } else if (MODE == 3) {   // dpp wave_shr + add
                a = __builtin_amdgcn_update_dpp(a, b, 0x138, 0xf, 0xf, false) + 1;
                b = __builtin_amdgcn_update_dpp(b, c, 0x138, 0xf, 0xf, false) + 1;
                c = __builtin_amdgcn_update_dpp(c, d, 0x138, 0xf, 0xf, false) + 1;
                d = __builtin_amdgcn_update_dpp(d, a, 0x138, 0xf, 0xf, false) + 1;
            } else if (MODE == 4) {   // dpp row_shr + add
                a = __builtin_amdgcn_update_dpp(a, b, 0x111, 0xf, 0xf, false) + 1;
                b = __builtin_amdgcn_update_dpp(b, c, 0x111, 0xf, 0xf, false) + 1;
                c = __builtin_amdgcn_update_dpp(c, d, 0x111, 0xf, 0xf, false) + 1;
                d = __builtin_amdgcn_update_dpp(d, a, 0x111, 0xf, 0xf, false) + 1;
            } else if (MODE == 6) {   // SERIAL dependent chain, packed i16 (one chain)
                typedef short s2 __attribute__((ext_vector_type(2)));
                s2 x = __builtin_bit_cast(s2, a), y = __builtin_bit_cast(s2, b);
                s2 m2 = {-2, -2};
                x = __builtin_elementwise_max(x + m2, y); x = __builtin_elementwise_max(x + m2, y);
                x = __builtin_elementwise_max(x + m2, y); x = __builtin_elementwise_max(x + m2, y);
                a = __builtin_bit_cast(int, x);
            } else if (MODE == 7) {   // SERIAL dependent chain, int32
                a = max(a + -2, b); a = max(a + -2, b); a = max(a + -2, b); a = max(a + -2, b);
            } else if (MODE == 8) {   // two interleaved dependent chains, packed
                typedef short s2 __attribute__((ext_vector_type(2)));
                s2 x = __builtin_bit_cast(s2, a), y = __builtin_bit_cast(s2, b), z = __builtin_bit_cast(s2, c);
                s2 m2 = {-2, -2};
                x = __builtin_elementwise_max(x + m2, y); z = __builtin_elementwise_max(z + m2, y);
                x = __builtin_elementwise_max(x + m2, y); z = __builtin_elementwise_max(z + m2, y);
                a = __builtin_bit_cast(int, x); c = __builtin_bit_cast(int, z);
            } else if (MODE == 5) {   // packed i16 max/add
                typedef short s2 __attribute__((ext_vector_type(2)));
                s2 x = __builtin_bit_cast(s2, a), y = __builtin_bit_cast(s2, b), z = __builtin_bit_cast(s2, c), w = __builtin_bit_cast(s2, d);
                s2 m2 = {-2, -2};
                x = __builtin_elementwise_max(x + m2, y); y = __builtin_elementwise_max(y + m2, z);
                z = __builtin_elementwise_max(z + m2, w); w = __builtin_elementwise_max(w + m2, x);
                a = __builtin_bit_cast(int, x); b = __builtin_bit_cast(int, y); c = __builtin_bit_cast(int, z); d = __builtin_bit_cast(int, w);
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = a + b + c + d + e + f + g + h;
    if (lane == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int MODE>
void run(const char* name, int ops_per_unroll) {
    int* d; long long* cyc;
    CHECK(hipMalloc(&d, 256 * 32 * 64 * 4));
    CHECK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        int grid = 256 * 4 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, 100, cyc);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters, cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        double insts = (double) iters * 16 * ops_per_unroll;     // per wave
        printf("%-12s waves/SIMD=%d  %.3f ms  memtime-cycles/inst=%.2f  ns/inst/wave=%.3f  wave-inst/ns/SIMD=%.3f  (memtime MHz ~ %.0f)\n",
               name, wps, ms, (double) c / insts, ms * 1e6 / insts, insts * wps / (ms * 1e6), (double) c / (ms * 1e3));
    }
}
int main() {
    run<0>("add+max", 16);
    run<1>("max3+sub", 16);
    run<2>("bfe+add(+and)", 24);
    run<3>("dpp_wshr+add", 8);
    run<4>("dpp_rshr+add", 8);
    run<5>("pk_i16", 8);
    run<6>("pk_serial", 8);
    run<7>("i32_serial", 8);
    run<8>("pk_2chains", 8);
    return 0;
}

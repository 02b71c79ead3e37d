// micro-benchmark: what an 8-byte instruction at an address = 4 (mod 8) costs on gfx950 with one wavefront per
// SIMD, and what an s_nop / s_waitcnt / widened VOP2 costs (the inputs of csrc/align8.py's padding rule).
// hipcc --offload-arch=gfx950 -O3 micro_fetch.hip -o micro_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define P8 "v_pk_max_i16 %0, %0, %4\n v_pk_max_i16 %1, %1, %4\n v_pk_max_i16 %2, %2, %4\n v_pk_max_i16 %3, %3, %4\n" \
           "v_pk_max_i16 %0, %0, %4\n v_pk_max_i16 %1, %1, %4\n v_pk_max_i16 %2, %2, %4\n v_pk_max_i16 %3, %3, %4\n"
#define P64 P8 P8 P8 P8 P8 P8 P8 P8
#define OPS : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(lane)

template <int MODE>
__global__ void __launch_bounds__(64) k(int* out, int iters) {
    const int lane = threadIdx.x;
    int a = lane, b = lane * 3, c = lane ^ 5, d = 7 - lane;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) asm volatile(".p2align 3\n" P64 P64 OPS);                                 // 128 aligned
        if (MODE == 1) asm volatile(".p2align 3\n s_nop 0\n" P64 P64 "s_nop 0\n" OPS);           // 128 misaligned + 2 nops
        if (MODE == 2) asm volatile(".p2align 3\n s_nop 0\n s_nop 0\n" P64 P64 OPS);             // 128 aligned + 2 nops
        if (MODE == 3) asm volatile(".p2align 3\n" P8 "s_nop 0\n s_nop 0\n" P8 "s_nop 0\n s_nop 0\n" P8 "s_nop 0\n s_nop 0\n" P8 "s_nop 0\n s_nop 0\n"
                                    P8 "s_nop 0\n s_nop 0\n" P8 "s_nop 0\n s_nop 0\n" P8 "s_nop 0\n s_nop 0\n" P8 "s_nop 0\n s_nop 0\n"
                                    P64 OPS);                                                    // 128 aligned + 16 nops
        if (MODE == 4) asm volatile(".p2align 3\n" P8 "s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n" P8 "s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n" P8 "s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n" P8 "s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n"
                                    P8 "s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n" P8 "s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n" P8 "s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n" P8 "s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n"
                                    P64 OPS);                                                    // 128 aligned + 16 waitcnt
        if (MODE == 55) asm volatile(".p2align 3\n" P8 "s_or_b32 s20, s20, 1\n s_or_b32 s21, s21, 1\n" P8 "s_or_b32 s20, s20, 1\n s_or_b32 s21, s21, 1\n" P8 "s_or_b32 s20, s20, 1\n s_or_b32 s21, s21, 1\n" P8 "s_or_b32 s20, s20, 1\n s_or_b32 s21, s21, 1\n"
                                    P8 "s_or_b32 s20, s20, 1\n s_or_b32 s21, s21, 1\n" P8 "s_or_b32 s20, s20, 1\n s_or_b32 s21, s21, 1\n" P8 "s_or_b32 s20, s20, 1\n s_or_b32 s21, s21, 1\n" P8 "s_or_b32 s20, s20, 1\n s_or_b32 s21, s21, 1\n"
                                    P64 OPS : "s20", "s21");                                     // 128 aligned + 16 SALU
        if (MODE == 6) asm volatile(".p2align 3\n" P8 "v_mov_b32_e32 v100, v101\n v_mov_b32_e32 v102, v101\n" P8 "v_mov_b32_e32 v100, v101\n v_mov_b32_e32 v102, v101\n" P8 "v_mov_b32_e32 v100, v101\n v_mov_b32_e32 v102, v101\n" P8 "v_mov_b32_e32 v100, v101\n v_mov_b32_e32 v102, v101\n"
                                    P8 "v_mov_b32_e32 v100, v101\n v_mov_b32_e32 v102, v101\n" P8 "v_mov_b32_e32 v100, v101\n v_mov_b32_e32 v102, v101\n" P8 "v_mov_b32_e32 v100, v101\n v_mov_b32_e32 v102, v101\n" P8 "v_mov_b32_e32 v100, v101\n v_mov_b32_e32 v102, v101\n"
                                    P64 OPS : "v100", "v101", "v102");                           // 128 + 16 VOP1 e32
        if (MODE == 7) asm volatile(".p2align 3\n" P8 "v_mov_b32_e64 v100, v101\n v_mov_b32_e64 v102, v101\n" P8 "v_mov_b32_e64 v100, v101\n v_mov_b32_e64 v102, v101\n" P8 "v_mov_b32_e64 v100, v101\n v_mov_b32_e64 v102, v101\n" P8 "v_mov_b32_e64 v100, v101\n v_mov_b32_e64 v102, v101\n"
                                    P8 "v_mov_b32_e64 v100, v101\n v_mov_b32_e64 v102, v101\n" P8 "v_mov_b32_e64 v100, v101\n v_mov_b32_e64 v102, v101\n" P8 "v_mov_b32_e64 v100, v101\n v_mov_b32_e64 v102, v101\n" P8 "v_mov_b32_e64 v100, v101\n v_mov_b32_e64 v102, v101\n"
                                    P64 OPS : "v100", "v101", "v102");                           // 128 + 16 VOP1 e64
        if (MODE == 8) asm volatile(".p2align 3\n s_nop 0\n" P8 "s_nop 0\n s_nop 0\n" P8 "s_nop 0\n s_nop 0\n" P8 "s_nop 0\n s_nop 0\n" P8 "s_nop 0\n s_nop 0\n"
                                    P8 "s_nop 0\n s_nop 0\n" P8 "s_nop 0\n s_nop 0\n" P8 "s_nop 0\n s_nop 0\n" P8 "s_nop 0\n s_nop 0\n"
                                    P64 "s_nop 0\n" OPS);                                        // 128 MISaligned + 18 nops
    }
    out[blockIdx.x * 64 + lane] = a + b + c + d;
}

template <int MODE>
double run(const char* name, int wps = 1) {
    int* d;
    CHECK(hipMalloc(&d, 256 * 32 * 64 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 200000;
    const int grid = 256 * 4 * wps;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters);   // warm-up at full length: clocks settle
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = ms * 1e6 / iters;
    printf("%-44s wps=%d  %.2f ms  ns/iteration=%.2f\n", name, wps, ms, ns);
    CHECK(hipFree(d));
    return ns;
}
int main() {
    // the clock ramps up during the first seconds: run the whole series three times and read the last one
    for (int rep = 0; rep < 3; rep++) {
        printf("---- pass %d\n", rep);
        const double base = run<0>("128 pk_max aligned");
        const double mis = run<1>("128 pk_max at +4 (+2 nops)");
        const double n2 = run<2>("128 aligned + 2 nops");
        const double n16 = run<3>("128 aligned + 16 nops (8 pairs)");
        const double w16 = run<4>("128 aligned + 16 s_waitcnt");
        const double v16 = run<6>("128 aligned + 16 v_mov_e32");
        const double x16 = run<7>("128 aligned + 16 v_mov_e64");
        const double m16 = run<8>("128 at +4 + 18 nops");
        const double base2 = run<0>("128 pk_max aligned (again)");
        printf("per aligned pk op        %.3f ns\n", base / 128);
        printf("per misaligned pk op     +%.3f ns\n", (mis - n2) / 128);
        printf("per s_nop 0              %.3f ns\n", (n16 - base) / 16);
        printf("per s_waitcnt            %.3f ns\n", (w16 - base) / 16);
        printf("per v_mov_b32_e32        %.3f ns\n", (v16 - base) / 16);
        printf("per v_mov_b32_e64        %.3f ns\n", (x16 - base) / 16);
        printf("misaligned with nop pairs vs aligned with nop pairs: %.3f ns per pk op\n", (m16 - n16) / 128);
        run<0>("128 pk_max aligned", 2);
        run<1>("128 pk_max at +4 (+2 nops)", 2);
    }
    return 0;
}

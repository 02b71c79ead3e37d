// micro-benchmark: dependent-issue and LDS latencies seen by ONE wavefront per SIMD on gfx950
// (the strip kernels' operating point).  hipcc --offload-arch=gfx950 -O3 micro_lat.hip -o micro_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// MODE 0: 8 independent v_pk_max_i16 chains      MODE 1: one serial chain of v_pk_max_i16
// MODE 2: serial chain alternating v_pk_max_i16 / v_pk_add_i16 clamp
// MODE 3: serial v_perm_b32                        MODE 4: serial dpp wave_shr:1 + pk_max
// MODE 5: ds_read_b64 pointer chase (pure LDS latency)
// MODE 10+k: ds_read_b64 issued, k*8 independent VALU ops, then wait + use  (latency hiding curve)
template <int MODE>
__global__ void __launch_bounds__(64) k(int* out, int iters) {
    __shared__ int2 lds[256];
    const int lane = threadIdx.x;
    lds[lane] = make_int2((lane * 8) & 2040, lane);
    lds[64 + lane] = make_int2((lane * 8) & 2040, lane);
    __syncthreads();
    int a = lane, b = lane * 3, c = lane ^ 5, d = 7 - lane, e = lane + 11, f = lane - 3, g = 2 * lane, h = 9;
    int addr = lane * 8;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 16; u++)
                asm volatile("v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n"
                             "v_pk_max_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8\n"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(lane));
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 16; u++)
                asm volatile("v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1\n"
                             "v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1\n"
                             : "+v"(a) : "v"(lane));
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 16; u++)
                asm volatile("v_pk_max_i16 %0, %0, %1\n v_pk_add_i16 %0, %0, -2 op_sel_hi:[1,0] clamp\n v_pk_max_i16 %0, %0, %1\n v_pk_add_i16 %0, %0, -2 op_sel_hi:[1,0] clamp\n"
                             "v_pk_max_i16 %0, %0, %1\n v_pk_add_i16 %0, %0, -2 op_sel_hi:[1,0] clamp\n v_pk_max_i16 %0, %0, %1\n v_pk_add_i16 %0, %0, -2 op_sel_hi:[1,0] clamp\n"
                             : "+v"(a) : "v"(lane));
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 16; u++)
                asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %0, %0, %1, %2\n"
                             "v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %0, %0, %1, %2\n"
                             : "+v"(a) : "v"(lane), "v"(0x03020100));
        } else if (MODE == 4) {
#pragma unroll
            for (int u = 0; u < 16; u++)
                asm volatile("s_nop 1\n v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_pk_max_i16 %0, %0, %1\n"
                             "s_nop 1\n v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_pk_max_i16 %0, %0, %1\n"
                             "s_nop 1\n v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_pk_max_i16 %0, %0, %1\n"
                             "s_nop 1\n v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_pk_max_i16 %0, %0, %1\n"
                             : "+v"(a), "+v"(b));
        } else if (MODE == 5) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                int2 r;
                asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)\n" : "=v"(r) : "v"(addr) : "memory");
                addr = r.x;
                b += r.y;
            }
        } else if (MODE >= 10) {
            constexpr int K = MODE - 10;
#pragma unroll
            for (int u = 0; u < 16; u++) {
                int2 r;
                asm volatile("ds_read_b64 %0, %1\n" : "=v"(r) : "v"(addr) : "memory");
#pragma unroll
                for (int q = 0; q < K; q++)
                    asm volatile("v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n"
                                 "v_pk_max_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8\n"
                                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(lane));
                asm volatile("s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, %1" : "+v"(c) : "v"(r.y) : "memory");
            }
        }
    }
    out[blockIdx.x * 64 + lane] = a + b + c + d + e + f + g + h + addr;
}

static double clock_ghz = 2.4;
template <int MODE>
void run(const char* name, double ops_per_iter, int wps = 1) {
    int* d;
    CHECK(hipMalloc(&d, 256 * 32 * 64 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    const int grid = 256 * 4 * wps;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, 2000);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ns_per_iter = ms * 1e6 / iters / 16;
    printf("%-28s wps=%d  %.3f ms  ns/unit=%.2f  ns/op=%.3f  (~%.1f cycles/op at %.2f GHz)\n", name, wps, ms, ns_per_iter,
           ns_per_iter / ops_per_iter, ns_per_iter / ops_per_iter * clock_ghz, clock_ghz);
    CHECK(hipFree(d));
}
int main() {
    run<0>("pk_max x8 independent", 8);
    run<1>("pk_max serial", 8);
    run<2>("pk_max/pk_add_sat serial", 8);
    run<3>("v_perm serial", 8);
    run<4>("dpp+pk_max serial (2 ops)", 8);
    run<5>("ds_read_b64 chase", 1);
    run<10>("ds_read + 0 valu + wait", 1);
    run<11>("ds_read + 8 valu + wait", 1);
    run<12>("ds_read + 16 valu + wait", 1);
    run<13>("ds_read + 24 valu + wait", 1);
    run<14>("ds_read + 32 valu + wait", 1);
    run<16>("ds_read + 48 valu + wait", 1);
    run<18>("ds_read + 64 valu + wait", 1);
    run<0>("pk_max x8 independent", 8, 2);
    run<1>("pk_max serial", 8, 2);
    run<5>("ds_read_b64 chase", 1, 2);
    return 0;
}

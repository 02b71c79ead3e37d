// What do the pieces of the strip kernel's chunk boundary cost a LONE wavefront on gfx950 (one per SIMD, nothing to hide
// latency behind)?  Each case times a long run of one construct with s_memrealtime (100 MHz) and prints ns per item:
//   valu        a dependent-free v_pk_add_u16 (the unit everything else is measured in)
//   taken       s_branch to the next instruction group (taken, forward, short)
//   cond_taken  s_cbranch_scc1 taken after an s_cmp
//   not_taken   s_cbranch_scc0 not taken after an s_cmp
//   loop16      backward branch closing a body of 16 / 256 / 2048 VALU instructions
//   readlane    v_readlane_b32 into an SGPR that the next SALU instruction uses
//   sload       s_load_dword (scalar cache hit) + s_waitcnt lgkmcnt(0)
//   lds         ds_read_b64 + s_waitcnt lgkmcnt(0)
//   glob_l2     global_load_dword of one hot line (sc1) + s_waitcnt vmcnt(0)
//   setpc       the far-jump sequence the compiler emits beyond the short branch range (s_getpc, add, addc, s_setpc)
// hipcc --offload-arch=gfx950 -O3 tools/micro_branch.hip -o tools/_micro_branch && tools/_micro_branch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define R4(X) X X X X
#define R16(X) R4(X) R4(X) R4(X) R4(X)
#define R64(X) R16(X) R16(X) R16(X) R16(X)
#define R256(X) R64(X) R64(X) R64(X) R64(X)

#define TIMED(NAME, ITEMS, BODY) \
    __global__ void __launch_bounds__(64) k_##NAME(long long* out, const int* g, int iters) { \
        __shared__ int lds[256]; \
        lds[threadIdx.x] = threadIdx.x; \
        __syncthreads(); \
        unsigned x = threadIdx.x, y = 3; int sv = 0; \
        const int* gp = g + (threadIdx.x & 15); \
        const unsigned lp = (unsigned) (threadIdx.x * 4); \
        int priv[4]; priv[threadIdx.x & 3] = 1; \
        const long long t0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < iters; it++) { BODY } \
        const long long t1 = __builtin_amdgcn_s_memrealtime(); \
        if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (long long) (ITEMS) * iters; } \
        if (x == 0x12345 && y == 77 && sv == 99) out[1000] = priv[x & 3]; \
    }

TIMED(valu, 256, asm volatile(R256("v_pk_add_u16 %0, %0, %1\n") : "+v"(x) : "v"(y));)
TIMED(taken, 64, asm volatile(R64("s_branch 1f\n s_nop 0\n1:\n v_pk_add_u16 %0, %0, %1\n") : "+v"(x) : "v"(y));)
TIMED(cond_taken, 64, asm volatile(R64("s_cmp_eq_u32 0, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:\n v_pk_add_u16 %0, %0, %1\n") : "+v"(x) : "v"(y) : "scc");)
TIMED(not_taken, 64, asm volatile(R64("s_cmp_eq_u32 0, 0\n s_cbranch_scc0 1f\n s_nop 0\n1:\n v_pk_add_u16 %0, %0, %1\n") : "+v"(x) : "v"(y) : "scc");)
// the "item" of the loop cases is one trip: body + backward branch
TIMED(loop16, 64, asm volatile("s_mov_b32 s20, 64\n2:\n" R16("v_pk_add_u16 %0, %0, %1\n") "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 2b\n" : "+v"(x) : "v"(y) : "s20", "scc");)
TIMED(loop256, 16, asm volatile("s_mov_b32 s20, 16\n2:\n" R256("v_pk_add_u16 %0, %0, %1\n") "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 2b\n" : "+v"(x) : "v"(y) : "s20", "scc");)
TIMED(loop2048, 4, asm volatile("s_mov_b32 s20, 4\n2:\n" R256("v_pk_add_u16 %0, %0, %1\n") R256("v_pk_add_u16 %0, %0, %1\n") R256("v_pk_add_u16 %0, %0, %1\n") R256("v_pk_add_u16 %0, %0, %1\n")
                             R256("v_pk_add_u16 %0, %0, %1\n") R256("v_pk_add_u16 %0, %0, %1\n") R256("v_pk_add_u16 %0, %0, %1\n") R256("v_pk_add_u16 %0, %0, %1\n")
                             "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 2b\n" : "+v"(x) : "v"(y) : "s20", "scc");)
TIMED(readlane, 64, asm volatile(R64("v_readlane_b32 s20, %0, 5\n s_add_u32 s21, s20, 1\n") : : "v"(x) : "s20", "s21", "scc");)
TIMED(sload, 64, asm volatile(R64("s_load_dword s20, %0, 0x0\n s_waitcnt lgkmcnt(0)\n") : : "s"(g) : "s20", "memory");)
TIMED(lds, 64, asm volatile(R64("ds_read_b64 v[20:21], %0\n s_waitcnt lgkmcnt(0)\n") : : "v"(lp) : "v20", "v21", "memory");)
TIMED(glob_l2, 64, asm volatile(R64("global_load_dword v20, %0, off sc1\n s_waitcnt vmcnt(0)\n") : : "v"(gp) : "v20", "memory");)
TIMED(setpc, 64, asm volatile(R64("s_getpc_b64 s[20:21]\n s_add_u32 s20, s20, 16\n s_addc_u32 s21, s21, 0\n s_setpc_b64 s[20:21]\n s_nop 0\n v_pk_add_u16 %0, %0, %1\n") : "+v"(x) : "v"(y) : "s20", "s21", "scc");)

template <typename K>
static void run(const char* name, K kernel, int blocks) {
    long long* out; int* g;
    CHECK(hipMalloc(&out, 4096 * 8)); CHECK(hipMalloc(&g, 4096));
    CHECK(hipMemset(g, 0, 4096));
    long long h[2048 * 2];
    double best = 1e30;
    for (int rep = 0; rep < 5; rep++) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), 0, 0, out, g, 200);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h, out, blocks * 16, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int b = 0; b < blocks; b++) { const double ns = h[2 * b] * 10.0 / (double) h[2 * b + 1]; if (ns > worst) worst = ns; }
        if (worst < best) best = worst;
    }
    printf("%-11s %4d wavefronts: %8.2f ns per item (slowest wavefront, best of 5)\n", name, blocks, best);
    CHECK(hipFree(out)); CHECK(hipFree(g));
}

int main() {
    for (int blocks : {1, 1024}) {         // alone on the chip / one wavefront on every SIMD (256 CUs x 4)
        run("valu", k_valu, blocks);
        run("taken", k_taken, blocks);
        run("cond_taken", k_cond_taken, blocks);
        run("not_taken", k_not_taken, blocks);
        run("loop16", k_loop16, blocks);
        run("loop256", k_loop256, blocks);
        run("loop2048", k_loop2048, blocks);
        run("readlane", k_readlane, blocks);
        run("sload", k_sload, blocks);
        run("lds", k_lds, blocks);
        run("glob_l2", k_glob_l2, blocks);
        run("setpc", k_setpc, blocks);
    }
    return 0;
}

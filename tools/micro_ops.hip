// micro-benchmark: issue cost of the packed kernel's VALU ops for different operand/bank patterns, one wavefront
// per SIMD.  hipcc --offload-arch=gfx950 -O3 micro_ops.hip -o micro_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define R4(X) X X X X
#define R16(X) R4(X) R4(X) R4(X) R4(X)
#define R32(X) R16(X) R16(X)
#define R64(X) R32(X) R32(X)
#define R128(X) R16(X) R16(X) R16(X) R16(X) R16(X) R16(X) R16(X) R16(X)
#define CLOB : : : "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29"
#define CASE(N, BODY) if (MODE == N) asm volatile(".p2align 3\n" R128(BODY) CLOB);

template <int MODE>
__global__ void __launch_bounds__(64) k(int* out, int iters) {
    for (int it = 0; it < iters; it++) {
        // four independent destinations per group of 4 so that no op depends on the previous one
        CASE(0, "v_pk_max_i16 v10, v11, v12\n")                      // srcs in banks 3,0
        CASE(1, "v_pk_max_i16 v10, v11, v15\n")                      // srcs in the same bank (3,3)
        CASE(2, "v_pk_max_i16 v10, v10, v12\n")                      // in place (dependent chain)
        CASE(3, "v_max_i32_e64 v10, v11, v12\n")                     // VOP3 int32
        CASE(4, "v_max_i32_e32 v10, v11, v12\n")                     // VOP2 int32 (4 bytes)
        CASE(5, "v_pk_add_u16 v10, v11, v12\n")
        CASE(6, "v_pk_add_i16 v10, v11, -2 op_sel_hi:[1,0] clamp\n")
        CASE(7, "v_perm_b32 v10, v11, v12, v13\n")                   // 3 sources, banks 3,0,1
        CASE(8, "v_perm_b32 v10, v11, v15, v19\n")                   // 3 sources, all bank 3
        CASE(9, "v_pk_max_i16 v10, v11, v12\n v_pk_max_i16 v14, v15, v16\n")   // alternating (2 ops per body)
        CASE(10, "v_pk_max_i16 v10, v11, v13\n")                     // banks 3,1
        CASE(11, "v_pk_max_i16 v10, v12, v14\n")                     // banks 0,2
        CASE(12, "v_pk_max_i16 v11, v12, v13\n")                     // dst bank 3, srcs 0,1
        CASE(13, "v_and_b32_e64 v10, v11, v12\n")
        CASE(14, "v_alignbit_b32 v10, v11, v12, 16\n")
        if (MODE == 20) asm volatile(".p2align 3\n" R32("v_pk_max_i16 v10, v24, v25\n v_pk_max_i16 v11, v24, v25\n v_pk_max_i16 v12, v24, v25\n v_pk_max_i16 v13, v24, v25\n") CLOB);
        if (MODE == 21) asm volatile(".p2align 3\n" R32("v_pk_max_i16 v10, v24, v25\n v_pk_max_i16 v14, v24, v25\n v_pk_max_i16 v18, v24, v25\n v_pk_max_i16 v22, v24, v25\n") CLOB);
        if (MODE == 22) asm volatile(".p2align 3\n" R64("v_pk_max_i16 v10, v24, v25\n v_pk_max_i16 v11, v24, v25\n") CLOB);
        if (MODE == 23) asm volatile(".p2align 3\n" R32("v_pk_max_i16 v10, v24, v25\n v_pk_max_i16 v11, v26, v27\n v_pk_max_i16 v12, v28, v29\n v_pk_max_i16 v13, v16, v17\n") CLOB);
        if (MODE == 24) asm volatile(".p2align 3\n" R32("v_pk_max_i16 v10, v10, v25\n v_pk_max_i16 v11, v11, v25\n v_pk_max_i16 v12, v12, v25\n v_pk_max_i16 v13, v13, v25\n") CLOB);
        if (MODE == 25) asm volatile(".p2align 3\n" R16("v_pk_max_i16 v10, v10, v25\n v_pk_max_i16 v11, v11, v25\n v_pk_max_i16 v12, v12, v25\n v_pk_max_i16 v13, v13, v25\n v_pk_max_i16 v14, v14, v25\n v_pk_max_i16 v15, v15, v25\n v_pk_max_i16 v16, v16, v25\n v_pk_max_i16 v17, v17, v25\n") CLOB);
        if (MODE == 26) asm volatile(".p2align 3\n" R16("v_pk_max_i16 v10, v10, v25\n v_pk_add_u16 v11, v11, v25\n v_perm_b32 v12, v12, v25, v26\n v_pk_max_i16 v13, v13, v25\n v_pk_add_i16 v14, v14, -2 op_sel_hi:[1,0] clamp\n v_pk_max_i16 v15, v15, v25\n v_pk_max_i16 v16, v16, v25\n v_pk_add_u16 v17, v17, v25\n") CLOB);
        if (MODE == 27) asm volatile(".p2align 3\n" R32("v_max_i32_e32 v10, v24, v25\n v_max_i32_e32 v11, v24, v25\n v_max_i32_e32 v12, v24, v25\n v_max_i32_e32 v13, v24, v25\n") CLOB);
        CASE(15, "v_pk_max_i16 v10, v11, v12\n v_pk_max_i16 v13, v10, v12\n")   // RAW at distance 1
    }
    if (iters < 0) out[threadIdx.x] = 1;
}
template <int MODE>
void run(const char* name, int ops) {
    int* d; CHECK(hipMalloc(&d, 1024));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 100000;
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(64), 0, 0, d, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(64), 0, 0, d, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-52s %.3f ns/op\n", name, ms * 1e6 / iters / ops);
    CHECK(hipFree(d));
}
int main() {
    for (int rep = 0; rep < 2; rep++) {
        printf("---- pass %d\n", rep);
        run<4>("v_max_i32_e32 (4 B)", 128);
        run<3>("v_max_i32_e64", 128);
        run<13>("v_and_b32_e64", 128);
        run<0>("v_pk_max_i16 srcs banks 3,0", 128);
        run<10>("v_pk_max_i16 srcs banks 3,1", 128);
        run<11>("v_pk_max_i16 srcs banks 0,2", 128);
        run<12>("v_pk_max_i16 dst bank 3 srcs 0,1", 128);
        run<1>("v_pk_max_i16 srcs same bank", 128);
        run<2>("v_pk_max_i16 in place (serial)", 128);
        run<15>("v_pk_max_i16 RAW distance 1", 256);
        run<9>("v_pk_max_i16 alternating", 256);
        run<20>("pk_max 4 dsts, 4 banks", 128);
        run<21>("pk_max 4 dsts, same bank", 128);
        run<22>("pk_max 2 dsts, 2 banks", 128);
        run<23>("pk_max 4 dsts 4 banks, distinct srcs", 128);
        run<24>("pk_max 4 in-place chains", 128);
        run<25>("pk_max 8 in-place chains", 128);
        run<26>("mixed 8 in-place chains", 128);
        run<27>("v_max_i32_e32 4 dsts (4 B)", 128);
        run<5>("v_pk_add_u16", 128);
        run<6>("v_pk_add_i16 const clamp", 128);
        run<7>("v_perm_b32 banks 3,0,1", 128);
        run<8>("v_perm_b32 all bank 3", 128);
        run<14>("v_alignbit_b32", 128);
    }
    return 0;
}

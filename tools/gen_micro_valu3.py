#!/usr/bin/env python3
"""Writes tools/micro_valu3.hip: which gfx950 VALU instructions issue at 2 cycles per wave64 instruction once a SIMD
has two wavefronts to choose from, which stay at 4, and whether a 4-cycle instruction of one wavefront overlaps with
2-cycle instructions of another.  Placement is controlled as in micro_valu2 (one workgroup per
CU through a 96 KiB LDS allocation, W wavefronts on each SIMD, checked through HW_REG_HW_ID)."""
import os

# name -> instruction text with {r} = the read-write register, %8/%9 = read-only operands
OPS = [
    ("v_add_u32", "v_add_u32 {r}, {r}, %8"),
    ("v_sub_u32", "v_sub_u32 {r}, {r}, %8"),
    ("v_and_b32", "v_and_b32 {r}, {r}, %8"),
    ("v_or_b32", "v_or_b32 {r}, {r}, %8"),
    ("v_xor_b32", "v_xor_b32 {r}, {r}, %8"),
    ("v_mov_b32", "v_mov_b32 {r}, %8"),
    ("v_lshlrev_b32", "v_lshlrev_b32 {r}, 1, {r}"),
    ("v_ashrrev_i32", "v_ashrrev_i32 {r}, 1, {r}"),
    ("v_max_i32", "v_max_i32 {r}, {r}, %8"),
    ("v_min_i32", "v_min_i32 {r}, {r}, %8"),
    ("v_max_u32", "v_max_u32 {r}, {r}, %8"),
    ("v_max_f32", "v_max_f32 {r}, {r}, %8"),
    ("v_min_f32", "v_min_f32 {r}, {r}, %8"),
    ("v_add_f32", "v_add_f32 {r}, {r}, %8"),
    ("v_sub_f32", "v_sub_f32 {r}, {r}, %8"),
    ("v_mul_f32", "v_mul_f32 {r}, {r}, %8"),
    ("v_fma_f32", "v_fma_f32 {r}, {r}, %8, %9"),
    ("v_fmac_f32", "v_fmac_f32 {r}, %8, %9"),
    ("v_max3_f32", "v_max3_f32 {r}, {r}, %8, %9"),
    ("v_max3_i32", "v_max3_i32 {r}, {r}, %8, %9"),
    ("v_cndmask_b32", "v_cndmask_b32 {r}, {r}, %8, vcc"),
    ("v_cmp_gt_i32", "v_cmp_gt_i32 vcc, {r}, %8"),
    ("v_add_co_u32", "v_add_co_u32 {r}, vcc, {r}, %8"),
    ("v_mul_i32_i24", "v_mul_i32_i24 {r}, {r}, %8"),
    ("v_mad_i32_i24", "v_mad_i32_i24 {r}, {r}, %8, %9"),
    ("v_mul_lo_u32", "v_mul_lo_u32 {r}, {r}, %8"),
    ("v_bfe_i32", "v_bfe_i32 {r}, {r}, 4, 8"),
    ("v_max_i16", "v_max_i16 {r}, {r}, %8"),
    ("v_add_u16", "v_add_u16 {r}, {r}, %8"),
    ("v_max_f16", "v_max_f16 {r}, {r}, %8"),
    ("v_add_f16", "v_add_f16 {r}, {r}, %8"),
    ("v_pk_max_i16", "v_pk_max_i16 {r}, {r}, %8"),
    ("v_pk_add_i16", "v_pk_add_i16 {r}, {r}, %8"),
    ("v_pk_sub_i16_clamp", "v_pk_sub_i16 {r}, {r}, %8 clamp"),
    ("v_pk_add_u16", "v_pk_add_u16 {r}, {r}, %8"),
    ("v_pk_min_u16", "v_pk_min_u16 {r}, {r}, %8"),
    ("v_pk_max_f16", "v_pk_max_f16 {r}, {r}, %8"),
    ("v_pk_add_f16", "v_pk_add_f16 {r}, {r}, %8"),
    ("v_pk_fma_f16", "v_pk_fma_f16 {r}, {r}, %8, %9"),
    ("v_perm_b32", "v_perm_b32 {r}, {r}, %8, %9"),
    ("v_alignbit_b32", "v_alignbit_b32 {r}, {r}, %8, 16"),
    ("v_med3_i32", "v_med3_i32 {r}, {r}, %8, %9"),
    ("v_add_u32_e64", "v_add_u32_e64 {r}, {r}, %8"),
    ("v_mov_b32_e64", "v_mov_b32_e64 {r}, %8"),
    ("v_max_i32_sdwa", "v_max_i32_sdwa {r}, {r}, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0"),
    ("v_add_u32_sdwa", "v_add_u32_sdwa {r}, {r}, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0"),
    ("v_mov_dpp_wave_shr", "v_mov_b32_dpp {r}, %8 wave_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_mov_dpp_row_shr", "v_mov_b32_dpp {r}, %8 row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_add_u32_dpp_row_shr", "v_add_u32_dpp {r}, %8, {r} row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_accvgpr_read", "v_accvgpr_read_b32 {r}, a1"),
    ("v_accvgpr_write", "v_accvgpr_write_b32 a2, {r}"),
    ("v_sat_pk_u8_i16", "v_sat_pk_u8_i16 {r}, {r}"),
    ("v_cvt_f32_i32", "v_cvt_f32_i32 {r}, {r}"),
    ("v_dot2_i32_i16", "v_dot2_i32_i16 {r}, {r}, %8, %9"),
    ("s_nop_0", "s_nop 0"),
]
# two instructions interleaved in one stream: (name, [instr, instr, ...]) cycled over the 64 slots
MIXES = [
    ("pk_max+add_u32 1:1", ["v_pk_max_i16 {r}, {r}, %8", "v_add_u32 {r}, {r}, %8"]),
    ("pk_max+add_u32 3:1", ["v_pk_max_i16 {r}, {r}, %8"] * 3 + ["v_add_u32 {r}, {r}, %8"]),
    ("pk_max+pk_add 1:1", ["v_pk_max_i16 {r}, {r}, %8", "v_pk_add_i16 {r}, {r}, %8"]),
    ("max_i32+add_u32 1:1", ["v_max_i32 {r}, {r}, %8", "v_add_u32 {r}, {r}, %8"]),
    ("pk_max+mov 1:1", ["v_pk_max_i16 {r}, {r}, %8", "v_mov_b32 {r}, %8"]),
    ("pk_max+s_nop 1:1", ["v_pk_max_i16 {r}, {r}, %8", "s_nop 0"]),
    ("pk_max+fma_f32 1:1", ["v_pk_max_i16 {r}, {r}, %8", "v_fma_f32 {r}, {r}, %8, %9"]),
    ("add_u32+and 1:1", ["v_add_u32 {r}, {r}, %8", "v_and_b32 {r}, {r}, %8"]),
    ("pk_max+ds_read 7:1", ["v_pk_max_i16 {r}, {r}, %8"] * 7 + ["ds_read_b32 v60, %9"]),
]
HETERO = [("v_pk_max_i16", "v_add_u32"), ("v_pk_max_i16", "v_pk_max_i16"), ("v_add_u32", "v_add_u32"),
          ("v_pk_max_i16", "v_fma_f32"), ("v_max_i32", "v_add_u32"), ("v_pk_max_i16", "v_mov_b32"),
          ("v_pk_max_i16", "s_nop_0"), ("v_pk_max_i16", "v_max_f32")]


def block(instrs):
    lines = [".p2align 3"]
    for k in range(64):
        lines.append(instrs[k % len(instrs)].replace("{r}", "%%%d" % (k % 8)))
    return "\\n\"\n            \"".join(lines) + "\\n"


def main():
    all_ops = [(n, [t]) for n, t in OPS] + MIXES
    out = []
    out.append("// GENERATED by tools/gen_micro_valu3.py -- see its docstring.  hipcc --offload-arch=gfx950 -O2 micro_valu3.hip -o micro_valu3")
    out.append("""#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
#define OPERANDS : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c0), "v"(c1) \\
    : "vcc", "a0", "a1", "a2", "v60"
""")
    out.append("static const char* op_names[] = {" + ", ".join('"%s"' % n for n, _ in all_ops) + "};")
    out.append("enum { N_OPS = %d, N_SINGLE = %d };" % (len(all_ops), len(OPS)))
    out.append("""
// role 0 = the first wavefront(s) of each SIMD (threads < role_split), role 1 = the rest
__global__ void __launch_bounds__(1024) k(unsigned* out, int iters, int op_a, int op_b, int role_split, long long* ticks, unsigned* hwid) {
    extern __shared__ int lds[];
    const int lane = threadIdx.x;
    unsigned r0 = lane, r1 = lane * 3, r2 = lane ^ 5, r3 = 7 - lane, r4 = lane + 11, r5 = lane - 3, r6 = 2 * lane, r7 = 9;
    unsigned c0 = 0x00010001u * (lane & 3), c1 = (lane & 63) * 4;
    if (lane == 12345) lds[lane] = 1;
    const int op = __builtin_amdgcn_readfirstlane(lane < role_split ? op_a : op_b);
    long long t0 = __builtin_amdgcn_s_memrealtime();
    switch (op) {""")
    for idx, (name, instrs) in enumerate(all_ops):
        out.append("    case %d:   // %s" % (idx, name))
        out.append("        for (int it = 0; it < iters; it++) {")
        out.append("            asm volatile(\"%s\" OPERANDS);" % block(instrs))
        out.append("        }")
        out.append("        break;")
    out.append("""    default: break;
    }
    long long t1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + lane] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    if ((lane & 63) == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        hwid[blockIdx.x * 16 + (lane >> 6)] = id;
        ticks[blockIdx.x * 16 + (lane >> 6)] = t1 - t0;      // 100 MHz
    }
}

struct Result { double ms; double role_ns[2]; int simd[4]; };
static int cus;
static unsigned* d_out; static long long* d_ticks; static unsigned* d_hw;
static const size_t LDS = 96 * 1024;

static Result launch(int op_a, int op_b, int waves_per_simd, int role_split_waves, int iters) {
    dim3 grid(cus), block(256 * waves_per_simd);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, grid, block, LDS, 0, d_out, iters, op_a, op_b, role_split_waves * 64, d_ticks, d_hw);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    Result r; float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); r.ms = ms;
    std::vector<long long> t(cus * 16); CHECK(hipMemcpy(t.data(), d_ticks, cus * 16 * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned> h(cus * 16); CHECK(hipMemcpy(h.data(), d_hw, cus * 16 * 4, hipMemcpyDeviceToHost));
    int nw = 4 * waves_per_simd; double s[2] = {0, 0}; int c[2] = {0, 0};
    for (int b = 0; b < cus; b++) for (int w = 0; w < nw; w++) { int role = w < role_split_waves ? 0 : 1; s[role] += t[b * 16 + w] * 10.0; c[role]++; }
    for (int q = 0; q < 2; q++) r.role_ns[q] = c[q] ? s[q] / c[q] : 0;
    for (int q = 0; q < 4; q++) r.simd[q] = 0;
    for (int w = 0; w < nw; w++) r.simd[(h[w] >> 4) & 3]++;
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return r;
}
static int find(const char* n) { for (int i = 0; i < N_OPS; i++) if (!strcmp(op_names[i], n)) return i; printf("no op %s\\n", n); exit(1); }

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "survey";
    int iters = argc > 2 ? atoi(argv[2]) : 400000;          // x64 instructions: >= 50 ms per run
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    cus = prop.multiProcessorCount;
    CHECK(hipMalloc(&d_out, (size_t) cus * 1024 * 4)); CHECK(hipMalloc(&d_ticks, (size_t) cus * 16 * 8)); CHECK(hipMalloc(&d_hw, (size_t) cus * 16 * 4));
    CHECK(hipFuncSetAttribute((const void*) k, hipFuncAttributeMaxDynamicSharedMemorySize, (int) LDS));
    printf("# %s, %d CUs; one workgroup per CU; %d iterations x 64 instructions per wavefront; clock 2.4 GHz assumed for the cycle columns\\n", prop.gcnArchName, cus, iters);
    launch(find("v_pk_max_i16"), find("v_pk_max_i16"), 2, 8, iters);      // warm-up: clocks settle
    const double insts = (double) iters * 64;
    if (!strcmp(mode, "survey") || !strcmp(mode, "mix")) {
        int lo = !strcmp(mode, "survey") ? 0 : N_SINGLE, hi = !strcmp(mode, "survey") ? N_SINGLE : N_OPS;
        printf("%-26s %2s %10s %22s %20s  %s\\n", "instruction", "W", "ms", "cycles/instr (one wave)", "cycles/instr (SIMD)", "waves on simd 0..3 of workgroup 0");
        for (int op = lo; op < hi; op++)
            for (int w = 1; w <= 4; w *= 2) {
                Result r = launch(op, op, w, 16, iters);
                double ns = r.ms * 1e6 / insts;
                printf("%-26s %2d %10.2f %22.2f %20.2f  %d %d %d %d\\n", op_names[op], w, r.ms, ns * 2.4, ns * 2.4 / w, r.simd[0], r.simd[1], r.simd[2], r.simd[3]);
            }
    } else {
        static const char* pairs[][2] = {""")
    out.append("            " + ", ".join('{"%s", "%s"}' % p for p in HETERO) + " };")
    out.append("""        printf("# two wavefronts per SIMD: the older (first) runs A, the younger runs B; ns per instruction seen by each role\\n");
        printf("%-18s %-18s %12s %12s %14s %14s %10s\\n", "A (older)", "B (younger)", "A alone ns", "B alone ns", "A together ns", "B together ns", "total ms");
        for (auto& p : pairs) {
            int a = find(p[0]), b = find(p[1]);
            Result ra = launch(a, a, 1, 16, iters), rb = launch(b, b, 1, 16, iters), rt = launch(a, b, 2, 4, iters);
            printf("%-18s %-18s %12.3f %12.3f %14.3f %14.3f %10.2f\\n", p[0], p[1], ra.role_ns[0] / insts, rb.role_ns[0] / insts,
                   rt.role_ns[0] / insts, rt.role_ns[1] / insts, rt.ms);
        }
    }
    return 0;
}""")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro_valu3.hip")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    main()

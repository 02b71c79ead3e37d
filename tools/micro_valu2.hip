// micro_valu2: VALU issue rate of gfx950 per instruction and per occupancy, with controlled placement.
//
// Question: MI355X_MICROARCH.md lists "v_fma_f32 (wave64) 2 cyc (SIMD-32); one wave alone: 4".
// Does a second wavefront on the SIMD double the issue rate of the instructions the strip kernels are made of
// (v_pk_max_i16, v_pk_add_i16, v_perm_b32, v_max_i32, DPP moves)?
//
// Placement is controlled, not assumed: ONE workgroup per CU (96 KiB of LDS each; 160 KiB per CU), a workgroup is
// 256*W threads = W wavefronts on each of the 4 SIMDs (checked through HW_REG_HW_ID and printed).  Every wavefront
// runs `iters` x 64 copies of one instruction written in inline assembly on 8 independent registers (INDEP) or on
// one register (SERIAL: each instruction needs its predecessor's result).
//
// Build: hipcc --offload-arch=gfx950 -O2 micro_valu2.hip -o micro_valu2 ; run: ./micro_valu2 [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

#define REP8(I)  I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
// 8 independent instructions; %0..%7 read-write, %8/%9 read-only
#define ASM8(fmt_) \
    asm volatile(fmt_(0) fmt_(1) fmt_(2) fmt_(3) fmt_(4) fmt_(5) fmt_(6) fmt_(7) \
        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c0), "v"(c1))
#define ASM8S(fmt_) \
    asm volatile(fmt_(0) fmt_(0) fmt_(0) fmt_(0) fmt_(0) fmt_(0) fmt_(0) fmt_(0) \
        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c0), "v"(c1))

#define F_FMA(i)      "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define F_ADD(i)      "v_add_u32 %" #i ", %" #i ", %8\n"
#define F_MAXI(i)     "v_max_i32 %" #i ", %" #i ", %8\n"
#define F_MAX3(i)     "v_max3_i32 %" #i ", %" #i ", %8, %9\n"
#define F_ADD3(i)     "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define F_PKMAX(i)    "v_pk_max_i16 %" #i ", %" #i ", %8\n"
#define F_PKADD(i)    "v_pk_add_i16 %" #i ", %" #i ", %8\n"
#define F_PKADDC(i)   "v_pk_add_i16 %" #i ", %" #i ", %8 clamp\n"
#define F_PKSUBC(i)   "v_pk_sub_i16 %" #i ", %" #i ", %8 clamp\n"
#define F_PKMINU(i)   "v_pk_min_u16 %" #i ", %" #i ", %8\n"
#define F_PERM(i)     "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define F_ALIGN(i)    "v_alignbit_b32 %" #i ", %" #i ", %8, 16\n"
#define F_AND(i)      "v_and_b32 %" #i ", %" #i ", %8\n"
#define F_ANDOR(i)    "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define F_DPPW(i)     "v_mov_b32_dpp %" #i ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_DPPR(i)     "v_mov_b32_dpp %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_ADDDPP(i)   "v_add_u32_dpp %" #i ", %8, %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_MOV(i)      "v_mov_b32 %" #i ", %8\n"
#define F_PKFMA32(i)  "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define F_PKMADU(i)   "v_pk_mad_u16 %" #i ", %" #i ", %8, %9\n"
#define F_PKMAXE32(i) "v_max_i32_e64 %" #i ", %" #i ", %8\n"
#define F_ACCRD(i)    "v_accvgpr_read_b32 %" #i ", a" #i "\n"
#define F_SNOP(i)     "s_nop 0\n"

enum { OP_FMA, OP_ADD, OP_MAXI, OP_MAX3, OP_ADD3, OP_PKMAX, OP_PKADD, OP_PKADDC, OP_PKSUBC, OP_PKMINU, OP_PERM,
       OP_ALIGN, OP_AND, OP_ANDOR, OP_DPPW, OP_DPPR, OP_ADDDPP, OP_MOV, OP_PKFMA32, OP_PKMADU, OP_MAXI64, OP_ACCRD,
       OP_SNOP, OP_MIX, OP_COUNT };
static const char* op_name[OP_COUNT] = {
    "v_fma_f32", "v_add_u32", "v_max_i32", "v_max3_i32", "v_add3_u32", "v_pk_max_i16", "v_pk_add_i16",
    "v_pk_add_i16 clamp", "v_pk_sub_i16 clamp", "v_pk_min_u16", "v_perm_b32", "v_alignbit_b32", "v_and_b32",
    "v_and_or_b32", "v_mov_dpp wave_shr", "v_mov_dpp row_shr", "v_add_u32_dpp row_shr", "v_mov_b32",
    "v_pk_fma_f32", "v_pk_mad_u16", "v_max_i32_e64 (8B enc)", "v_accvgpr_read", "s_nop 0", "pk16 cell mix" };

template <int OP, bool SERIAL>
__global__ void __launch_bounds__(1024) k(unsigned* out, int iters, long long* cycles, unsigned* hwid) {
    extern __shared__ int lds[];
    int lane = threadIdx.x;
    unsigned r0 = lane, r1 = lane * 3, r2 = lane ^ 5, r3 = 7 - lane, r4 = lane + 11, r5 = lane - 3, r6 = 2 * lane, r7 = 9;
    unsigned c0 = 0x00010001u * (lane & 3), c1 = 0x03020100u;
    double d0 = lane, d1 = 1.0, d2 = 2.0, d3 = 3.0;
    if (lane == 12345) lds[lane] = 1;       // keep the LDS allocation alive
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#define GO(F) do { if (SERIAL) ASM8S(F); else ASM8(F); } while (0)
            if (OP == OP_FMA) GO(F_FMA);
            else if (OP == OP_ADD) GO(F_ADD);
            else if (OP == OP_MAXI) GO(F_MAXI);
            else if (OP == OP_MAX3) GO(F_MAX3);
            else if (OP == OP_ADD3) GO(F_ADD3);
            else if (OP == OP_PKMAX) GO(F_PKMAX);
            else if (OP == OP_PKADD) GO(F_PKADD);
            else if (OP == OP_PKADDC) GO(F_PKADDC);
            else if (OP == OP_PKSUBC) GO(F_PKSUBC);
            else if (OP == OP_PKMINU) GO(F_PKMINU);
            else if (OP == OP_PERM) GO(F_PERM);
            else if (OP == OP_ALIGN) GO(F_ALIGN);
            else if (OP == OP_AND) GO(F_AND);
            else if (OP == OP_ANDOR) GO(F_ANDOR);
            else if (OP == OP_DPPW) GO(F_DPPW);
            else if (OP == OP_DPPR) GO(F_DPPR);
            else if (OP == OP_ADDDPP) GO(F_ADDDPP);
            else if (OP == OP_MOV) GO(F_MOV);
            else if (OP == OP_PKMADU) GO(F_PKMADU);
            else if (OP == OP_MAXI64) GO(F_PKMAXE32);
            else if (OP == OP_ACCRD) GO(F_ACCRD);
            else if (OP == OP_SNOP) GO(F_SNOP);
            else if (OP == OP_PKFMA32) {
                asm volatile("v_pk_fma_f32 %0, %0, %2, %3\nv_pk_fma_f32 %1, %1, %2, %3\n"
                             "v_pk_fma_f32 %0, %0, %2, %3\nv_pk_fma_f32 %1, %1, %2, %3\n"
                             "v_pk_fma_f32 %0, %0, %2, %3\nv_pk_fma_f32 %1, %1, %2, %3\n"
                             "v_pk_fma_f32 %0, %0, %2, %3\nv_pk_fma_f32 %1, %1, %2, %3\n"
                             : "+v"(d0), "+v"(d1) : "v"(d2), "v"(d3));
            } else if (OP == OP_MIX) {
                // the packed cell of sw_kernel_pk16.inc, two rows' worth: perm, add, 3x max, sub-clamp, max, max, sub
                asm volatile(
                    "v_perm_b32 %4, %8, %9, %0\n"
                    "v_pk_max_i16 %5, %1, %5\n"
                    "v_pk_add_i16 %6, %2, %4\n"
                    "v_pk_sub_i16 %5, %5, %8 clamp\n"
                    "v_pk_max_i16 %6, %6, %5\n"
                    "v_pk_max_i16 %7, %3, %7\n"
                    "v_pk_max_i16 %6, %6, %7\n"
                    "v_pk_sub_i16 %3, %6, %8\n"
                    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c0), "v"(c1));
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + lane] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (unsigned) (d0 + d1);
    if ((lane & 63) == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        hwid[blockIdx.x * 16 + (lane >> 6)] = id;
        if (lane == 0) cycles[blockIdx.x] = t1 - t0;
    }
}

typedef void (*kern_t)(unsigned*, int, long long*, unsigned*);
struct Entry { int op; bool serial; kern_t fn; };
#define E(op) { op, false, k<op, false> }, { op, true, k<op, true> }
static Entry entries[] = {
    E(OP_FMA), E(OP_ADD), E(OP_MAXI), E(OP_MAX3), E(OP_ADD3), E(OP_PKMAX), E(OP_PKADD), E(OP_PKADDC), E(OP_PKSUBC),
    E(OP_PKMINU), E(OP_PERM), E(OP_ALIGN), E(OP_AND), E(OP_ANDOR), E(OP_DPPW), E(OP_DPPR), E(OP_ADDDPP), E(OP_MOV),
    { OP_PKFMA32, false, k<OP_PKFMA32, false> }, E(OP_PKMADU), E(OP_MAXI64), { OP_ACCRD, false, k<OP_ACCRD, false> },
    { OP_SNOP, false, k<OP_SNOP, false> }, { OP_MIX, false, k<OP_MIX, false> } };

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("# device %s  CUs %d  clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    unsigned* d; long long* cyc; unsigned* hw;
    CHECK(hipMalloc(&d, (size_t) cus * 1024 * 4));
    CHECK(hipMalloc(&cyc, (size_t) cus * 8));
    CHECK(hipMalloc(&hw, (size_t) cus * 16 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t lds_bytes = 96 * 1024;
    printf("# one workgroup per CU (96 KiB LDS), W wavefronts per SIMD = blockDim/256; %d iterations x 64 instructions\n", iters);
    printf("%-24s %-6s %2s %10s %14s %18s %16s %s\n", "instruction", "deps", "W", "ms", "cyc/inst/wave", "wave-inst/ns/SIMD",
           "cyc/inst/SIMD", "placement(WG0 simd ids)");
    for (auto& en : entries) {
        CHECK(hipFuncSetAttribute((const void*) en.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes));
        for (int w = 1; w <= 4; w *= 2) {
            dim3 grid(cus), block(256 * w);
            hipLaunchKernelGGL(en.fn, grid, block, lds_bytes, 0, d, 200, cyc, hw);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(en.fn, grid, block, lds_bytes, 0, d, iters, cyc, hw);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<long long> c(cus); CHECK(hipMemcpy(c.data(), cyc, cus * 8, hipMemcpyDeviceToHost));
            std::vector<unsigned> h(cus * 16); CHECK(hipMemcpy(h.data(), hw, cus * 16 * 4, hipMemcpyDeviceToHost));
            double csum = 0; for (auto v : c) csum += (double) v;
            double insts = (double) iters * 64;
            // s_memtime ticks at a constant 100 MHz on this part: convert with the event time instead
            double ns_per_inst_wave = ms * 1e6 / insts;
            double rate = insts * w / (ms * 1e6);
            int simd_count[4] = {0, 0, 0, 0};
            for (int i = 0; i < 4 * w; i++) simd_count[(h[i] >> 4) & 3]++;
            // distinct CUs used (se, sh, cu bits) as a sanity check that every workgroup had its own CU
            printf("%-24s %-6s %2d %10.3f %14.2f %18.3f %16.2f  s0:%d s1:%d s2:%d s3:%d  memtime/inst %.3f\n", op_name[en.op],
                   en.serial ? "serial" : "indep", w, ms, ns_per_inst_wave * 2.4, rate, 2.4 / rate,
                   simd_count[0], simd_count[1], simd_count[2], simd_count[3], csum / cus / insts);
        }
    }
    printf("# cyc columns assume 2.4 GHz; compare rows, and see micro_clock for the clock actually held\n");
    return 0;
}

// Does gfx950 need the `s_nop 0` hipcc puts between a packed (VOP3P) instruction and a VALU instruction that reads its
// result?  Dependent chains of the packed kernel's instructions WITHOUT any wait state, results checked against the
// arithmetic: hipcc --offload-arch=gfx950 -O3 tools/micro_hazard.hip -o tools/_micro_hazard && tools/_micro_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define R4(X) X X X X
#define R16(X) R4(X) R4(X) R4(X) R4(X)

// per lane: x = seed; 16 x { x = pk_add_u16(x, d); x = pk_max_i16(x, f); y = pk_add_i16(x, -3) sat; z = pk_max_i16(y, x); x = z }
__global__ void __launch_bounds__(64) chain(const unsigned* in, unsigned* out, int iters) {
    const int gid = blockIdx.x * 64 + threadIdx.x;
    unsigned x = in[gid], d = 0x00030005u, f = 0x00100020u;
    for (int it = 0; it < iters; it++) {
        asm volatile(R16("v_pk_add_u16 %0, %0, %1\n"
                         "v_pk_max_i16 %0, %0, %2\n"
                         "v_pk_add_i16 v20, %0, -3 op_sel_hi:[1,0] clamp\n"
                         "v_pk_max_i16 %0, v20, %0\n"
                         "v_pk_sub_i16 %0, %0, %1 clamp\n")
                     : "+v"(x) : "v"(d), "v"(f) : "v20");
    }
    out[gid] = x;
}

// the packed instructions of the exact-tracking path: min_u16, mul_lo_u16, mad_u16, sub_u16 -- same question
__global__ void __launch_bounds__(64) chain2(const unsigned* in, unsigned* out, int iters) {
    const int gid = blockIdx.x * 64 + threadIdx.x;
    unsigned x = in[gid], three = 0x00030003u, seven = 0x00070007u, cap = 0x0fff0fffu;
    for (int it = 0; it < iters; it++) {
        asm volatile(R16("v_pk_mul_lo_u16 %0, %0, %1\n"
                         "v_pk_mad_u16 %0, %0, %1, %2\n"
                         "v_pk_min_u16 %0, %0, %3\n"
                         "v_pk_sub_u16 %0, %0, %2\n"
                         "v_pk_add_u16 %0, %0, %1\n")
                     : "+v"(x) : "v"(three), "v"(seven), "v"(cap));
    }
    out[gid] = x;
}

static unsigned ref(unsigned x, int iters) {
    auto lo = [](unsigned v) { return (int) (short) (v & 0xffff); };
    auto hi = [](unsigned v) { return (int) (short) (v >> 16); };
    auto pack = [](int l, int h) { return ((unsigned) (unsigned short) l) | ((unsigned) (unsigned short) h << 16); };
    auto sat = [](int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); };
    for (int it = 0; it < iters * 16; it++) {
        unsigned a = pack((lo(x) + 5) & 0xffff, (hi(x) + 3) & 0xffff);          // u16 add wraps
        a = pack(std::max(lo(a), 0x20), std::max(hi(a), 0x10));
        unsigned y = pack(sat(lo(a) - 3), sat(hi(a) - 3));
        unsigned z = pack(std::max(lo(y), lo(a)), std::max(hi(y), hi(a)));
        x = pack(sat(lo(z) - 5), sat(hi(z) - 3));
    }
    return x;
}

static unsigned ref2(unsigned x, int iters) {
    auto f = [](unsigned v) {
        v = (v * 3) & 0xffff;
        v = (v * 3 + 7) & 0xffff;
        v = v < 0x0fff ? v : 0x0fff;
        v = (v - 7) & 0xffff;
        return (v + 3) & 0xffff;
    };
    unsigned lo = x & 0xffff, hi = x >> 16;
    for (int it = 0; it < iters * 16; it++) { lo = f(lo); hi = f(hi); }
    return lo | (hi << 16);
}

int main() {
    const int blocks = 1024, n = blocks * 64, iters = 2000;
    std::vector<unsigned> h(n), o(n);
    for (int i = 0; i < n; i++) h[i] = (unsigned) (i * 2654435761u) & 0x3fff3fffu;
    unsigned *din, *dout;
    CHECK(hipMalloc(&din, n * 4)); CHECK(hipMalloc(&dout, n * 4));
    CHECK(hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice));
    long bad = 0;
    for (int rep = 0; rep < 20; rep++) {
        hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, din, dout, iters);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i += 97) if (o[i] != ref(h[i], iters)) bad++;
    }
    printf("dependent packed chains without wait states: %ld mismatches in %d checked results x 20 launches\n", bad, n / 97 + 1);
    long bad2 = 0;
    for (int rep = 0; rep < 20; rep++) {
        hipLaunchKernelGGL(chain2, dim3(blocks), dim3(64), 0, 0, din, dout, iters);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i += 97) if (o[i] != ref2(h[i], iters)) bad2++;
    }
    printf("same with mul_lo / mad / min / sub (exact-tracking path): %ld mismatches\n", bad2);
    return bad != 0 || bad2 != 0;
}

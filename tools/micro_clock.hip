// long-running VALU load: what clock does the chip hold? (s_memtime vs s_memrealtime @100MHz)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) k(int* out, int iters, long long* res) {
    int lane = threadIdx.x;
    int a = lane, b = lane * 3, c = lane ^ 5, d = 7 - lane, e = lane + 11, f = lane - 3, g = 2 * lane, h = 9;
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            a = max(a + -2, b); b = max(b + -2, c); c = max(c + -2, d); d = max(d + -2, e);
            e = max(e + -2, f); f = max(f + -2, g); g = max(g + -2, h); h = max(h + -2, a);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + lane] = a + b + c + d + e + f + g + h;
    if (lane == 0) { res[blockIdx.x * 2] = t1 - t0; res[blockIdx.x * 2 + 1] = r1 - r0; }
}
int main() {
    int* d; long long* res; int grid = 256 * 4 * 2;
    hipMalloc(&d, grid * 64 * 4); hipMalloc(&res, grid * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int iters : {200000, 2000000}) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, d, iters, res);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        static long long h[4096 * 2]; hipMemcpy(h, res, grid * 16, hipMemcpyDeviceToHost);
        double fmin = 1e9, fmax = 0; 
        for (int b = 0; b < grid; b++) { double f = (double) h[2*b] / h[2*b+1] * 100.0; if (f < fmin) fmin = f; if (f > fmax) fmax = f; }
        printf("iters=%d  %.1f ms  clock MHz min %.0f max %.0f ; inst/ns/SIMD=%.3f\n", iters, ms, fmin, fmax, (double) iters * 256 * 2 / (ms * 1e6));
    }
    return 0;
}

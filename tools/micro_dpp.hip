// micro-probe: DPP wave_shr:1 semantics and pinned-host system-scope stores on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out, int* host_flag) {
    int lane = threadIdx.x;
    int src = 100 + lane;
    int old = -7;
    int y = __builtin_amdgcn_update_dpp(old, src, 0x138, 0xf, 0xf, false);
    out[lane] = y;
    if (lane == 0) __hip_atomic_store(host_flag, 42, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int main() {
    int* d; int* hf;
    hipMalloc(&d, 64 * 4);
    hipHostMalloc((void**)&hf, 64, hipHostMallocMapped);
    *hf = 0;
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e; hipEventCreate(&e);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d, hf);
    hipEventRecord(e, s);
    int spins = 0;
    while (hipEventQuery(e) == hipErrorNotReady && spins < 100000000) spins++;
    printf("event ready after %d spins, host flag %d\n", spins, *hf);
    int h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("lane0=%d lane1=%d lane16=%d lane32=%d lane63=%d\n", h[0], h[1], h[16], h[32], h[63]);
    return 0;
}

// does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) let two independent kernels of ONE stream overlap on this stack (gfx950, ROCm 7.2)?
// two single-block spin kernels of 1 ms each on one stream: ~1 ms = overlapped, ~2 ms = the flag is ignored.   hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long ticks, int* out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (out) *out = 1;
}
int main() {
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    long long ticks = 100000;   // 1 ms at 100 MHz
    int* null_out = nullptr;
    void* args[] = {&ticks, &null_out};
    for (int flags = 0; flags < 2; ++flags) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a, st);
            hipExtLaunchKernel((const void*)spin, dim3(1), dim3(64), args, 0, st, nullptr, nullptr, 0);
            hipExtLaunchKernel((const void*)spin, dim3(1), dim3(64), args, 0, st, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0);
            hipEventRecord(b, st);
            hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            printf("flags=%d: two 1-ms kernels on one stream took %.3f ms\n", flags, ms);
        }
    }
    return 0;
}

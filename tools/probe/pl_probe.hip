// probe of v_permlane32_swap semantics
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
    unsigned a = threadIdx.x, b = threadIdx.x + 100;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 512); k<<<1, 64>>>(d); unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("r0:"); for (int i = 0; i < 64; i += 8) printf(" %u", h[i]); printf("\nr1:"); for (int i = 0; i < 64; i += 8) printf(" %u", h[64 + i]); printf("\n");
    return 0;
}

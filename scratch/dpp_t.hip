#include <hip/hip_runtime.h>
__device__ __forceinline__ float hsum3_asm(float v) {
    float t, r;
    asm("v_add_f32_dpp %0, %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(t) : "v"(v));
    asm("v_add_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v), "v"(t));
    return r;
}
__global__ void k(const float* a, const float* b, float* o) {
    float x = a[threadIdx.x] * b[threadIdx.x];
    float y = a[threadIdx.x] + b[threadIdx.x];
    o[threadIdx.x] = hsum3_asm(x) * hsum3_asm(y);
}

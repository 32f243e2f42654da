// headact.hip -- the tails of the prediction heads (gfx950): what sits between a 1x1 predictor convolution and its loss.
//
// The predictors have 1, 2, 12 or 20 output channels (mg_net.py:597-605, 676-695, 796-825; layers.py:163-167); the convolution
// kernels work on channels padded to 32, channels-last, 16-bit.  The reference then applies `.float()`, sigmoid (centre heat map,
// :694), sigmoid / 0.5 (inverse depth, :819-823) -- and autograd undoes each of those plus the channel slice with one ATen kernel
// apiece (cast, sigmoid_backward, mul, zero-fill, strided copy): ~8 launches per head and step on maps of a few hundred KB.
//   head_act_fwd  padded 16-bit [B,h,w,P] -> fp32 [B,C,h,w]: y = x | sigmoid(x) | 2 sigmoid(x)
//   head_act_bwd  gradient wrt y (fp32, any strides: the loss kernels' NHWC tables or an NCHW map) -> gradient wrt the PADDED
//                 16-bit predictor output [B,h,w,P], channels >= C zero: d x = g | g y (1 - y) | g y (1 - y / 2)
// One thread per low-resolution pixel; 64-byte rows; trivially memory-bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgnet_hip.h"

namespace {

__device__ __forceinline__ float h2f(uint16_t h, int f16) {
    return f16 ? (float)__builtin_bit_cast(_Float16, h) : __uint_as_float((uint32_t)h << 16);
}
__device__ __forceinline__ uint16_t f2h(float v, int f16) {
    return f16 ? __builtin_bit_cast(uint16_t, (_Float16)v) : __builtin_bit_cast(uint16_t, (__bf16)v);
}

__global__ __launch_bounds__(256) void head_act_fwd(const uint16_t* __restrict__ x, int P, int C, long npix, long hw, int kind, int f16,
                                                    float* __restrict__ y) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const long b = i / hw, r = i % hw;
    for (int c = 0; c < C; ++c) {
        float v = h2f(x[i * P + c], f16);
        if (kind) {
            v = 1.f / (1.f + expf(-v));
            if (kind == 2) v = v / 0.5f;    // mg_net.py:822: sigmoid(x) / 0.5
        }
        y[(b * C + c) * hw + r] = v;
    }
}

// g: element strides (sb, sc, sp) per batch / channel / pixel; y: the forward's output [B,C,h,w] (kind != 0).
// One thread per (pixel, 8-channel group): 16-byte stores, four lanes cover a pixel's 64-byte row (the one-thread-per-pixel form wrote
// 32 scattered 2-byte values per lane: 126 us for the 20-class map of a 1024 x 2048 batch, 0.26 ms per step over the six heads).
__global__ __launch_bounds__(256) void head_act_bwd(const float* __restrict__ g, long sb, long sc, long sp, const float* __restrict__ y, int P,
                                                    int C, long npix, long hw, int kind, int f16, float gscale, uint16_t* __restrict__ dx) {
    const int gpp = P / 8;                                   // 8-channel groups per pixel (P is a multiple of 8)
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long i = t / gpp;
    const int c0 = (int)(t - i * gpp) * 8;
    if (i >= npix) return;
    const long b = i / hw, r = i % hw;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    if (c0 < C) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + k;
            float v = 0.f;
            if (c < C) {
                v = g[b * sb + c * sc + r * sp] * gscale;
                if (kind) {
                    const float yv = y[(b * C + c) * hw + r];
                    v *= kind == 1 ? yv * (1.f - yv) : yv * (1.f - 0.5f * yv);   // d (2 s) = 2 s (1 - s) = y (1 - y / 2)
                }
            }
            w[k >> 1] |= (uint32_t)f2h(v, f16) << ((k & 1) * 16);
        }
    }
    reinterpret_cast<uint4*>(dx)[t] = make_uint4(w[0], w[1], w[2], w[3]);
}

}  // namespace

extern "C" {

int mgn_head_act_fwd(const void* x_padded, int B, int h, int w, int P, int C, int kind, int is_f16, float* y, void* stream) {
    if (!x_padded || !y || B < 1 || h < 1 || w < 1 || C < 1 || C > P || kind < 0 || kind > 2) return MGN_EINVAL;
    const long hw = (long)h * w, npix = B * hw;
    hipLaunchKernelGGL(head_act_fwd, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x_padded, P, C,
                       npix, hw, kind, is_f16 ? 1 : 0, y);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_head_act_bwd(const float* g, long sb, long sc, long sp, const float* y, int B, int h, int w, int P, int C, int kind, int is_f16,
                     float gscale, void* dx_padded, void* stream) {
    if (!g || !dx_padded || B < 1 || h < 1 || w < 1 || C < 1 || C > P || kind < 0 || kind > 2 || (kind && !y)) return MGN_EINVAL;
    const long hw = (long)h * w, npix = B * hw;
    if (P % 8 || ((uintptr_t)dx_padded & 15)) return MGN_EINVAL;
    const long nthreads = npix * (P / 8);
    hipLaunchKernelGGL(head_act_bwd, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, sb, sc, sp, y, P, C, npix, hw,
                       kind, is_f16 ? 1 : 0, gscale, (uint16_t*)dx_padded);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

}  // extern "C"

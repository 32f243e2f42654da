// pool.hip -- 3x3 stride-2 pad-1 max pooling on channels-last bf16 activations (ResNet stem, res_net.py:109:
// F.max_pool2d(x, kernel_size=3, stride=2, padding=1)), forward with a 1-byte arg-max tap per element and a gather-style
// backward (each input pixel collects from the <= 4 windows that contain it: no atomics, deterministic).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgnet_hip.h"

namespace {

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }

// one thread: 8 channels (16 bytes) of one output pixel
__global__ __launch_bounds__(256) void maxpool_fwd(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, uint8_t* __restrict__ idx,
                                                   int N, int IH, int IW, int C, int OH, int OW) {
    const int cv = C / 8;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * OH * OW * cv) return;
    const int c8 = (int)(i % cv);
    long r = i / cv;
    const int ow = (int)(r % OW); r /= OW;
    const int oh = (int)(r % OH);
    const int n = (int)(r / OH);
    float best[8];
    uint16_t bits[8];
    uint8_t arg[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { best[k] = -3.4e38f; bits[k] = 0xff7f; arg[k] = 0; }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh * 2 - 1 + kh;
        if (ih < 0 || ih >= IH) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int iw = ow * 2 - 1 + kw;
            if (iw < 0 || iw >= IW) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(x + (((long)n * IH + ih) * IW + iw) * C + c8 * 8);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint16_t b = (uint16_t)(w[k >> 1] >> ((k & 1) * 16));
                const float f = bf2f(b);
                if (f > best[k]) { best[k] = f; bits[k] = b; arg[k] = (uint8_t)(kh * 3 + kw); }  // first maximum wins, like ATen
            }
        }
    }
    uint4 o;
    o.x = bits[0] | ((uint32_t)bits[1] << 16); o.y = bits[2] | ((uint32_t)bits[3] << 16);
    o.z = bits[4] | ((uint32_t)bits[5] << 16); o.w = bits[6] | ((uint32_t)bits[7] << 16);
    const long op = (((long)n * OH + oh) * OW + ow) * C + c8 * 8;
    *reinterpret_cast<uint4*>(y + op) = o;
    uint2 a;
    a.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | ((uint32_t)arg[3] << 24);
    a.y = arg[4] | (arg[5] << 8) | (arg[6] << 16) | ((uint32_t)arg[7] << 24);
    *reinterpret_cast<uint2*>(idx + op) = a;
}

// one thread: 8 channels of one INPUT pixel
__global__ __launch_bounds__(256) void maxpool_bwd(const uint16_t* __restrict__ dy, const uint8_t* __restrict__ idx, uint16_t* __restrict__ dx,
                                                   int N, int IH, int IW, int C, int OH, int OW) {
    const int cv = C / 8;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * IH * IW * cv) return;
    const int c8 = (int)(i % cv);
    long r = i / cv;
    const int iw = (int)(r % IW); r /= IW;
    const int ih = (int)(r % IH);
    const int n = (int)(r / IH);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    // windows (oh, ow) with oh*2-1+kh == ih  =>  oh in {(ih+1)/2, (ih+1)/2 - 1 ...}: at most 2 per axis
    for (int oh = (ih + 1 - 2 + 1) / 2 < 0 ? 0 : (ih) / 2; oh <= (ih + 1) / 2 && oh < OH; ++oh) {
        const int kh = ih - (oh * 2 - 1);
        if (kh < 0 || kh > 2) continue;
        for (int ow = iw / 2; ow <= (iw + 1) / 2 && ow < OW; ++ow) {
            const int kw = iw - (ow * 2 - 1);
            if (kw < 0 || kw > 2) continue;
            const long op = (((long)n * OH + oh) * OW + ow) * C + c8 * 8;
            const uint4 g = *reinterpret_cast<const uint4*>(dy + op);
            const uint2 a = *reinterpret_cast<const uint2*>(idx + op);
            const uint32_t gw[4] = {g.x, g.y, g.z, g.w};
            const uint32_t aw[2] = {a.x, a.y};
            const int tap = kh * 3 + kw;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if ((int)((aw[k >> 2] >> ((k & 3) * 8)) & 0xff) == tap) acc[k] += bf2f((uint16_t)(gw[k >> 1] >> ((k & 1) * 16)));
        }
    }
    auto f2bf = [](float f) -> uint32_t { uint32_t u = __float_as_uint(f); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
    uint4 o;
    o.x = f2bf(acc[0]) | (f2bf(acc[1]) << 16); o.y = f2bf(acc[2]) | (f2bf(acc[3]) << 16);
    o.z = f2bf(acc[4]) | (f2bf(acc[5]) << 16); o.w = f2bf(acc[6]) | (f2bf(acc[7]) << 16);
    *reinterpret_cast<uint4*>(dx + (((long)n * IH + ih) * IW + iw) * C + c8 * 8) = o;
}

}  // namespace

extern "C" {

int mgn_maxpool3x3s2_fwd(const void* x_bf16, void* y_bf16, uint8_t* argmax, int N, int IH, int IW, int C, void* stream) {
    if (!x_bf16 || !y_bf16 || !argmax || N < 1 || IH < 1 || IW < 1 || C < 8 || C % 8) return MGN_EINVAL;
    const int OH = (IH + 2 - 3) / 2 + 1, OW = (IW + 2 - 3) / 2 + 1;
    const long n = (long)N * OH * OW * (C / 8);
    hipLaunchKernelGGL(maxpool_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x_bf16,
                       (uint16_t*)y_bf16, argmax, N, IH, IW, C, OH, OW);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_maxpool3x3s2_bwd(const void* dy_bf16, const uint8_t* argmax, void* dx_bf16, int N, int IH, int IW, int C, void* stream) {
    if (!dy_bf16 || !dx_bf16 || !argmax || N < 1 || IH < 1 || IW < 1 || C < 8 || C % 8) return MGN_EINVAL;
    const int OH = (IH + 2 - 3) / 2 + 1, OW = (IW + 2 - 3) / 2 + 1;
    const long n = (long)N * IH * IW * (C / 8);
    hipLaunchKernelGGL(maxpool_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dy_bf16, argmax,
                       (uint16_t*)dx_bf16, N, IH, IW, C, OH, OW);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

}  // extern "C"

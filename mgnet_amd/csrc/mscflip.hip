// mscflip.hip -- multi-scale + flip inference for MI355X (gfx950).
//
// Replaces the per-pass tensor algebra of mgnet/modeling/mg_net.py:427-520 `forward_multi_scale_flip`: 7 scales x 2 flips = 14
// passes, each of which rescales the normalised frame (bilinear, align_corners=True), flips it, runs the network, and then -- at FULL
// output resolution, for 20 + 1 + 2 + 1 channels -- upsamples the stride-8 head outputs, soft-maxes / rescales / inverts them, flips
// them back and adds them to running averages.  In the reference that is ~12 ATen passes over [N,20,H,W] fp32 per network pass.
// Here: ONE kernel per head output and pass that reads the low-resolution map (L2-resident), evaluates upsample -> soft-max |
// offset scaling | 1 / depth -> un-flip in registers and does the read-modify-write of the accumulator: 8 B per output element and
// pass, the HBM floor of a running sum held in fp32.  HBM-bound streaming work, no LDS, no MFMA.
//
//   msc_input       normalised fp32 frame [N,3,H,W] -> rescaled (+ flipped) network input [N,h,w,8] 16-bit channels-last, channels
//                   3..7 zero (the stem kernels' packed layout, like csrc/prep.hip)
//   msc_accumulate  acc[N,C,H,W] (+)= f(upsample(lr[N,C,h,w]))   f = soft-max over C | identity | (v * stride) / scale with the x
//                   offset negated on flipped passes | 1 / max(v, 1e-6);   the last pass also divides by the number of passes
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgnet_hip.h"

namespace {

constexpr int MSC_MAX_C = 32;

__device__ __forceinline__ float ld_any(const void* p, long i, int dt) {
    if (dt == 0) return ((const float*)p)[i];
    const uint16_t h = ((const uint16_t*)p)[i];
    if (dt == 1) return __uint_as_float((uint32_t)h << 16);            // bf16
    return (float)__builtin_bit_cast(_Float16, h);                      // fp16
}

// torch's upsample_bilinear2d, align_corners=True: source index = dst * (in - 1) / (out - 1) in fp32, floor, lambda
struct Axis { int i0, i1; float l0, l1; };
__device__ __forceinline__ Axis axis(int dst, float scale, int in) {
#pragma clang fp contract(off)   // (torch rounds the source index before taking its fraction: a fused scale * dst - floor moves lambda by 1 ulp of s)
    const float s = scale * (float)dst;
    Axis a;
    a.i0 = min((int)s, in - 1);
    a.i1 = a.i0 + (a.i0 < in - 1 ? 1 : 0);
    a.l1 = s - (float)a.i0;
    a.l0 = 1.f - a.l1;
    return a;
}

struct AccParams {
    const void* lr;
    float* acc;
    long sn, sc, sh, sw;   // element strides of lr[N,C,h,w]
    int N, C, h, w, H, W;
    int dtype, mode, flip, first;
    float stride, scale, divide;   // mode 2: (v * stride) / scale; divide > 0: acc = (acc + v) / divide (the last pass)
};
MGN_PLAN_RO(AccParams, MGN_RO(lr))   // pointers the kernels only read through (launch-plan dependency analysis, csrc/mgn_launch.h)

__device__ __forceinline__ float msc_post(float v, int mode, int c, const AccParams& p) {
    if (mode == 2) {           // offsets: * stride / scale; x component (channel 1) mirrored on flipped passes
        v = (v * p.stride) / p.scale;
        if (p.flip && c == 1) v = -v;
    } else if (mode == 3) {    // inv2depth (depth.py:15)
        v = 1.f / fmaxf(v, 1e-6f);
    }
    return v;
}
__device__ __forceinline__ void msc_store(float* a, float v, const AccParams& p) {
    float o = p.first ? v : *a + v;
    if (p.divide > 0.f) o = o / p.divide;
    *a = o;
}
__device__ __forceinline__ void unpack8h(const uint4& q, int f16, float (&v)[8]) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (f16) {
            v[2 * k] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[k] & 0xffffu));
            v[2 * k + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[k] >> 16));
        } else {
            v[2 * k] = __uint_as_float(w[k] << 16);
            v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
        }
    }
}

// VEC: 16-bit map whose channels are contiguous and padded to a multiple of 8 per pixel (the predictors' channel-padded NHWC outputs):
// 16-byte loads of 8 channels per corner, every value in a register with a static index (the first version indexed a 32-entry array
// under a run-time channel count with a run-time format switch per load: 512 registers, 6 016 spilled -- 6.8 ms per pass).
// !VEC: any strides / fp32: channel by channel; the soft-max then takes two passes over the (L2-resident) low-resolution corners.
template <int MODE, bool VEC>
__global__ __launch_bounds__(256) void msc_accumulate(AccParams p) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), n = blockIdx.z;
    if (x >= p.W || y >= p.H) return;
    // r = flip(upsample(net(flip(frame)))): the value at output column x is the upsampled map at column W-1-x
    const int xs = p.flip ? p.W - 1 - x : x;
    const float sy = p.H > 1 ? (float)(p.h - 1) / (float)(p.H - 1) : 0.f, sx = p.W > 1 ? (float)(p.w - 1) / (float)(p.W - 1) : 0.f;
    const Axis ay = axis(y, sy, p.h), ax = axis(xs, sx, p.w);
    const long b00 = n * p.sn + ay.i0 * p.sh + ax.i0 * p.sw, b01 = n * p.sn + ay.i0 * p.sh + ax.i1 * p.sw;
    const long b10 = n * p.sn + ay.i1 * p.sh + ax.i0 * p.sw, b11 = n * p.sn + ay.i1 * p.sh + ax.i1 * p.sw;
    const long plane = (long)p.H * p.W;
    float* a = p.acc + ((long)n * p.C) * plane + (long)y * p.W + x;
    if constexpr (VEC) {
        const uint16_t* lr = (const uint16_t*)p.lr;
        const int f16 = p.dtype == 2;
        float v[MSC_MAX_C];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < MSC_MAX_C / 8; ++k) {
            if (k * 8 < p.C) {
                float c00[8], c01[8], c10[8], c11[8];
                unpack8h(*reinterpret_cast<const uint4*>(lr + b00 + k * 8), f16, c00);
                unpack8h(*reinterpret_cast<const uint4*>(lr + b01 + k * 8), f16, c01);
                unpack8h(*reinterpret_cast<const uint4*>(lr + b10 + k * 8), f16, c10);
                unpack8h(*reinterpret_cast<const uint4*>(lr + b11 + k * 8), f16, c11);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[k * 8 + j] = ay.l0 * (ax.l0 * c00[j] + ax.l1 * c01[j]) + ay.l1 * (ax.l0 * c10[j] + ax.l1 * c11[j]);
                    if (k * 8 + j < p.C) mx = fmaxf(mx, v[k * 8 + j]);
                }
            }
        }
        float inv = 1.f;
        if (MODE == 0) {   // F.softmax(r, 1)
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < MSC_MAX_C; ++c)
                if (c < p.C) { v[c] = expf(v[c] - mx); sum += v[c]; }
            inv = 1.f / sum;
        }
#pragma unroll
        for (int c = 0; c < MSC_MAX_C; ++c)
            if (c < p.C) msc_store(a + c * plane, MODE == 0 ? v[c] * inv : msc_post(v[c], MODE, c, p), p);
    } else {
        auto blend = [&](int c) {
            const float v00 = ld_any(p.lr, b00 + c * p.sc, p.dtype), v01 = ld_any(p.lr, b01 + c * p.sc, p.dtype);
            const float v10 = ld_any(p.lr, b10 + c * p.sc, p.dtype), v11 = ld_any(p.lr, b11 + c * p.sc, p.dtype);
            return ay.l0 * (ax.l0 * v00 + ax.l1 * v01) + ay.l1 * (ax.l0 * v10 + ax.l1 * v11);
        };
        if (MODE == 0) {
            float mx = -3.0e38f;
            for (int c = 0; c < p.C; ++c) mx = fmaxf(mx, blend(c));
            float sum = 0.f;
            for (int c = 0; c < p.C; ++c) sum += expf(blend(c) - mx);
            const float inv = 1.f / sum;
            for (int c = 0; c < p.C; ++c) msc_store(a + c * plane, expf(blend(c) - mx) * inv, p);
        } else {
            for (int c = 0; c < p.C; ++c) msc_store(a + c * plane, msc_post(blend(c), MODE, c, p), p);
        }
    }
}

struct InParams {
    const float* src;   // [N,3,H,W] fp32
    uint16_t* dst;      // [N,h,w,8] 16-bit
    int N, H, W, h, w, flip, f16;
};
MGN_PLAN_RO(InParams, MGN_RO(src))   // pointers the kernels only read through (launch-plan dependency analysis, csrc/mgn_launch.h)

__global__ __launch_bounds__(256) void msc_input(InParams p) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), n = blockIdx.z;
    if (x >= p.w || y >= p.h) return;
    const int xs = p.flip ? p.w - 1 - x : x;   // torch.flip of the rescaled frame
    const float sy = p.h > 1 ? (float)(p.H - 1) / (float)(p.h - 1) : 0.f, sx = p.w > 1 ? (float)(p.W - 1) / (float)(p.w - 1) : 0.f;
    const Axis ay = axis(y, sy, p.H), ax = axis(xs, sx, p.W);
    uint32_t o[4] = {0u, 0u, 0u, 0u};
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* s = p.src + ((long)n * 3 + c) * p.H * p.W;
        const float v00 = s[(long)ay.i0 * p.W + ax.i0], v01 = s[(long)ay.i0 * p.W + ax.i1];
        const float v10 = s[(long)ay.i1 * p.W + ax.i0], v11 = s[(long)ay.i1 * p.W + ax.i1];
        v[c] = ay.l0 * (ax.l0 * v00 + ax.l1 * v01) + ay.l1 * (ax.l0 * v10 + ax.l1 * v11);
    }
    if (p.f16 == 2) {   // fp32 trunk (SOLVER.AMP.ENABLED False): [N,h,w,3] fp32 = a channels_last [N,3,h,w] tensor
        float* d = reinterpret_cast<float*>(p.dst) + (((long)n * p.h + y) * p.w + x) * 3;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2];
        return;
    }
    uint16_t hv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        hv[c] = p.f16 ? __builtin_bit_cast(uint16_t, (_Float16)v[c]) : __builtin_bit_cast(uint16_t, (__bf16)v[c]);
    o[0] = (uint32_t)hv[0] | ((uint32_t)hv[1] << 16);
    o[1] = (uint32_t)hv[2];
    reinterpret_cast<uint4*>(p.dst)[((long)n * p.h + y) * p.w + x] = make_uint4(o[0], o[1], o[2], o[3]);
}

}  // namespace

extern "C" {

int mgn_msc_input(const float* norm_nchw, int N, int H, int W, int h, int w, int flip, int out_f16, void* out_nhwc8, void* stream) {
    if (!norm_nchw || !out_nhwc8 || N < 1 || H < 1 || W < 1 || h < 1 || w < 1 || ((uintptr_t)out_nhwc8 & 15)) return MGN_EINVAL;
    if (out_f16 < 0 || out_f16 > 2) return MGN_EINVAL;
    InParams p{norm_nchw, (uint16_t*)out_nhwc8, N, H, W, h, w, flip ? 1 : 0, out_f16};   // out_f16: 0 bf16, 1 fp16 ([N,h,w,8]); 2 fp32 ([N,h,w,3])
    hipLaunchKernelGGL(msc_input, dim3((w + 63) / 64, (h + 3) / 4, N), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_msc_accumulate(const void* lr, int dtype, long sn, long sc, long sh, long sw, int N, int C, int h, int w, int H, int W, int mode,
                       int flip, int first, float stride, float scale, float divide, float* acc, void* stream) {
    if (!lr || !acc || N < 1 || C < 1 || C > MSC_MAX_C || h < 1 || w < 1 || H < 1 || W < 1) return MGN_EINVAL;
    if (dtype < 0 || dtype > 2 || mode < 0 || mode > 3 || (mode == 2 && !(scale > 0.f))) return MGN_EINVAL;
    AccParams p{lr, acc, sn, sc, sh, sw, N, C, h, w, H, W, dtype, mode, flip ? 1 : 0, first ? 1 : 0, stride, scale, divide};
    const dim3 grid((W + 63) / 64, (H + 3) / 4, N), block(256);
    hipStream_t s = (hipStream_t)stream;
    // vector path: 16-bit, channels contiguous, every pixel row 16-byte aligned and padded to a multiple of 8 channels
    const bool vec = dtype != 0 && sc == 1 && sn % 8 == 0 && sh % 8 == 0 && sw % 8 == 0 && sw >= (C + 7) / 8 * 8 && ((uintptr_t)lr & 15) == 0;
#define MGN_MSC(M) do { if (vec) hipLaunchKernelGGL((msc_accumulate<M, true>), grid, block, 0, s, p); \
                        else hipLaunchKernelGGL((msc_accumulate<M, false>), grid, block, 0, s, p); } while (0)
    switch (mode) {
        case 0: MGN_MSC(0); break;
        case 1: MGN_MSC(1); break;
        case 2: MGN_MSC(2); break;
        default: MGN_MSC(3); break;
    }
#undef MGN_MSC
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

}  // extern "C"

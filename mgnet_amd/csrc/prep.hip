// prep.hip -- network input assembly: uint8 frames -> normalised, channel-padded NHWC bf16.
//
// Replaces mgnet/modeling/mg_net.py:250-264: `x.float()/255`, `(x - pixel_mean)/pixel_std` for image / image_prev /
// image_next and the channel concatenation fed to PoseCNN -- ~12 elementwise torch kernels and three fp32 copies of every
// frame in the reference.  One pass: reads 3 bytes per frame and pixel, writes one 8-, 16- or 32-byte NHWC pixel whose padding
// channels are zero (the layout the packed-tap stem convolution consumes).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgnet_hip.h"
#include "h16.h"

namespace {

struct PrepParams {
    const uint8_t* frames[3];  // each [B, 3, H, W] uint8
    float scale[3], shift[3];  // y = u8 * scale[c] + shift[c]  ( = (u8/255 - mean_c) / std_c )
    int nf, B, H, W, Cp;
    uint16_t* out;             // [B, H, W, Cp] bf16
};
MGN_PLAN_RO(PrepParams, MGN_RO(frames))   // pointers the kernels only read through (launch-plan dependency analysis, csrc/mgn_launch.h)

__device__ __forceinline__ uint32_t f2bf(float f) { return mgn_f2h(f); }   // this TU's 16-bit format (h16.h)

__global__ __launch_bounds__(256) void prep_kernel(PrepParams p) {
    const long hw = (long)p.H * p.W;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)p.B * hw) return;
    const long b = i / hw, px = i - b * hw;
    uint32_t v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = 0;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        if (f >= p.nf) break;
        const uint8_t* src = p.frames[f] + (b * 3) * hw + px;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[f * 3 + c] = f2bf((float)src[c * hw] * p.scale[c] + p.shift[c]);
    }
    if (p.Cp == 4) {   // the dense-row backbone stem (csrc/conv_stem.hip CP = 4): 8 bytes per pixel
        *reinterpret_cast<uint2*>(p.out + i * 4) = make_uint2(v[0] | (v[1] << 16), v[2]);
        return;
    }
    uint4* dst = reinterpret_cast<uint4*>(p.out + i * p.Cp);
    dst[0] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
    if (p.Cp == 16) dst[1] = make_uint4(v[8] | (v[9] << 16), v[10] | (v[11] << 16), v[12] | (v[13] << 16), v[14] | (v[15] << 16));
}

// uint8 frames -> fp32 in [0,1] (x.float() / 255, mg_net.py:320-335: the un-jittered frames of the photometric loss), stacked
// into one batch tensor: replaces torch.stack + a type-promoting division (two passes, the second at 2.4 TB/s)
struct U8Frames { const uint8_t* f[16]; };
MGN_PLAN_RO(U8Frames, MGN_RO(f))
__global__ __launch_bounds__(256) void u8_frames_to_f32(U8Frames fr, long n16, float divisor, float* __restrict__ out) {
    const uint8_t* src = fr.f[blockIdx.y];
    float4* dst = reinterpret_cast<float4*>(out) + (long)blockIdx.y * n16 * 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        const uint4 v = reinterpret_cast<const uint4*>(src)[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)   // IEEE division: bit-identical to torch's true division
            dst[i * 4 + k] = make_float4(__fdiv_rn((float)(w[k] & 255u), divisor), __fdiv_rn((float)((w[k] >> 8) & 255u), divisor),
                                         __fdiv_rn((float)((w[k] >> 16) & 255u), divisor), __fdiv_rn((float)(w[k] >> 24), divisor));
    }
}

// the same conversion into a pixel-interleaved RGBx batch [B][H][W][4] (4th channel 0): the layout the reprojection kernel gathers
// with one 16-byte load per bilinear corner (mgn_reproj_cfg.frame_layout = MGN_FRAMES_CTX_RGBX_F32); each thread converts 4 consecutive pixels
__global__ __launch_bounds__(256) void u8_frames_to_f32_nhwc4(U8Frames fr, long hw4, float divisor, float* __restrict__ out) {
    const uint8_t* src = fr.f[blockIdx.y];
    float4* dst = reinterpret_cast<float4*>(out) + (long)blockIdx.y * hw4 * 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw4; i += (long)gridDim.x * 256) {
        const uint32_t r = reinterpret_cast<const uint32_t*>(src)[i], g = reinterpret_cast<const uint32_t*>(src + hw4 * 4)[i],
                       b = reinterpret_cast<const uint32_t*>(src + hw4 * 8)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            dst[i * 4 + k] = make_float4(__fdiv_rn((float)((r >> (8 * k)) & 255u), divisor), __fdiv_rn((float)((g >> (8 * k)) & 255u), divisor),
                                         __fdiv_rn((float)((b >> (8 * k)) & 255u), divisor), 0.f);
    }
}

// uint8 planes -> packed uint8 RGBX pixels (the frames stay bytes; the reprojection kernels convert in registers): each thread packs 4
// consecutive pixels from three dword loads into one 16-byte store
struct U8Frames48 { const uint8_t* f[48]; };
MGN_PLAN_RO(U8Frames48, MGN_RO(f))
__global__ __launch_bounds__(256) void u8_frames_to_rgbx(U8Frames48 fr, long hw4, uint4* __restrict__ out) {
    const uint8_t* src = fr.f[blockIdx.y];
    uint4* dst = out + (long)blockIdx.y * hw4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw4; i += (long)gridDim.x * 256) {
        const uint32_t r = reinterpret_cast<const uint32_t*>(src)[i], g = reinterpret_cast<const uint32_t*>(src + hw4 * 4)[i],
                       b = reinterpret_cast<const uint32_t*>(src + hw4 * 8)[i];
        uint32_t px[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            px[k] = ((r >> (8 * k)) & 255u) | (((g >> (8 * k)) & 255u) << 8) | (((b >> (8 * k)) & 255u) << 16);
        dst[i] = make_uint4(px[0], px[1], px[2], px[3]);
    }
}

}  // namespace

#ifndef MGN_F16
extern "C" int mgn_u8_frames_to_rgbx(const void* const* frames_u8, int n_frames, long hw, void* out_u8, void* stream) {
    if (!frames_u8 || n_frames < 1 || n_frames > 48 || hw < 4 || hw % 4 || !out_u8 || ((uintptr_t)out_u8 & 15)) return MGN_EINVAL;
    U8Frames48 fr;
    for (int i = 0; i < 48; ++i) {
        fr.f[i] = i < n_frames ? (const uint8_t*)frames_u8[i] : nullptr;
        if (i < n_frames && (!fr.f[i] || ((uintptr_t)fr.f[i] & 3))) return MGN_EINVAL;
    }
    const long hw4 = hw / 4;
    long bx = (hw4 + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(u8_frames_to_rgbx, dim3((unsigned)bx, (unsigned)n_frames), dim3(256), 0, (hipStream_t)stream, fr, hw4, (uint4*)out_u8);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

extern "C" int mgn_u8_frames_to_f32_nhwc4(const void* const* frames_u8, int n_frames, long hw, float divisor, float* out, void* stream) {
    if (!frames_u8 || n_frames < 1 || n_frames > 16 || hw < 4 || hw % 4 || !(divisor != 0.f) || !out) return MGN_EINVAL;
    U8Frames fr;
    for (int i = 0; i < 16; ++i) {
        fr.f[i] = i < n_frames ? (const uint8_t*)frames_u8[i] : nullptr;
        if (i < n_frames && (!fr.f[i] || ((uintptr_t)fr.f[i] & 3))) return MGN_EINVAL;
    }
    const long hw4 = hw / 4;
    long bx = (hw4 + 255) / 256;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(u8_frames_to_f32_nhwc4, dim3((unsigned)bx, (unsigned)n_frames), dim3(256), 0, (hipStream_t)stream, fr, hw4, divisor, out);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

extern "C" int mgn_u8_frames_to_f32(const void* const* frames_u8, int n_frames, long n_per_frame, float divisor, float* out,
                                    void* stream) {
    if (!frames_u8 || n_frames < 1 || n_frames > 16 || n_per_frame < 16 || n_per_frame % 16 || !(divisor != 0.f) || !out) return MGN_EINVAL;
    U8Frames fr;
    for (int i = 0; i < 16; ++i) {
        fr.f[i] = i < n_frames ? (const uint8_t*)frames_u8[i] : nullptr;
        if (i < n_frames && (!fr.f[i] || ((uintptr_t)fr.f[i] & 15))) return MGN_EINVAL;
    }
    const long n16 = n_per_frame / 16;
    long bx = (n16 + 255) / 256;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(u8_frames_to_f32, dim3((unsigned)bx, (unsigned)n_frames), dim3(256), 0, (hipStream_t)stream, fr, n16, divisor, out);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
#endif

extern "C" int MGN_SYM(mgn_prep_input)(const void* const* frames_u8, int n_frames, int B, int H, int W, const float* pixel_mean3,
                              const float* pixel_std3, void* out_bf16, int Cp, void* stream) {
    if (!frames_u8 || n_frames < 1 || n_frames > 3 || B < 1 || H < 1 || W < 1 || !pixel_mean3 || !pixel_std3 || !out_bf16) return MGN_EINVAL;
    if ((Cp != 4 && Cp != 8 && Cp != 16) || n_frames * 3 > Cp) return MGN_EINVAL;
    PrepParams p;
    for (int f = 0; f < 3; ++f) p.frames[f] = f < n_frames ? (const uint8_t*)frames_u8[f] : nullptr;
    for (int c = 0; c < 3; ++c) {  // pixel_mean / pixel_std are in the 0..1 domain (mg_net.py:86-91: cfg value / 255)
        p.scale[c] = 1.0f / (255.0f * pixel_std3[c]);
        p.shift[c] = -pixel_mean3[c] / pixel_std3[c];
    }
    p.nf = n_frames; p.B = B; p.H = H; p.W = W; p.Cp = Cp; p.out = (uint16_t*)out_bf16;
    const long n = (long)B * H * W;
    hipLaunchKernelGGL(prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

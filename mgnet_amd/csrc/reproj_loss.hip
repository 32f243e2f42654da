// reproj_loss.hip -- MGNet's self-supervised photometric reprojection loss for MI355X (gfx950, wave64).
//
// Replaces (reference file:line): mgnet/modeling/loss.py:111-294 MultiViewPhotometricLoss and everything it
// calls in mgnet/geometry: camera.py:107-182 (reconstruct/project), camera_utils.py:24-55 (view_synthesis ->
// F.grid_sample bilinear/zeros/align_corners), pose.py:40-95 + pose_utils.py:9-59 (euler pose), depth.py:11-51
// (inv2depth, smoothness), image.py:42-69 (gradients).  The reference runs ~5,100 ATen ops and ~8 KB/px of
// intermediates per fwd+bwd; here one pass over the inputs produces the two losses AND the pixel-wise
// photometric gradient, and a light streaming pass finishes the backward.
//
// Kernel structure ("row march"):
//   * one wavefront owns a strip of 60 image columns (64 lanes = 60 + 2x2 halo) and RH rows and walks down
//     the rows; a lane IS a pixel column, so every global load of a row is one coalesced 256-byte request
//   * 3x3 SSIM windows and the 3x3 adjoint windows of the backward are separable sums: vertical part in
//     registers (the lane keeps the two previous rows), horizontal part with DPP wave shifts
//     (v_add_f32_dpp wave_shr:1 / wave_shl:1) -- no LDS traffic for the stencils
//   * the software pipeline per row t:  R(t) sample/warp  ->  S(t-1) SSIM, min/automask, loss, adjoint
//     coefficients  ->  G(t-2) 3x3 adjoint, bilinear/projective chain rule, d/d inv-depth, pose partial sums.
//     Per-pixel state that must survive two rows (bilinear derivatives etc.) lives in a wave-private LDS ring.
//   * F.pad(reflect) is realised by loading reflected rows/columns into the halo; its adjoint becomes a
//     weight of 2 on the neighbour next to the border (see stage G).
//   * geometry: K.(R.(Kinv.[u,v,1].d)+t) is folded to d.(M.[u,v,1]) + K.t with M = K.R.Kinv prepared per image,
//     which costs 3 FMAs per pixel and context instead of three 3x3 mat-vecs.
//   * no atomics: per-block partial sums + two tiny finalize kernels (fp64) => bitwise run-to-run determinism.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgnet_hip.h"

namespace {

constexpr int WAVE = 64;
constexpr int HALO = 2;
constexpr int STRIP = WAVE - 2 * HALO;  // 60 owned columns per wavefront
constexpr int NACC = 32;                // accumulators per (block, scale)
constexpr int NSTATE = 19;              // floats of per-pixel state kept for two rows (per scale wave)
constexpr int NSHARE = 11;              // floats per pixel and row published by the image wave
constexpr int RING = 4;                 // rows t+1 (being gathered) .. t-2

// accumulator slots
constexpr int A_PSUM = 0, A_SX = 1, A_SY = 2, A_SINV = 3, A_NMASK = 4, A_NMX = 5, A_NMY = 6, A_POSE = 8;  // 8..31: [j][12]

constexpr float SSIM_C1 = 1e-4f, SSIM_C2 = 9e-4f;

struct CamConst {      // per image
    float M[2][9];     // K.R_j.Kinv
    float Kt[2][3];    // K.t_j
    float K[9];
    float Kinv[9];
};

struct Stats {         // per (image, scale), written by finalize, read by the backward kernel
    float inv_mc, cxs, cys, dmean;
};

struct Params {
    const float* inv[MGN_MAX_SCALES];
    float* ginv[MGN_MAX_SCALES];
    const void* img;      // frames: fp32 planes / fp32 RGBx / uint8 RGBX according to the kernel's FMT
    const void* prev;
    const void* nxt;
    const uint8_t* mask;
    const CamConst* cam;
    float* partials;
    float* dbg;
    int B, H, W, n, RH, nseg, nstrips;
    float ssim_w;
    int automask;     // loss.py:139-144: the un-warped context frames compete in the per-pixel min (reference default: 1)
    int reduce_mean;  // loss.py:242-243 photometric_reduce_op "mean": mean over the warped maps instead of their min (automask must be 0)
    int pad_mode;     // F.grid_sample padding_mode of the warp (camera_utils.py:24-55): 0 "zeros", 1 "border", 2 "reflection"
};
MGN_PLAN_RO(Params, MGN_RO(inv) MGN_RO(img) MGN_RO(prev) MGN_RO(nxt) MGN_RO(mask) MGN_RO(cam))   // pointers the kernels only read through (launch-plan dependency analysis, csrc/mgn_launch.h)

// bound_ctrl=1: lanes without a source read 0, so no "old" operand has to be materialised (saves a v_mov per shift)
__device__ __forceinline__ float dpp_from_left(float x) {  // lane i <- lane i-1 (lane 0 <- 0)
    const int v = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_right(float x) {  // lane i <- lane i+1 (lane 63 <- 0)
    const int v = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float hsum3(float v) { return dpp_from_left(v) + v + dpp_from_right(v); }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ int reflect_clamp(int i, int n) {
    i = i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i);
    return min(max(i, 0), n - 1);
}

// loss.py:200-220 on window SUMS (9 taps).  Returns clamp((1-ssim)/2,0,1); if GRAD also the coefficients of
// d ssim / d x[r] = alpha + beta*y[r] + gamma*x[r]  (valid for every tap r of the window).
// Exactness contract: when x == y bitwise over the window, every intermediate below is bitwise symmetric in x and
// y (explicit fmaf pattern for the products, no compiler contraction inside ssim_from_sums), so ssim == 1 and the
// automask term of a static pixel is exactly 0 -- as in the reference, where the torch ops are rounded one by one.
__device__ __forceinline__ float dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
    return fmaf(a0, b0, fmaf(a1, b1, a2 * b2));
}

template <bool GRAD>
__device__ __forceinline__ float ssim_from_sums(float Sx, float Sy, float Sxx, float Syy, float Sxy, float& alpha,
                                                float& beta, float& gamma, bool& gate) {
#pragma clang fp contract(off)
    constexpr float r9 = 1.0f / 9.0f;
    const float mux = Sx * r9, muy = Sy * r9;
    const float mxx = mux * mux, myy = muy * muy, mxy = mux * muy;
    const float sgx = Sxx * r9 - mxx, sgy = Syy * r9 - myy, sgxy = Sxy * r9 - mxy;
    const float N1 = 2.f * mxy + SSIM_C1, N2 = 2.f * sgxy + SSIM_C2;
    const float D1 = mxx + myy + SSIM_C1, D2 = sgx + sgy + SSIM_C2;
    // quotient = v_rcp_f32 + one Newton step: exactly 1 when numerator == denominator bitwise, so that
    // ssim(x,x) == 1 and the automask term of identical frames is exactly 0 like the reference's true division
    const float num = N1 * N2, den = D1 * D2;
    const float i12 = frcp(den);
    const float q0 = num * i12;
    const float s = fmaf(fmaf(-den, q0, num), i12, q0);
    const float val = (1.f - s) * 0.5f;
    if (GRAD) {
        gate = (val >= 0.f) && (val <= 1.f);  // torch.clamp backward passes at the bounds
        const float iD1 = D2 * i12, iD2 = D1 * i12;
        constexpr float c29 = 2.0f / 9.0f;
        alpha = c29 * (muy * (N2 - N1) * i12 - s * mux * (iD1 - iD2));
        beta = c29 * N1 * i12;
        gamma = -c29 * s * iD2;
    }
    return fminf(fmaxf(val, 0.f), 1.f);
}

// The same for BOTH context frames at once (prev, next share the target's sums Sy, Syy): the element-wise algebra on
// 2-vectors compiles to packed fp32 (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two frames per VALU issue).  Every component
// goes through exactly the scalar function's operation sequence (contraction off, explicit fma in the Newton step), so the
// values -- and the exactness contract above -- are unchanged.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool GRAD>
__device__ __forceinline__ f32x2 ssim_from_sums2(f32x2 Sx, float Sy, f32x2 Sxx, float Syy, f32x2 Sxy, f32x2& alpha, f32x2& beta,
                                                 f32x2& gamma, bool& gate0, bool& gate1) {
#pragma clang fp contract(off)
    constexpr float r9 = 1.0f / 9.0f;
    const f32x2 mux = Sx * r9;
    const float muy = Sy * r9;
    const f32x2 mxx = mux * mux, mxy = mux * muy;
    const float myy = muy * muy;
    const f32x2 sgx = Sxx * r9 - mxx, sgxy = Sxy * r9 - mxy;
    const float sgy = Syy * r9 - myy;
    const f32x2 N1 = 2.f * mxy + SSIM_C1, N2 = 2.f * sgxy + SSIM_C2;
    const f32x2 D1 = mxx + myy + SSIM_C1, D2 = sgx + sgy + SSIM_C2;
    const f32x2 num = N1 * N2, den = D1 * D2;
    f32x2 i12;
    i12.x = frcp(den.x);
    i12.y = frcp(den.y);
    const f32x2 q0 = num * i12;
    const f32x2 s = __builtin_elementwise_fma(__builtin_elementwise_fma(-den, q0, num), i12, q0);
    const f32x2 val = (1.f - s) * 0.5f;
    if (GRAD) {
        gate0 = (val.x >= 0.f) && (val.x <= 1.f);
        gate1 = (val.y >= 0.f) && (val.y <= 1.f);
        const f32x2 iD1 = D2 * i12, iD2 = D1 * i12;
        constexpr float c29 = 2.0f / 9.0f;
        alpha = c29 * (muy * (N2 - N1) * i12 - s * mux * (iD1 - iD2));
        beta = c29 * N1 * i12;
        gamma = -c29 * s * iD2;
    }
    f32x2 r;
    r.x = fminf(fmaxf(val.x, 0.f), 1.f);
    r.y = fminf(fmaxf(val.y, 0.f), 1.f);
    return r;
}

// F.grid_sample(bilinear, zeros, align_corners=True) at pixel position (ix,iy) for the 3 channel planes of a context frame,
// plus d out_c / d ix and d out_c / d iy.  The planes are addressed through buffer resources (base in SGPRs, a 32-bit byte
// offset per lane: no 64-bit address arithmetic) whose range check supplies the zero padding: a corner outside the image in
// y is outside the plane's byte range by itself, a corner outside in x gets an out-of-range offset, and the hardware returns
// 0 for both -- no clamping of the corner coordinates and no select per loaded value.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, int voff_bytes, int soff_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff_bytes, soff_bytes, 0));
}

// Frame formats (mgn_reproj_cfg.frame_layout):
//   0  the reference's tensors: target and context frames [B,3,H,W] fp32 planes
//   1  context frames pixel-interleaved fp32 RGBx [B,H,W,4] (target planar): one 16-byte gather per bilinear corner
//   2  all three frames as the uint8 RGBX [B,H,W,4] the step receives them in (mg_net.py:320-335 computes `uint8.float() / 255` on the
//      way in): ONE dword per pixel -- a bilinear corner is one 4-byte gather instead of three, a target / context row one coalesced
//      dword load instead of three, 4 B/px of HBM traffic per frame instead of 12 -- converted in registers.
constexpr int FMT_PLANAR = 0, FMT_RGBX_F32 = 1, FMT_RGBX_U8 = 2;

// byte k of a packed RGBX pixel -> byte / 255 exactly as the reference's fp32 division rounds it, for every one of the 256 values
// (tests/test_reproj_gpu.py checks all of them): v_cvt_f32_ubyte<k>, then the quotient as a two-term product with 1/255 split into
// fp32 high and low parts -- b * hi alone is 1 ulp off for 126 of the 256 values, fma(b, hi, b * lo) for none.
template <int K>
__device__ __forceinline__ float u8unit(uint32_t w) {
    constexpr float HI = (float)(1.0 / 255.0);
    constexpr float LO = (float)(1.0 / 255.0 - (double)HI);
    const float b = (float)((w >> (8 * K)) & 255u);
    return fmaf(b, HI, b * LO);
}
__device__ __forceinline__ void u8unit3(uint32_t w, float out[3]) {
    out[0] = u8unit<0>(w); out[1] = u8unit<1>(w); out[2] = u8unit<2>(w);
}

template <int FMT>
struct Gather {       // the 12 corner values of one sample position (3 channel planes) and its fractional offsets
    float v[3][4];   // [channel][00, 10, 01, 11]
    float tx, ty;
    float mx, my;    // d (padded position) / d (projected position); only touched by the PAD instantiations
};
template <>
struct Gather<FMT_RGBX_U8> {   // the four corners as packed pixels: 4 registers carried across the row instead of 12
    uint32_t w[4];   // [00, 10, 01, 11]
    float tx, ty;
    float mx, my;
};

// F.grid_sample padding modes for align_corners=True (ATen GridSampler: reflect over [0, 2 (size-1)], then clip; the gradient factor of
// the transform is 0 where the position was clipped, -1 on a reflected branch).  "reflection" folds |x| modulo 2 (size-1) with a
// truncated division like the reference's CPU kernel (not an exact fmod); positions far outside the image fold chaotically in fp32
// whatever the formula (SURVEY: points behind the camera), there the reference's value is noise as well.
__device__ __forceinline__ float pad_coord(float in, int size, int mode, float& mult) {
#pragma clang fp contract(off)
    mult = 1.f;
    const float hi = (float)(size - 1);
    if (mode == 2 && size > 1) {
        const float ts = 2.f * hi;
        const bool neg = in < 0.f;
        const float a = fabsf(in);
        const float df = truncf(a / ts);
        const float extra = a - df * ts;
        const float refl = ts - extra;
        const bool flip = extra > refl;
        in = flip ? refl : extra;
        mult = (flip != neg) ? -1.f : 1.f;
    }
    if (in <= 0.f) { mult = 0.f; return 0.f; }
    if (in >= hi) { mult = 0.f; return hi; }
    return in;
}
// first half: corner addresses and the 12 loads (nothing waits for them here)
// ILV: the context frame is pixel-interleaved RGBx ([H][W][4] fp32 = a 4-channel channels_last torch tensor, 4th channel unused):
// ONE 16-byte load per corner through one resource instead of three dword loads through three plane resources -- the same 12
// values, a third of the vector-memory instructions (the kernel's second limiter after VALU issue).  (The compiler
// narrows the 16-byte load to buffer_load_dwordx3 since only three elements are used; the 16-byte pixel keeps every load aligned.)
template <int FMT, bool PAD = false>
__device__ __forceinline__ void bilinear3_issue(const rsrc_t (&plane)[3], int W, int H, float ix, float iy, Gather<FMT>& g, int pad_mode = 0) {
    constexpr bool ILV = FMT == FMT_RGBX_F32;
    if constexpr (PAD) {   // (a compile-time variant: the default "zeros" instantiations stay instruction for instruction what they were)
        ix = pad_coord(ix, W, pad_mode, g.mx);
        iy = pad_coord(iy, H, pad_mode, g.my);
    }
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    g.tx = ix - fx0;
    g.ty = iy - fy0;
    const int xi = (int)__builtin_amdgcn_fmed3f(fx0, -2.f, (float)W);   // NaN -> -2: every corner out of range
    const int yi = (int)__builtin_amdgcn_fmed3f(fy0, -2.f, (float)H);
    const bool x0ok = (unsigned)xi < (unsigned)W, x1ok = (unsigned)(xi + 1) < (unsigned)W;
    constexpr int OUT = (int)0x80000000u;      // beyond any plane (H*W <= 2^28 pixels)
    constexpr int PX = ILV ? 16 : 4;           // bytes from a pixel to its right neighbour
    const int o00 = (yi * W + xi) * PX;
    const int a00 = x0ok ? o00 : OUT, a10 = x1ok ? o00 + PX : OUT;
    const int a01 = x0ok ? o00 + PX * W : OUT, a11 = x1ok ? o00 + PX * W + PX : OUT;
    if constexpr (FMT == FMT_RGBX_U8) {
        g.w[0] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(plane[0], a00, 0, 0);
        g.w[1] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(plane[0], a10, 0, 0);
        g.w[2] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(plane[0], a01, 0, 0);
        g.w[3] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(plane[0], a11, 0, 0);
    } else if constexpr (ILV) {
        // (the result goes through a float vector and .x/.y/.z: indexing an unsigned ext-vector of this builtin's result and
        //  bit-casting the element is miscompiled by this toolchain into ONE dword load splatted over the elements)
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 c00 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(plane[0], a00, 0, 0));
        const f32x4 c10 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(plane[0], a10, 0, 0));
        const f32x4 c01 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(plane[0], a01, 0, 0));
        const f32x4 c11 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(plane[0], a11, 0, 0));
        g.v[0][0] = c00.x; g.v[1][0] = c00.y; g.v[2][0] = c00.z;
        g.v[0][1] = c10.x; g.v[1][1] = c10.y; g.v[2][1] = c10.z;
        g.v[0][2] = c01.x; g.v[1][2] = c01.y; g.v[2][2] = c01.z;
        g.v[0][3] = c11.x; g.v[1][3] = c11.y; g.v[2][3] = c11.z;
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            g.v[c][0] = bload(plane[c], a00, 0);
            g.v[c][1] = bload(plane[c], a10, 0);
            g.v[c][2] = bload(plane[c], a01, 0);
            g.v[c][3] = bload(plane[c], a11, 0);
        }
    }
}
// second half: blend
template <bool GRAD, int FMT>
__device__ __forceinline__ void bilinear3_finish(const Gather<FMT>& g, float out[3], float ex[3], float ey[3]) {
    const float tx = g.tx, ty = g.ty, sx = 1.f - tx, sy = 1.f - ty;
    const float w00 = sx * sy, w10 = tx * sy, w01 = sx * ty, w11 = tx * ty;
    float cv[4][3];
    if constexpr (FMT == FMT_RGBX_U8) {
#pragma unroll
        for (int k = 0; k < 4; ++k) u8unit3(g.w[k], cv[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) cv[k][c] = g.v[c][k];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v00 = cv[0][c], v10 = cv[1][c], v01 = cv[2][c], v11 = cv[3][c];
        out[c] = v00 * w00 + v10 * w10 + v01 * w01 + v11 * w11;
        if (GRAD) {
            ex[c] = (v10 - v00) * sy + (v11 - v01) * ty;
            ey[c] = (v01 - v00) * sx + (v11 - v10) * tx;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// prep: per image constants.  pose_utils.py:9-51 (R = Rx.Ry.Rz, t = vec[:3]); camera.py:72-81 (Kinv is a
// clone of K with 4 entries replaced).  Products in fp64, trig in fp32 like the reference.
// ---------------------------------------------------------------------------------------------------
__device__ void euler_mats(const float* ang, double X[9], double Y[9], double Z[9]) {
    const double cx = cosf(ang[0]), sx = sinf(ang[0]), cy = cosf(ang[1]), sy = sinf(ang[1]), cz = cosf(ang[2]), sz = sinf(ang[2]);
    const double x[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
    const double y[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
    const double z[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
    for (int k = 0; k < 9; ++k) { X[k] = x[k]; Y[k] = y[k]; Z[k] = z[k]; }
}
__device__ void mm3(const double* a, const double* b, double* o) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

__global__ void reproj_prep(const float* cam, int cam_stride, int cam_ld, const float* pose, int B, CamConst* out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double K[9], Ki[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) K[r * 3 + c] = cam[(size_t)b * cam_stride + r * cam_ld + c];
    for (int k = 0; k < 9; ++k) Ki[k] = K[k];
    const float fx = (float)K[0], fy = (float)K[4], cx = (float)K[2], cy = (float)K[5];
    Ki[0] = 1.0f / fx;
    Ki[4] = 1.0f / fy;
    Ki[2] = -1.0f * cx / fx;
    Ki[5] = -1.0f * cy / fy;
    CamConst cc;
    for (int k = 0; k < 9; ++k) { cc.K[k] = (float)K[k]; cc.Kinv[k] = (float)Ki[k]; }
    for (int j = 0; j < 2; ++j) {
        const float* v = pose + ((size_t)b * 2 + j) * 6;
        double X[9], Y[9], Z[9], XY[9], R[9], KR[9], M[9];
        euler_mats(v + 3, X, Y, Z);
        mm3(X, Y, XY);
        mm3(XY, Z, R);
        for (int k = 0; k < 9; ++k) R[k] = (double)(float)R[k];  // the reference holds R in fp32
        mm3(K, R, KR);
        mm3(KR, Ki, M);
        for (int k = 0; k < 9; ++k) cc.M[j][k] = (float)M[k];
        for (int r = 0; r < 3; ++r) cc.Kt[j][r] = (float)(K[r * 3] * v[0] + K[r * 3 + 1] * v[1] + K[r * 3 + 2] * v[2]);
    }
    out[b] = cc;
}

// ---------------------------------------------------------------------------------------------------
// the row-march kernel.  Block = 1 "image" wavefront + n_scales "scale" wavefronts that walk down the SAME
// 60-column strip in lock step (one barrier per row):
//   image wave  : everything that depends on the images only -- window sums of the target, the un-warped
//                 (automask) photometric maps, edge-aware smoothness weights, mask -- computed once and published
//                 through LDS to the scale waves (it runs one row ahead)
//   scale wave i: warp with depth scale i, SSIM vs target, per-pixel min, loss sums and the adjoint.
// All waves of a block touch the same image rows at the same time, so the inputs stream from HBM once.
// ---------------------------------------------------------------------------------------------------
constexpr int SH_SY = 0, SH_SYY = 3, SH_PU = 6, SH_WX = 8, SH_WY = 9, SH_LIVE = 10;

// The per-row barrier of the march kernel.  Only LDS traffic is exchanged between the waves, so only the LDS counter is
// drained: __syncthreads() also waits for every outstanding global load (vmcnt(0)), which would end the prefetch of the next
// row's gathers at each row.
__device__ __forceinline__ void row_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool GRAD, int FMT = FMT_PLANAR, bool PAD = false, bool L1MIN = false>
__global__ __launch_bounds__(WAVE*(MGN_MAX_SCALES + 1)) void reproj_march(Params p) {
    constexpr bool ILV = FMT == FMT_RGBX_F32, U8 = FMT == FMT_RGBX_U8;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x >> 6;  // 0 = image wave, 1..n = scale waves
    // XCD-aware tile order.  The hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own L2);
    // with tile = blockIdx the horizontally adjacent strips of a row segment -- which share 128-byte lines, because a strip is
    // 240 bytes wide and starts 8 bytes before a multiple of 240 -- land on 8 different L2s and every shared line is fetched
    // once per XCD (measured: 1.81 GB read per launch against 0.82 GB of input).  Here the tiles are dealt in groups of
    // 8 row segments: XCD k takes ALL strips of the k-th segment of the group, so neighbours in x share an L2 while the
    // 8 XCDs still walk through the same part of the frame together (one contiguous eighth of the tile list per XCD removed
    // the re-reads as well but ran 6 % slower: eight frames in flight at once).
    int bid = blockIdx.x;
    {
        const int G = p.nstrips, super = 8 * G, first = (bid / super) * super;
        if (first + super <= (int)gridDim.x) {   // (the last, partial group keeps the identity order)
            const int r = bid - first;
            bid = first + (r & 7) * G + (r >> 3);
        }
    }
    const int strip = bid % p.nstrips;
    const int seg = (bid / p.nstrips) % p.nseg;
    const int b = bid / (p.nstrips * p.nseg);

    const int H = p.H, W = p.W, HWp = H * W;
    const int cu = strip * STRIP - HALO + lane;
    const bool col_in = (cu >= 0) && (cu < W);
    const int ucol = reflect_clamp(cu, W);
    const bool lane_own = col_in && lane >= HALO && lane < WAVE - HALO;
    const int r0 = seg * p.RH;
    const int rend = min(r0 + p.RH, H);
    const float fu = (float)ucol;
    const float* imgb = (const float*)p.img + (size_t)b * 3 * HWp;            // (planar fp32 target)
    const uint32_t* img32 = (const uint32_t*)p.img + (size_t)b * HWp;          // (U8: one packed pixel per dword)
    const float ssim_w = p.ssim_w;
    const float l1_w3 = (1.f - ssim_w) * (1.f / 3.f), ssim_w3 = ssim_w * (1.f / 3.f);
    float* share = smem + (size_t)p.n * RING * NSTATE * WAVE + lane;  // [2][NSHARE][WAVE]

    if (wave == 0) {
        // =========================================== image wave ===========================================
        const float* refb[2] = {(const float*)p.prev + (size_t)b * (ILV ? 4 : 3) * HWp, (const float*)p.nxt + (size_t)b * (ILV ? 4 : 3) * HWp};
        const uint32_t* ref32[2] = {(const uint32_t*)p.prev + (size_t)b * HWp, (const uint32_t*)p.nxt + (size_t)b * HWp};
        const uint8_t* maskb = p.mask ? p.mask + (size_t)b * HWp : nullptr;
        float y1[3] = {0.f, 0.f, 0.f}, y2[3] = {0.f, 0.f, 0.f}, rf1[2][3], rf2[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) rf1[j][c] = rf2[j][c] = 0.f;
        bool m1 = false;
        float nmask = 0.f, nmx = 0.f, nmy = 0.f;
        for (int s = r0 - 3; s <= rend + 1; ++s) {
            if (s <= rend) {  // wave-uniform
                const int vrow = reflect_clamp(s + 1, H);
                const int off = vrow * W + ucol;
                float y0[3], rf0[2][3];
                if constexpr (U8) {   // three coalesced dword loads per row instead of nine
                    const uint32_t wy = img32[off], w0 = ref32[0][off], w1 = ref32[1][off];
                    u8unit3(wy, y0);
                    u8unit3(w0, rf0[0]);
                    u8unit3(w1, rf0[1]);
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        y0[c] = imgb[c * HWp + off];
                        rf0[0][c] = ILV ? refb[0][off * 4 + c] : refb[0][c * HWp + off];
                        rf0[1][c] = ILV ? refb[1][off * 4 + c] : refb[1][c * HWp + off];
                    }
                }
                const bool m0 = maskb ? (maskb[off] != 0) : true;
                if (s >= r0 - 1) {
                    const int q = s;
                    float* sh = share + (size_t)(s & 1) * NSHARE * WAVE;
                    float pu[2] = {0.f, 0.f};
                    float igx = 0.f, igy = 0.f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float Sy = hsum3(y2[c] + y1[c] + y0[c]);
                        const float Syy = hsum3(dot3(y2[c], y1[c], y0[c], y2[c], y1[c], y0[c]));
                        sh[(SH_SY + c) * WAVE] = Sy;
                        sh[(SH_SYY + c) * WAVE] = Syy;
                        {   // un-warped context images vs target (automask, loss.py:139-144), both frames at once
                            f32x2 Sx, Sxx, Sxy, d0, d1, d2;
                            bool g0, g1;
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const float a = rf2[j][c], bq = rf1[j][c], cc = rf0[j][c];
                                Sx[j] = hsum3(a + bq + cc);
                                Sxx[j] = hsum3(dot3(a, bq, cc, a, bq, cc));
                                Sxy[j] = hsum3(dot3(a, bq, cc, y2[c], y1[c], y0[c]));
                            }
                            if constexpr (L1MIN) {   // ssim_loss_weight = 0, reduce "min": the map is the per-pixel minimum of the three L1 channels
                                const float e0 = fabsf(rf1[0][c] - y1[c]), e1 = fabsf(rf1[1][c] - y1[c]);
                                pu[0] = (c == 0 || e0 < pu[0]) ? e0 : pu[0];
                                pu[1] = (c == 0 || e1 < pu[1]) ? e1 : pu[1];
                            } else {
                                const f32x2 v = ssim_from_sums2<false>(Sx, Sy, Sxx, Syy, Sxy, d0, d1, d2, g0, g1);
                                pu[0] += ssim_w3 * v.x + l1_w3 * fabsf(rf1[0][c] - y1[c]);
                                pu[1] += ssim_w3 * v.y + l1_w3 * fabsf(rf1[1][c] - y1[c]);
                            }
                        }
                        igx += fabsf(y1[c] - dpp_from_right(y1[c]));
                        igy += fabsf(y1[c] - y0[c]);
                    }
                    const bool row_in = (q >= 0) && (q < H);
                    const bool live = row_in && col_in && m1;
                    const bool hasx = cu + 1 < W, hasy = q + 1 < H;
                    // smoothness weights exp(-mean_c |d img|) (depth.py:24-25), pre-masked (loss.py:284-285)
                    sh[SH_PU * WAVE] = pu[0];
                    sh[(SH_PU + 1) * WAVE] = pu[1];
                    sh[SH_WX * WAVE] = (live && hasx) ? __expf(-igx * (1.f / 3.f)) : 0.f;
                    sh[SH_WY * WAVE] = (live && hasy) ? __expf(-igy * (1.f / 3.f)) : 0.f;
                    sh[SH_LIVE * WAVE] = live ? 1.f : 0.f;
                    if (lane_own && m1 && q >= r0 && q < rend) {
                        nmask += 1.f;
                        if (hasx) nmx += 1.f;
                        if (hasy) nmy += 1.f;
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    y2[c] = y1[c]; y1[c] = y0[c];
                    rf2[0][c] = rf1[0][c]; rf1[0][c] = rf0[0][c];
                    rf2[1][c] = rf1[1][c]; rf1[1][c] = rf0[1][c];
                }
                m1 = m0;
            }
            row_barrier();
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            nmask += __shfl_xor(nmask, o);
            nmx += __shfl_xor(nmx, o);
            nmy += __shfl_xor(nmy, o);
        }
        if (lane == 0) {
            float* pp = p.partials + (size_t)bid * p.n * NACC;
            pp[A_NMASK] = nmask; pp[A_NMX] = nmx; pp[A_NMY] = nmy; pp[7] = 0.f;
        }
        return;
    }

    // ============================================= scale wave =============================================
    const int i = wave - 1;
    const bool active = i < p.n;  // wave-uniform (blockDim is 64*(n+1), so always true; kept for safety)
    const CamConst& cam = p.cam[b];
    // a_j(u,v) = M_j.[u,v,1] = base_j + col1_j * v
    float base[2][3], col1[2][3], kt[2][3];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            base[j][k] = cam.M[j][k * 3] * fu + cam.M[j][k * 3 + 2];
            col1[j][k] = cam.M[j][k * 3 + 1];
            kt[j][k] = cam.Kt[j][k];
        }
    // (U8: a frame is HWp dwords = what one fp32 plane is, so the pointer arithmetic below is in dwords for every format)
    const float* refb[2] = {(const float*)p.prev + (size_t)b * (ILV ? 4 : (U8 ? 1 : 3)) * HWp,
                            (const float*)p.nxt + (size_t)b * (ILV ? 4 : (U8 ? 1 : 3)) * HWp};
    // (ILV / U8: one resource over the whole interleaved frame, 16 / 4 bytes per pixel; the other two entries are unused)
    const uint32_t pbytes = ILV ? 16u * HWp : 4u * HWp;
    const int pstep = (ILV || U8) ? 0 : HWp;
    const rsrc_t plane[2][3] = {{make_rsrc(refb[0], pbytes), make_rsrc(refb[0] + pstep, pbytes), make_rsrc(refb[0] + 2 * pstep, pbytes)},
                                {make_rsrc(refb[1], pbytes), make_rsrc(refb[1] + pstep, pbytes), make_rsrc(refb[1] + 2 * pstep, pbytes)}};
    // weights realising the adjoint of F.pad(reflect): a border pixel's adjoint window is seen twice by its neighbour
    const float exp_to_right = (cu == 0) ? 2.f : 1.f;      // value exported to lane+1
    const float exp_to_left = (cu == W - 1) ? 2.f : 1.f;   // value exported to lane-1
    const float* invb = p.inv[active ? i : 0] + (size_t)b * HWp;
    float* gout = GRAD ? p.ginv[active ? i : 0] + (size_t)b * HWp : nullptr;
    float* dbg = p.dbg ? p.dbg + ((size_t)i * p.B + b) * HWp : nullptr;
    float* ringw = smem + (size_t)i * RING * NSTATE * WAVE + lane;

    float a_psum = 0.f, a_sx = 0.f, a_sy = 0.f, a_sinv = 0.f;
    float pacc[2][9];  // pose sums per context: sum s, sum s*row, sum dXc (the column factor fu is a per-lane constant)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 9; ++k) pacc[j][k] = 0.f;

    // rows t-1 (suffix 1) and t-2 (suffix 2)
    float xw1[2][3], xw2[2][3], y1[3] = {0.f, 0.f, 0.f}, y2[3] = {0.f, 0.f, 0.f};
    float inv1 = 0.f;
    float cA1[2][3], cB1[2][3], cC1[2][3], cA2[2][3], cB2[2][3], cC2[2][3], l1g1[2][3];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            xw1[j][c] = xw2[j][c] = 0.f;
            cA1[j][c] = cB1[j][c] = cC1[j][c] = cA2[j][c] = cB2[j][c] = cC2[j][c] = l1g1[j][c] = 0.f;
        }

    // Software pipeline of the gathers: the vector-memory path costs ~20 cycles per wave-instruction and the 24 corner loads
    // of a row were consumed right where they were issued, so VALU and memory phases of all waves of a block (in lock step
    // through the row barrier) alternated instead of overlapping -- with the loads removed the kernel takes 1.4 ms, with
    // them 2.5.  Now iteration t blends the corners of row t (issued one iteration earlier, carried in registers), then
    // issues the loads of row t+1, and the SSIM / adjoint stages run while those are in flight.
    Gather<FMT> gth[2];
    float y_pre[3] = {0.f, 0.f, 0.f};
    uint32_t y_pre32 = 0u;
    float inv_pre = invb[reflect_clamp(r0 - 2, H) * W + ucol];   // inverse depth of row t+1, loaded one iteration ahead
    float inv_cur = 0.f;                                         // ... of row t
    for (int t = r0 - 3; t <= rend + 1; ++t) {
        const int it = t - (r0 - 3);
        float y0[3] = {0.f, 0.f, 0.f}, xw0[2][3];
        // ------------------------------ stage R, second half: blend row t ------------------------------
        if (t >= r0 - 2) {
            float* st = ringw + (size_t)(it & (RING - 1)) * NSTATE * WAVE;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float ex[3], ey[3];
                bilinear3_finish<GRAD, FMT>(gth[j], xw0[j], ex, ey);
                if (GRAD) {
                    if constexpr (PAD) {   // chain rule through the padding transform of the sampling position
#pragma unroll
                        for (int c = 0; c < 3; ++c) { ex[c] *= gth[j].mx; ey[c] *= gth[j].my; }
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        st[(j * 9 + c) * WAVE] = ex[c];
                        st[(j * 9 + 3 + c) * WAVE] = ey[c];
                    }
                }
            }
            if constexpr (U8) {
                u8unit3(y_pre32, y0);
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) y0[c] = y_pre[c];
            }
        }
        const float inv0 = inv_cur;
        // ------------------------------ stage R, first half: geometry and loads of row t+1 ------------------------------
        if (t + 1 <= rend + 1) {  // wave-uniform
            const int vrow = reflect_clamp(t + 1, H);
            const int off = vrow * W + ucol;
            const float fv = (float)vrow;
            const float inv_n = inv_pre;
            inv_pre = invb[reflect_clamp(t + 2, H) * W + ucol];
            const float d = frcp(fmaxf(inv_n, 1e-6f));  // depth.py:15
            float* st = ringw + (size_t)((it + 1) & (RING - 1)) * NSTATE * WAVE;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float a0 = base[j][0] + col1[j][0] * fv, a1 = base[j][1] + col1[j][1] * fv, a2 = base[j][2] + col1[j][2] * fv;
                const float X = d * a0 + kt[j][0], Y = d * a1 + kt[j][1], z = d * a2 + kt[j][2];
                const bool zf = z >= 1e-5f;             // camera.py:172 clamp(min=1e-5)
                const float rz = frcp(fmaxf(z, 1e-5f));
                const float ix = X * rz, iy = Y * rz;   // == ((Xn+1)/2)(W-1) of grid_sample
                bilinear3_issue<FMT, PAD>(plane[j], W, H, ix, iy, gth[j], p.pad_mode);
                if (GRAD) {
                    st[(j * 9 + 6) * WAVE] = zf ? rz : -rz;  // rz > 0: the sign carries the clamp flag
                    st[(j * 9 + 7) * WAVE] = ix;
                    st[(j * 9 + 8) * WAVE] = iy;
                }
            }
            if (GRAD) st[18 * WAVE] = (inv_n >= 1e-6f) ? d : -d;  // d > 0: the sign carries "inverse depth not clamped"
            if constexpr (U8) {
                y_pre32 = img32[off];
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) y_pre[c] = imgb[c * HWp + off];
            }
            inv_cur = inv_n;
        }
        if (t >= r0 - 2 && active) {  // wave-uniform
            // ------------------------------ stage S: row q = t-1 ------------------------------
            float cA0[2][3], cB0[2][3], cC0[2][3], l1g0[2][3];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) cA0[j][c] = cB0[j][c] = cC0[j][c] = l1g0[j][c] = 0.f;
            if (t >= r0) {  // wave-uniform
                const int q = t - 1;
                const float* sh = share + (size_t)(q & 1) * NSHARE * WAVE;
                const bool row_own = (q >= r0) && (q < rend);
                float pw[2] = {0.f, 0.f};
                float al[2][3], be[2][3], ga[2][3];
                bool gt[2][3];
                int cwin[2] = {0, 0};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float Sy = sh[(SH_SY + c) * WAVE], Syy = sh[(SH_SYY + c) * WAVE];
                    {   // both warped context frames at once (packed fp32)
                        f32x2 Sx, Sxx, Sxy, a2, b2, g2;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float a = xw2[j][c], bq = xw1[j][c], cc = xw0[j][c];
                            Sx[j] = hsum3(a + bq + cc);
                            Sxx[j] = hsum3(dot3(a, bq, cc, a, bq, cc));
                            Sxy[j] = hsum3(dot3(a, bq, cc, y2[c], y1[c], y0[c]));
                        }
                        if constexpr (L1MIN) {   // first minimal channel wins (torch.min over the concatenated channels, loss.py:244)
                            const float e0 = fabsf(xw1[0][c] - y1[c]), e1 = fabsf(xw1[1][c] - y1[c]);
                            if (c == 0 || e0 < pw[0]) { pw[0] = e0; cwin[0] = c; }
                            if (c == 0 || e1 < pw[1]) { pw[1] = e1; cwin[1] = c; }
                            gt[0][c] = gt[1][c] = false;
                            al[0][c] = al[1][c] = be[0][c] = be[1][c] = ga[0][c] = ga[1][c] = 0.f;
                        } else {
                            const f32x2 v = ssim_from_sums2<GRAD>(Sx, Sy, Sxx, Syy, Sxy, a2, b2, g2, gt[0][c], gt[1][c]);
                            if (GRAD) {
                                al[0][c] = a2.x; al[1][c] = a2.y; be[0][c] = b2.x; be[1][c] = b2.y; ga[0][c] = g2.x; ga[1][c] = g2.y;
                            }
                            pw[0] += ssim_w3 * v.x + l1_w3 * fabsf(xw1[0][c] - y1[c]);
                            pw[1] += ssim_w3 * v.y + l1_w3 * fabsf(xw1[1][c] - y1[c]);
                        }
                    }
                }
                const float pu0 = sh[SH_PU * WAVE], pu1 = sh[(SH_PU + 1) * WAVE];
                const float wxm = sh[SH_WX * WAVE], wym = sh[SH_WY * WAVE];
                const bool live = sh[SH_LIVE * WAVE] != 0.f;
                // loss.py:241-246: min over [warp_prev, unwarp_prev, warp_next, unwarp_next]; first index wins ties.
                // Options (wave-uniform): without automasking only the warped maps compete; reduce "mean": their average, both carry
                // half of the gradient
                float best = pw[0];
                int win = 0;
                if (p.automask && pu0 < best) { best = pu0; win = 1; }
                if (pw[1] < best) { best = pw[1]; win = 2; }
                if (p.automask && pu1 < best) { best = pu1; win = 3; }
                const float gshare = p.reduce_mean ? 0.5f : 1.f;
                if (p.reduce_mean) best = 0.5f * (pw[0] + pw[1]);
                const bool own = row_own && lane_own;
                // smoothness (depth.py:18-51, loss.py:257-294), un-normalised: |d inv| * exp(-mean_c |d img|)
                const float sxv = fabsf(inv1 - dpp_from_right(inv1)) * wxm;
                const float syv = fabsf(inv1 - inv0) * wym;
                if (own) {
                    if (live) a_psum += best;
                    a_sinv += inv1;
                    a_sx += sxv;
                    a_sy += syv;
                    if (dbg) dbg[q * W + cu] = best;
                }
                if (GRAD) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const bool G = live && (p.reduce_mean || win == 2 * j);
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float k = (G && gt[j][c]) ? -0.5f * ssim_w3 * gshare : 0.f;
                            cA0[j][c] = k * al[j][c];
                            cB0[j][c] = k * be[j][c];
                            cC0[j][c] = k * ga[j][c];
                            const float df = xw1[j][c] - y1[c];
                            if constexpr (L1MIN) l1g0[j][c] = (G && c == cwin[j]) ? (float)((df > 0.f) - (df < 0.f)) : 0.f;
                            else l1g0[j][c] = G ? l1_w3 * gshare * (float)((df > 0.f) - (df < 0.f)) : 0.f;
                        }
                    }
                }
            }

            // ------------------------------ stage G: row r = t-2 ------------------------------
            if (GRAD && t >= r0 + 2 && t - 2 < rend) {  // wave-uniform
                const int r = t - 2;
                const float wu = (r == 1) ? 2.f : 1.f, wd = (r == H - 2) ? 2.f : 1.f;
                const float* sr = ringw + (size_t)((it + 2) & (RING - 1)) * NSTATE * WAVE;  // slot of row t-2
                const float ds = sr[18 * WAVE];
                const float dd_ = fabsf(ds), ddf_ = ds > 0.f ? -ds * ds : 0.f;  // d depth / d inv (0 where clamped)
                const float fr = (float)r;
                float ddsum = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float gix = 0.f, giy = 0.f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float vA = wu * cA2[j][c] + cA1[j][c] + wd * cA0[j][c];
                        const float vB = wu * cB2[j][c] + cB1[j][c] + wd * cB0[j][c];
                        const float vC = wu * cC2[j][c] + cC1[j][c] + wd * cC0[j][c];
                        const float sA = dpp_from_left(vA * exp_to_right) + vA + dpp_from_right(vA * exp_to_left);
                        const float sB = dpp_from_left(vB * exp_to_right) + vB + dpp_from_right(vB * exp_to_left);
                        const float sC = dpp_from_left(vC * exp_to_right) + vC + dpp_from_right(vC * exp_to_left);
                        const float g = sA + y2[c] * sB + xw2[j][c] * sC + l1g1[j][c];  // d L / d warped_c(r)
                        gix += g * sr[(j * 9 + c) * WAVE];
                        giy += g * sr[(j * 9 + 3 + c) * WAVE];
                    }
                    const float rzs = sr[(j * 9 + 6) * WAVE], ix = sr[(j * 9 + 7) * WAVE], iy = sr[(j * 9 + 8) * WAVE];
                    const float rz = fabsf(rzs);
                    const float dX = gix * rz, dY = giy * rz;
                    const float dz = rzs > 0.f ? -(gix * ix + giy * iy) * rz : 0.f;
                    const float a0 = base[j][0] + col1[j][0] * fr, a1 = base[j][1] + col1[j][1] * fr, a2 = base[j][2] + col1[j][2] * fr;
                    ddsum += dX * a0 + dY * a1 + dz * a2;
                    if (lane_own) {
                        const float sX = dd_ * dX, sY = dd_ * dY, sZ = dd_ * dz;
                        pacc[j][0] += sX; pacc[j][1] += sY; pacc[j][2] += sZ;
                        pacc[j][3] += sX * fr; pacc[j][4] += sY * fr; pacc[j][5] += sZ * fr;
                        pacc[j][6] += dX; pacc[j][7] += dY; pacc[j][8] += dz;
                    }
                }
                if (lane_own) gout[r * W + cu] = ddsum * ddf_;
            }

            // ------------------------------ shift the row pipeline ------------------------------
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    xw2[j][c] = xw1[j][c]; xw1[j][c] = xw0[j][c];
                    if (GRAD) {
                        cA2[j][c] = cA1[j][c]; cA1[j][c] = cA0[j][c];
                        cB2[j][c] = cB1[j][c]; cB1[j][c] = cB0[j][c];
                        cC2[j][c] = cC1[j][c]; cC1[j][c] = cC0[j][c];
                        l1g1[j][c] = l1g0[j][c];
                    }
                }
#pragma unroll
            for (int c = 0; c < 3; ++c) { y2[c] = y1[c]; y1[c] = y0[c]; }
            inv1 = inv0;
        }
        row_barrier();
    }

    // ---- wave reduction -> partials[bid][i][:]  (every slot written: the finalize kernels read all of them) ----
    if (active) {
        float* pp = p.partials + ((size_t)bid * p.n + i) * NACC;
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            float v;
            if (k == A_PSUM) v = a_psum;
            else if (k == A_SX) v = a_sx;
            else if (k == A_SY) v = a_sy;
            else if (k == A_SINV) v = a_sinv;
            else if (k < A_POSE) v = 0.f;
            else {  // expand to the 12-slot layout of the finalize kernel: dM[row][{u, v, 1}] (9), dKt (3)
                const int j = (k - A_POSE) / 12, e = (k - A_POSE) % 12;
                if (e < 9) {
                    const int row = e / 3, col = e % 3;
                    v = col == 0 ? pacc[j][row] * fu : (col == 1 ? pacc[j][3 + row] : pacc[j][row]);
                } else {
                    v = pacc[j][6 + (e - 9)];
                }
            }
            if (k >= A_NMASK && k < A_POSE && i == 0) continue;  // written by the image wave
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) pp[k] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// finalize 1: per (image, scale) sum of the block partials in fp64.  grid = (B, n), block = 256
// ---------------------------------------------------------------------------------------------------
__global__ void reproj_fin1(const float* partials, int n, int blocks_per_image, double* persum) {
    __shared__ double sh[8][NACC];
    const int b = blockIdx.x, i = blockIdx.y;
    const int k = threadIdx.x % NACC, g = threadIdx.x / NACC;  // 8 groups
    double s = 0.0;
    for (int blk = g; blk < blocks_per_image; blk += 8)
        s += (double)partials[(((size_t)b * blocks_per_image + blk) * n + i) * NACC + k];
    sh[g][k] = s;
    __syncthreads();
    if (g == 0) {
        double tot = 0.0;
        for (int q = 0; q < 8; ++q) tot += sh[q][k];
        persum[((size_t)b * n + i) * NACC + k] = tot;
    }
}

// finalize 2: losses, pose gradient (euler chain rule), statistics for the backward kernel.  1 block.
struct FinHdr { float pscale; float pad[3]; };

__global__ void reproj_fin2(const double* persum, const CamConst* cam, const float* pose, int B, int H, int W, int n,
                            float photo_w, float smooth_w, int want_grad, float* losses, float* d_pose, FinHdr* hdr,
                            Stats* stats) {
    __shared__ double cnt[3];
    const int tid = threadIdx.x;
    if (tid == 0) {
        double nm = 0, nx = 0, ny = 0;
        for (int b = 0; b < B; ++b) {
            const double* ps = persum + ((size_t)b * n + 0) * NACC;
            nm += ps[A_NMASK]; nx += ps[A_NMX]; ny += ps[A_NMY];
        }
        cnt[0] = nm; cnt[1] = nx; cnt[2] = ny;
        const double hw = (double)H * (double)W;
        double Lp = 0.0, Ls = 0.0;
        for (int i = 0; i < n; ++i) {
            double ps_ = 0.0, sx = 0.0, sy = 0.0;
            const double cxs = (double)smooth_w / ((double)n * nx * (double)(1 << i));
            const double cys = (double)smooth_w / ((double)n * ny * (double)(1 << i));
            for (int b = 0; b < B; ++b) {
                const double* ps = persum + ((size_t)b * n + i) * NACC;
                ps_ += ps[A_PSUM];
                const float mean = (float)(ps[A_SINV] / hw);
                const float mc = fmaxf(mean, 1e-6f);  // depth.py:48-50
                sx += ps[A_SX] / (double)mc;
                sy += ps[A_SY] / (double)mc;
                Stats s;
                s.inv_mc = (float)(1.0 / (double)mc);
                s.cxs = (float)cxs;
                s.cys = (float)cys;
                s.dmean = (mean >= 1e-6f) ? (float)(-(cxs * ps[A_SX] + cys * ps[A_SY]) / ((double)mc * mc * hw)) : 0.f;
                stats[b * n + i] = s;
            }
            Lp += ps_ / nm;
            Ls += (sx / nx + sy / ny) / (double)(1 << i);
        }
        losses[0] = (float)(Lp / n * photo_w);
        losses[1] = (float)(Ls / n * smooth_w);
        hdr->pscale = (float)((double)photo_w / ((double)n * nm));
    }
    __syncthreads();
    if (want_grad && tid < B * 2) {
        const int b = tid / 2, j = tid % 2;
        const double scale = (double)photo_w / ((double)n * cnt[0]);
        double dM[9] = {0}, dKt[3] = {0};
        for (int i = 0; i < n; ++i) {
            const double* ps = persum + ((size_t)b * n + i) * NACC + A_POSE + j * 12;
            for (int k = 0; k < 9; ++k) dM[k] += ps[k];
            for (int k = 0; k < 3; ++k) dKt[k] += ps[9 + k];
        }
        const CamConst& cc = cam[b];
        double K[9], Ki[9];
        for (int k = 0; k < 9; ++k) { K[k] = cc.K[k]; Ki[k] = cc.Kinv[k]; }
        // M = K R Kinv  =>  dR = K^T dM Kinv^T ;  Kt = K t  =>  dt = K^T dKt
        double T[9], dR[9], dt[3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) T[r * 3 + c] = K[r] * dM[c] + K[3 + r] * dM[3 + c] + K[6 + r] * dM[6 + c];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) dR[r * 3 + c] = T[r * 3] * Ki[c * 3] + T[r * 3 + 1] * Ki[c * 3 + 1] + T[r * 3 + 2] * Ki[c * 3 + 2];
        for (int r = 0; r < 3; ++r) dt[r] = K[r] * dKt[0] + K[3 + r] * dKt[1] + K[6 + r] * dKt[2];
        const float* v = pose + ((size_t)b * 2 + j) * 6;
        double X[9], Y[9], Z[9], XY[9], YZ[9], tmp[9], dX[9], dY[9], dZ[9];
        euler_mats(v + 3, X, Y, Z);
        mm3(X, Y, XY);
        mm3(Y, Z, YZ);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                dX[r * 3 + c] = dR[r * 3] * YZ[c * 3] + dR[r * 3 + 1] * YZ[c * 3 + 1] + dR[r * 3 + 2] * YZ[c * 3 + 2];
                tmp[r * 3 + c] = X[r] * dR[c] + X[3 + r] * dR[3 + c] + X[6 + r] * dR[6 + c];
                dZ[r * 3 + c] = XY[r] * dR[c] + XY[3 + r] * dR[3 + c] + XY[6 + r] * dR[6 + c];
            }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) dY[r * 3 + c] = tmp[r * 3] * Z[c * 3] + tmp[r * 3 + 1] * Z[c * 3 + 1] + tmp[r * 3 + 2] * Z[c * 3 + 2];
        const double cx = X[4], sx = X[7], cy = Y[0], sy = Y[2], cz = Z[0], sz = Z[3];
        float* o = d_pose + ((size_t)b * 2 + j) * 6;
        o[0] = (float)(scale * dt[0]);
        o[1] = (float)(scale * dt[1]);
        o[2] = (float)(scale * dt[2]);
        o[3] = (float)(scale * (dX[4] * (-sx) + dX[5] * (-cx) + dX[7] * cx + dX[8] * (-sx)));
        o[4] = (float)(scale * (dY[0] * (-sy) + dY[2] * cy + dY[6] * (-cy) + dY[8] * (-sy)));
        o[5] = (float)(scale * (dZ[0] * (-sz) + dZ[1] * (-cz) + dZ[3] * cz + dZ[4] * (-sz)));
    }
}

// ---------------------------------------------------------------------------------------------------
// backward: g_inv[i] <- gp * pscale * g_inv[i] + gs * d(smoothness)/d inv[i]      (streaming, 1 thread / pixel)
// ---------------------------------------------------------------------------------------------------
struct BwdParams {
    const float* inv[MGN_MAX_SCALES];
    float* ginv[MGN_MAX_SCALES];
    const void* img;      // fp32 planes [B,3,H,W], or (U8 kernels) uint8 RGBX [B,H,W,4]
    const uint8_t* mask;
    const float* grad_losses;
    const FinHdr* hdr;
    const Stats* stats;
    const float* d_pose;
    float* d_pose_out;
    int B, H, W, n;
};
MGN_PLAN_RO(BwdParams, MGN_RO(inv) MGN_RO(img) MGN_RO(mask) MGN_RO(grad_losses) MGN_RO(hdr) MGN_RO(stats) MGN_RO(d_pose))   // pointers the kernels only read through (launch-plan dependency analysis, csrc/mgn_launch.h)

template <bool U8>
__global__ __launch_bounds__(256) void reproj_bwd(BwdParams p) {
    const int W = p.W, H = p.H, HWp = H * W;
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = blockIdx.y, b = blockIdx.z;
    const float gp = p.grad_losses[0] * p.hdr->pscale, gs = p.grad_losses[1];
    if (blockIdx.x == 0 && v == 0 && b == 0)
        for (int k = threadIdx.x; k < p.B * 12; k += blockDim.x) p.d_pose_out[k] = p.grad_losses[0] * p.d_pose[k];
    if (u >= W) return;
    const int off = v * W + u;
    const float* im = (const float*)p.img + (size_t)b * 3 * HWp;
    const uint32_t* im32 = (const uint32_t*)p.img + (size_t)b * HWp;
    const uint8_t* mk = p.mask ? p.mask + (size_t)b * HWp : nullptr;
    const bool hasR = u + 1 < W, hasL = u > 0, hasD = v + 1 < H, hasU = v > 0;
    float gR = 0.f, gL = 0.f, gD = 0.f, gU = 0.f;  // image gradient magnitudes towards right/left/down/up
    if constexpr (U8) {
        float ctr[3], o[3];
        u8unit3(im32[off], ctr);
        if (hasR) { u8unit3(im32[off + 1], o); gR = fabsf(ctr[0] - o[0]) + fabsf(ctr[1] - o[1]) + fabsf(ctr[2] - o[2]); }
        if (hasL) { u8unit3(im32[off - 1], o); gL = fabsf(o[0] - ctr[0]) + fabsf(o[1] - ctr[1]) + fabsf(o[2] - ctr[2]); }
        if (hasD) { u8unit3(im32[off + W], o); gD = fabsf(ctr[0] - o[0]) + fabsf(ctr[1] - o[1]) + fabsf(ctr[2] - o[2]); }
        if (hasU) { u8unit3(im32[off - W], o); gU = fabsf(o[0] - ctr[0]) + fabsf(o[1] - ctr[1]) + fabsf(o[2] - ctr[2]); }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float ctr = im[c * HWp + off];
            if (hasR) gR += fabsf(ctr - im[c * HWp + off + 1]);
            if (hasL) gL += fabsf(im[c * HWp + off - 1] - ctr);
            if (hasD) gD += fabsf(ctr - im[c * HWp + off + W]);
            if (hasU) gU += fabsf(im[c * HWp + off - W] - ctr);
        }
    }
    const bool mC = mk ? mk[off] != 0 : true;
    const bool mL = hasL && (mk ? mk[off - 1] != 0 : true);
    const bool mU = hasU && (mk ? mk[off - W] != 0 : true);
    // weight of the pair (p, p+1) belongs to p and is masked by mask[p] (loss.py:284-285)
    const float wR = (hasR && mC) ? __expf(-gR * (1.f / 3.f)) : 0.f;
    const float wL = mL ? __expf(-gL * (1.f / 3.f)) : 0.f;
    const float wD = (hasD && mC) ? __expf(-gD * (1.f / 3.f)) : 0.f;
    const float wU = mU ? __expf(-gU * (1.f / 3.f)) : 0.f;
    for (int i = 0; i < p.n; ++i) {
        const float* iv = p.inv[i] + (size_t)b * HWp;
        const Stats s = p.stats[b * p.n + i];
        const float c0 = iv[off];
        auto sgn = [](float x) { return (float)((x > 0.f) - (x < 0.f)); };
        float ddx = 0.f, ddy = 0.f;
        if (hasR) ddx += wR * sgn(c0 - iv[off + 1]);
        if (hasL) ddx -= wL * sgn(iv[off - 1] - c0);
        if (hasD) ddy += wD * sgn(c0 - iv[off + W]);
        if (hasU) ddy -= wU * sgn(iv[off - W] - c0);
        const float ddn = s.cxs * ddx + s.cys * ddy;
        float* g = p.ginv[i] + (size_t)b * HWp;
        g[off] = gp * g[off] + gs * (ddn * s.inv_mc + s.dmean);
    }
}

// Same computation, 4 pixels per thread with 16-byte loads/stores (W % 4 == 0).
template <bool U8>
__global__ __launch_bounds__(256) void reproj_bwd4(BwdParams p) {
    const int W = p.W, H = p.H, HWp = H * W;
    const int u = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int v = blockIdx.y, b = blockIdx.z;
    const float gp = p.grad_losses[0] * p.hdr->pscale, gs = p.grad_losses[1];
    if (blockIdx.x == 0 && v == 0 && b == 0)
        for (int k = threadIdx.x; k < p.B * 12; k += blockDim.x) p.d_pose_out[k] = p.grad_losses[0] * p.d_pose[k];
    if (u >= W) return;
    const int off = v * W + u;
    const float* im = (const float*)p.img + (size_t)b * 3 * HWp;
    const uint32_t* im32 = (const uint32_t*)p.img + (size_t)b * HWp;
    const uint8_t* mk = p.mask ? p.mask + (size_t)b * HWp : nullptr;
    const bool hasL = u > 0, hasR4 = u + 4 < W, hasD = v + 1 < H, hasU = v > 0;
    // image gradient magnitudes of the pairs (k-1,k) for k=0..4 along x, and (up,ctr), (ctr,down) per pixel
    float gx[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, gu[4] = {0.f, 0.f, 0.f, 0.f}, gd[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (U8) {   // 4 packed pixels per 16-byte load: 5 loads per thread instead of 15
        const uint4 c4 = *reinterpret_cast<const uint4*>(im32 + off);
        const uint32_t cw[4] = {c4.x, c4.y, c4.z, c4.w};
        float cv[4][3], l[3] = {0.f, 0.f, 0.f}, r[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) u8unit3(cw[k], cv[k]);
        if (hasL) u8unit3(im32[off - 1], l);
        if (hasR4) u8unit3(im32[off + 4], r);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            gx[0] += fabsf(l[c] - cv[0][c]);
            gx[1] += fabsf(cv[0][c] - cv[1][c]);
            gx[2] += fabsf(cv[1][c] - cv[2][c]);
            gx[3] += fabsf(cv[2][c] - cv[3][c]);
            gx[4] += fabsf(cv[3][c] - r[c]);
        }
        if (hasU) {
            const uint4 u4 = *reinterpret_cast<const uint4*>(im32 + off - W);
            const uint32_t uw[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float o[3];
                u8unit3(uw[k], o);
                gu[k] = fabsf(o[0] - cv[k][0]) + fabsf(o[1] - cv[k][1]) + fabsf(o[2] - cv[k][2]);
            }
        }
        if (hasD) {
            const uint4 d4 = *reinterpret_cast<const uint4*>(im32 + off + W);
            const uint32_t dw[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float o[3];
                u8unit3(dw[k], o);
                gd[k] = fabsf(cv[k][0] - o[0]) + fabsf(cv[k][1] - o[1]) + fabsf(cv[k][2] - o[2]);
            }
        }
    } else
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* ic = im + c * HWp + off;
        const float4 ctr = *reinterpret_cast<const float4*>(ic);
        const float cv[4] = {ctr.x, ctr.y, ctr.z, ctr.w};
        const float l = hasL ? ic[-1] : 0.f, r = hasR4 ? ic[4] : 0.f;
        gx[0] += fabsf(l - cv[0]);
        gx[1] += fabsf(cv[0] - cv[1]);
        gx[2] += fabsf(cv[1] - cv[2]);
        gx[3] += fabsf(cv[2] - cv[3]);
        gx[4] += fabsf(cv[3] - r);
        if (hasU) {
            const float4 up = *reinterpret_cast<const float4*>(ic - W);
            gu[0] += fabsf(up.x - cv[0]); gu[1] += fabsf(up.y - cv[1]); gu[2] += fabsf(up.z - cv[2]); gu[3] += fabsf(up.w - cv[3]);
        }
        if (hasD) {
            const float4 dn = *reinterpret_cast<const float4*>(ic + W);
            gd[0] += fabsf(cv[0] - dn.x); gd[1] += fabsf(cv[1] - dn.y); gd[2] += fabsf(cv[2] - dn.z); gd[3] += fabsf(cv[3] - dn.w);
        }
    }
    bool mC[4] = {true, true, true, true}, mU[4] = {hasU, hasU, hasU, hasU};
    bool mL = hasL;
    if (mk) {
        const uchar4 m4 = *reinterpret_cast<const uchar4*>(mk + off);
        mC[0] = m4.x != 0; mC[1] = m4.y != 0; mC[2] = m4.z != 0; mC[3] = m4.w != 0;
        if (hasU) {
            const uchar4 u4 = *reinterpret_cast<const uchar4*>(mk + off - W);
            mU[0] = u4.x != 0; mU[1] = u4.y != 0; mU[2] = u4.z != 0; mU[3] = u4.w != 0;
        }
        if (hasL) mL = mk[off - 1] != 0;
    }
    // pair weights: the pair (p, p+1) belongs to p and is masked by mask[p] (loss.py:284-285)
    float wx[5], wd[4], wu[4];
    wx[0] = mL ? __expf(-gx[0] * (1.f / 3.f)) : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool pair_exists = (k < 3) || hasR4;
        wx[k + 1] = (mC[k] && pair_exists) ? __expf(-gx[k + 1] * (1.f / 3.f)) : 0.f;
        wd[k] = (mC[k] && hasD) ? __expf(-gd[k] * (1.f / 3.f)) : 0.f;
        wu[k] = mU[k] ? __expf(-gu[k] * (1.f / 3.f)) : 0.f;
    }
    auto sgn = [](float x) { return (float)((x > 0.f) - (x < 0.f)); };
    for (int i = 0; i < p.n; ++i) {
        const float* iv = p.inv[i] + (size_t)b * HWp + off;
        const Stats s = p.stats[b * p.n + i];
        const float4 c4 = *reinterpret_cast<const float4*>(iv);
        const float cv[4] = {c4.x, c4.y, c4.z, c4.w};
        const float l = hasL ? iv[-1] : 0.f, r = hasR4 ? iv[4] : 0.f;
        float upv[4] = {0.f, 0.f, 0.f, 0.f}, dnv[4] = {0.f, 0.f, 0.f, 0.f};
        if (hasU) { const float4 t4 = *reinterpret_cast<const float4*>(iv - W); upv[0] = t4.x; upv[1] = t4.y; upv[2] = t4.z; upv[3] = t4.w; }
        if (hasD) { const float4 t4 = *reinterpret_cast<const float4*>(iv + W); dnv[0] = t4.x; dnv[1] = t4.y; dnv[2] = t4.z; dnv[3] = t4.w; }
        const float e[6] = {l, cv[0], cv[1], cv[2], cv[3], r};
        float* g = p.ginv[i] + (size_t)b * HWp + off;
        const float4 g4 = *reinterpret_cast<const float4*>(g);
        const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // pairs: (k-1,k) has weight wx[k], (k,k+1) has weight wx[k+1]
            const float ddx = wx[k + 1] * sgn(e[k + 1] - e[k + 2]) - wx[k] * sgn(e[k] - e[k + 1]);
            const float ddy = wd[k] * sgn(cv[k] - dnv[k]) - wu[k] * sgn(upv[k] - cv[k]);
            const float ddn = s.cxs * ddx + s.cys * ddy;
            o[k] = gp * gv[k] + gs * (ddn * s.inv_mc + s.dmean);
        }
        *reinterpret_cast<float4*>(g) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
struct Layout {
    int RH, nseg, nstrips, nblocks;
    size_t lds_bytes;
    size_t off_cam, off_partials, off_persum, off_hdr, off_stats, total;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int make_layout(const mgn_reproj_cfg* c, Layout* L) {
    if (!c || c->B < 1 || c->H < 2 || c->W < 2 || c->n_scales < 1 || c->n_scales > MGN_MAX_SCALES) return MGN_EINVAL;
    if ((long long)c->H * c->W > (1LL << 28)) return MGN_EINVAL;   // 32-bit byte offsets into one image plane
    L->nstrips = (c->W + STRIP - 1) / STRIP;
    int RH = c->rows_per_wave;
    if (RH <= 0) {
        RH = 64;
        const long long target = 1536;  // blocks: 256 CUs x 3 resident blocks x 2 rounds
        while (RH > 8 && (long long)c->B * L->nstrips * ((c->H + RH - 1) / RH) < target) RH >>= 1;
    }
    if (RH < 4) return MGN_EINVAL;
    L->RH = RH;
    L->nseg = (c->H + RH - 1) / RH;
    L->nblocks = c->B * L->nseg * L->nstrips;
    L->lds_bytes = sizeof(float) * WAVE * ((size_t)c->n_scales * RING * NSTATE + 2 * NSHARE);
    size_t o = 0;
    L->off_cam = o;      o = align_up(o + sizeof(CamConst) * c->B, 256);
    L->off_partials = o; o = align_up(o + sizeof(float) * NACC * c->n_scales * (size_t)L->nblocks, 256);
    L->off_persum = o;   o = align_up(o + sizeof(double) * NACC * c->n_scales * c->B, 256);
    L->off_hdr = o;      o = align_up(o + sizeof(FinHdr), 256);
    L->off_stats = o;    o = align_up(o + sizeof(Stats) * c->n_scales * c->B, 256);
    L->total = o;
    return MGN_OK;
}

int check_options(const mgn_reproj_cfg* c) {
    if (c->padding_mode < 0 || c->padding_mode > 2) return MGN_EINVAL;
    if ((c->automask_loss != 0 && c->automask_loss != 1) || (c->photometric_reduce_op != 0 && c->photometric_reduce_op != 1)) return MGN_EINVAL;
    if (c->automask_loss && c->photometric_reduce_op != 0) return MGN_EINVAL;   // loss.py:105-109: automasking goes with "min"
    if (c->frame_layout < 0 || c->frame_layout > MGN_FRAMES_RGBX_U8) return MGN_EINVAL;
    if (!(c->ssim_loss_weight >= 0.f)) return MGN_EINVAL;   // (0: 3-channel L1 maps, see mgn_reproj_loss_fwd)
    return MGN_OK;
}

}  // namespace

extern "C" {

const char* mgn_version(void) { return "mgnet_hip 0.1 gfx950"; }

int mgn_reproj_workspace_bytes(const mgn_reproj_cfg* cfg, size_t* bytes) {
    Layout L;
    int rc = make_layout(cfg, &L);
    if (rc != MGN_OK) return rc;
    if (!bytes) return MGN_EINVAL;
    *bytes = L.total;
    return MGN_OK;
}

int mgn_reproj_loss_fwd(const mgn_reproj_cfg* cfg, const float* const* inv_depth, const void* img, const void* prev,
                        const void* next, const uint8_t* mask, const float* cam, int cam_stride, int cam_ld,
                        const float* pose, int want_grad, float* losses, float* d_pose, float* const* g_inv,
                        float* dbg_minmap, void* workspace, size_t workspace_bytes, void* stream_) {
    Layout L;
    int rc = make_layout(cfg, &L);
    if (rc != MGN_OK) return rc;
    rc = check_options(cfg);
    if (rc != MGN_OK) return rc;
    if (!inv_depth || !img || !prev || !next || !cam || !pose || !losses || !workspace) return MGN_EINVAL;
    if (cam_ld < 3 || cam_stride < 2 * cam_ld + 3) return MGN_EINVAL;
    // ssim_loss_weight = 0 (loss.py:196-197: 3-channel L1 maps): "min" over channels and sources exists only WITH a reprojection mask,
    // "mean" only WITHOUT one -- the reference's boolean indexing raises IndexError for the other two combinations (loss.py:236-246)
    if (cfg->ssim_loss_weight == 0.f && ((cfg->photometric_reduce_op == 0) == (mask == nullptr))) return MGN_EINVAL;
    if (want_grad && (!d_pose || !g_inv)) return MGN_EINVAL;
    if (workspace_bytes < L.total) return MGN_ENOSPC;
    if (cfg->B * 2 > 256) return MGN_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;

    Params p;
    for (int i = 0; i < MGN_MAX_SCALES; ++i) {
        p.inv[i] = i < cfg->n_scales ? inv_depth[i] : nullptr;
        p.ginv[i] = (want_grad && i < cfg->n_scales) ? g_inv[i] : nullptr;
        if (i < cfg->n_scales && (!p.inv[i] || (want_grad && !p.ginv[i]))) return MGN_EINVAL;
    }
    p.img = img; p.prev = prev; p.nxt = next; p.mask = mask;
    p.cam = (const CamConst*)(ws + L.off_cam);
    p.partials = (float*)(ws + L.off_partials);
    p.dbg = dbg_minmap;
    p.B = cfg->B; p.H = cfg->H; p.W = cfg->W; p.n = cfg->n_scales;
    p.RH = L.RH; p.nseg = L.nseg; p.nstrips = L.nstrips;
    p.ssim_w = cfg->ssim_loss_weight;
    p.automask = cfg->automask_loss;
    p.reduce_mean = cfg->photometric_reduce_op;
    p.pad_mode = cfg->padding_mode;

    hipLaunchKernelGGL(reproj_prep, dim3((cfg->B + 63) / 64), dim3(64), 0, stream, cam, cam_stride, cam_ld, pose, cfg->B,
                       (CamConst*)(ws + L.off_cam));
    mgn_plan::prof_mark(0, cfg->prof_begin, (hipStream_t)stream);
    const dim3 mgrid(L.nblocks), mblock(WAVE * (cfg->n_scales + 1));
#define MGN_MARCH(G, F, P, M) hipLaunchKernelGGL((reproj_march<G, F, P, M>), mgrid, mblock, L.lds_bytes, stream, p)
#define MGN_MARCH_P(F, P) do { if (l1min) { if (want_grad) MGN_MARCH(true, F, P, true); else MGN_MARCH(false, F, P, true); } \
                               else { if (want_grad) MGN_MARCH(true, F, P, false); else MGN_MARCH(false, F, P, false); } } while (0)
#define MGN_MARCH_F(F) do { if (cfg->padding_mode) MGN_MARCH_P(F, true); else MGN_MARCH_P(F, false); } while (0)
    const bool l1min = cfg->ssim_loss_weight == 0.f && cfg->photometric_reduce_op == 0;   // 3-channel L1 maps, minimum over channels and sources
    if (cfg->frame_layout == MGN_FRAMES_RGBX_U8) MGN_MARCH_F(FMT_RGBX_U8);               // img / prev / next are uint8 [B][H][W][4]
    else if (cfg->frame_layout == MGN_FRAMES_CTX_RGBX_F32) MGN_MARCH_F(FMT_RGBX_F32);    // prev / next are fp32 [B][H][W][4] (4th channel unused)
    else MGN_MARCH_F(FMT_PLANAR);
#undef MGN_MARCH_P
#undef MGN_MARCH_F
#undef MGN_MARCH
    mgn_plan::prof_mark(1, cfg->prof_end, (hipStream_t)stream);
    hipLaunchKernelGGL(reproj_fin1, dim3(cfg->B, cfg->n_scales), dim3(256), 0, stream, (const float*)(ws + L.off_partials),
                       cfg->n_scales, L.nseg * L.nstrips, (double*)(ws + L.off_persum));
    hipLaunchKernelGGL(reproj_fin2, dim3(1), dim3(256), 0, stream, (const double*)(ws + L.off_persum),
                       (const CamConst*)(ws + L.off_cam), pose, cfg->B, cfg->H, cfg->W, cfg->n_scales,
                       cfg->photometric_loss_weight, cfg->smoothing_loss_weight, want_grad, losses, d_pose,
                       (FinHdr*)(ws + L.off_hdr), (Stats*)(ws + L.off_stats));
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_reproj_loss_bwd(const mgn_reproj_cfg* cfg, const float* const* inv_depth, const void* img, const uint8_t* mask,
                        const float* grad_losses, const float* d_pose, float* const* g_inv, float* d_pose_out,
                        const void* workspace, size_t workspace_bytes, void* stream_) {
    Layout L;
    int rc = make_layout(cfg, &L);
    if (rc != MGN_OK) return rc;
    rc = check_options(cfg);
    if (rc != MGN_OK) return rc;
    if (!inv_depth || !img || !grad_losses || !d_pose || !g_inv || !d_pose_out || !workspace) return MGN_EINVAL;
    if (workspace_bytes < L.total) return MGN_ENOSPC;
    const char* ws = (const char*)workspace;
    BwdParams p;
    for (int i = 0; i < MGN_MAX_SCALES; ++i) {
        p.inv[i] = i < cfg->n_scales ? inv_depth[i] : nullptr;
        p.ginv[i] = i < cfg->n_scales ? g_inv[i] : nullptr;
        if (i < cfg->n_scales && (!p.inv[i] || !p.ginv[i])) return MGN_EINVAL;
    }
    p.img = img; p.mask = mask; p.grad_losses = grad_losses;
    p.hdr = (const FinHdr*)(ws + L.off_hdr);
    p.stats = (const Stats*)(ws + L.off_stats);
    p.d_pose = d_pose; p.d_pose_out = d_pose_out;
    p.B = cfg->B; p.H = cfg->H; p.W = cfg->W; p.n = cfg->n_scales;
    const bool vec4 = (cfg->W % 4 == 0) && (((uintptr_t)img | (uintptr_t)mask) % 16 == 0);
    bool aligned = vec4;
    for (int i = 0; i < cfg->n_scales; ++i) aligned = aligned && (((uintptr_t)inv_depth[i] | (uintptr_t)g_inv[i]) % 16 == 0);
    const bool u8 = cfg->frame_layout == MGN_FRAMES_RGBX_U8;
    const dim3 g4((cfg->W / 4 + 255) / 256, cfg->H, cfg->B), g1((cfg->W + 255) / 256, cfg->H, cfg->B);
    if (aligned && u8) hipLaunchKernelGGL(reproj_bwd4<true>, g4, dim3(256), 0, (hipStream_t)stream_, p);
    else if (aligned) hipLaunchKernelGGL(reproj_bwd4<false>, g4, dim3(256), 0, (hipStream_t)stream_, p);
    else if (u8) hipLaunchKernelGGL(reproj_bwd<true>, g1, dim3(256), 0, (hipStream_t)stream_, p);
    else hipLaunchKernelGGL(reproj_bwd<false>, g1, dim3(256), 0, (hipStream_t)stream_, p);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

}  // extern "C"

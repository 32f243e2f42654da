// geometry.hip -- the stand-alone stages of mgnet/geometry for MI355X (gfx950): the public functions of the reference's
// geometry package that user code can call outside the fused reprojection loss (reproj_loss.hip fuses the same stages).
//
// Replaces (reference file:line):
//   mgn_view_synthesis_*  camera_utils.py:24-55 view_synthesis = Camera.reconstruct (camera.py:107-141) -> Camera.project
//                         (camera.py:143-182) -> F.grid_sample(bilinear, zeros, align_corners=True), one pass, nothing
//                         but the warped image written
//   mgn_reconstruct_*     camera.py:107-141 Camera.reconstruct (+ pose.py:77-83 transform_points for frame "w")
//   mgn_project_*         camera.py:143-182 Camera.project (+ transform_points)
// The 3x3 products of intrinsics and poses (K.R.Kinv etc., B x 9 numbers) are formed by the caller; every kernel takes an
// affine map per image:  A [B,9] row-major and t [B,3].
//
// All kernels are HBM streaming kernels: one thread per pixel, a lane is a pixel column (coalesced planar NCHW rows), the
// per-image gradient sums of A and t leave each block as 12 partial sums (deterministic: no atomics; the caller adds the
// [blocks,12] partials).  Bytes per pixel (fp32): view synthesis forward 4 + 4C gathered + 4C written; backward adds 4C read,
// 4 written.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgnet_hip.h"

namespace {

constexpr int TPB = 256;

struct Affine {
    float a[9], t[3];
};

__device__ __forceinline__ Affine load_affine(const float* A, const float* t, int b) {
    Affine r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.a[k] = A[b * 9 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) r.t[k] = t[b * 3 + k];
    return r;
}

// block sum of 12 per-thread values -> out[12] (fixed order: deterministic)
__device__ __forceinline__ void block_sum12(float (&v)[12], float* out) {
    __shared__ float sh[TPB / 64][12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        float x = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < TPB / 64; ++w) s += sh[w][threadIdx.x];
        out[threadIdx.x] = s;
    }
}

struct Corners {
    int o00, o10, o01, o11;
    bool k00, k10, k01, k11;
    float tx, ty;
};

// F.grid_sample(align_corners=True, padding_mode="zeros") corner set at pixel position (ix, iy)
__device__ __forceinline__ Corners corners(float ix, float iy, int H, int W) {
    Corners c;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    c.tx = ix - fx0;
    c.ty = iy - fy0;
    const int xi = (int)fminf(fmaxf(fx0, -2.f), (float)W), yi = (int)fminf(fmaxf(fy0, -2.f), (float)H);
    const bool x0ok = (unsigned)xi < (unsigned)W, x1ok = (unsigned)(xi + 1) < (unsigned)W;
    const bool y0ok = (unsigned)yi < (unsigned)H, y1ok = (unsigned)(yi + 1) < (unsigned)H;
    const int x0 = min(max(xi, 0), W - 1), x1 = min(max(xi + 1, 0), W - 1);
    const int y0 = min(max(yi, 0), H - 1), y1 = min(max(yi + 1, 0), H - 1);
    c.o00 = y0 * W + x0; c.o10 = y0 * W + x1; c.o01 = y1 * W + x0; c.o11 = y1 * W + x1;
    c.k00 = x0ok && y0ok; c.k10 = x1ok && y0ok; c.k01 = x0ok && y1ok; c.k11 = x1ok && y1ok;
    return c;
}

// grid (ceil(W/TPB), H, B)
template <bool BWD>
__global__ __launch_bounds__(TPB) void view_synthesis_kernel(const float* __restrict__ ref, const float* __restrict__ depth,
                                                             const float* __restrict__ A, const float* __restrict__ t,
                                                             const float* __restrict__ g_out, int C, int H, int W,
                                                             float* __restrict__ out, float* __restrict__ d_depth,
                                                             float* __restrict__ partials) {
    const int u = blockIdx.x * TPB + threadIdx.x, v = blockIdx.y, b = blockIdx.z;
    const int HW = H * W;
    const Affine m = load_affine(A, t, b);
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    if (u < W) {
        const int off = v * W + u;
        const float fu = (float)u, fv = (float)v;
        const float d = depth[(size_t)b * HW + off];
        const float a0 = m.a[0] * fu + m.a[1] * fv + m.a[2], a1 = m.a[3] * fu + m.a[4] * fv + m.a[5], a2 = m.a[6] * fu + m.a[7] * fv + m.a[8];
        const float X = d * a0 + m.t[0], Y = d * a1 + m.t[1], z = d * a2 + m.t[2];
        const bool zf = z >= 1e-5f;                       // camera.py:172 clamp(min=1e-5)
        const float rz = 1.0f / fmaxf(z, 1e-5f);
        const float ix = X * rz, iy = Y * rz;             // == ((Xnorm + 1) / 2) * (W - 1) of grid_sample
        const Corners c = corners(ix, iy, H, W);
        const float sx = 1.f - c.tx, sy = 1.f - c.ty;
        float gix = 0.f, giy = 0.f;
        for (int ch = 0; ch < C; ++ch) {
            const float* rc = ref + ((size_t)b * C + ch) * HW;
            const float v00 = c.k00 ? rc[c.o00] : 0.f, v10 = c.k10 ? rc[c.o10] : 0.f;
            const float v01 = c.k01 ? rc[c.o01] : 0.f, v11 = c.k11 ? rc[c.o11] : 0.f;
            if (!BWD) {
                out[((size_t)b * C + ch) * HW + off] = v00 * (sx * sy) + v10 * (c.tx * sy) + v01 * (sx * c.ty) + v11 * (c.tx * c.ty);
            } else {
                const float g = g_out[((size_t)b * C + ch) * HW + off];
                gix += g * ((v10 - v00) * sy + (v11 - v01) * c.ty);
                giy += g * ((v01 - v00) * sx + (v11 - v10) * c.tx);
            }
        }
        if (BWD) {
            const float dX = gix * rz, dY = giy * rz, dz = zf ? -(gix * ix + giy * iy) * rz : 0.f;
            d_depth[(size_t)b * HW + off] = dX * a0 + dY * a1 + dz * a2;
            const float dd[3] = {dX, dY, dz};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                acc[k * 3] = dd[k] * d * fu; acc[k * 3 + 1] = dd[k] * d * fv; acc[k * 3 + 2] = dd[k] * d;
                acc[9 + k] = dd[k];
            }
        }
    }
    if (BWD) block_sum12(acc, partials + (((size_t)b * gridDim.y + v) * gridDim.x + blockIdx.x) * 12);
}

// points[b,k,v,u] = depth * (A_k . [u,v,1]) + t_k
template <bool BWD>
__global__ __launch_bounds__(TPB) void reconstruct_kernel(const float* __restrict__ depth, const float* __restrict__ A,
                                                          const float* __restrict__ t, const float* __restrict__ g, int H, int W,
                                                          float* __restrict__ points, float* __restrict__ d_depth,
                                                          float* __restrict__ partials) {
    const int u = blockIdx.x * TPB + threadIdx.x, v = blockIdx.y, b = blockIdx.z;
    const int HW = H * W;
    const Affine m = load_affine(A, t, b);
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    if (u < W) {
        const int off = v * W + u;
        const float fu = (float)u, fv = (float)v;
        const float d = depth[(size_t)b * HW + off];
        float dsum = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float a = m.a[k * 3] * fu + m.a[k * 3 + 1] * fv + m.a[k * 3 + 2];
            if (!BWD) {
                points[((size_t)b * 3 + k) * HW + off] = a * d + m.t[k];
            } else {
                const float gk = g[((size_t)b * 3 + k) * HW + off];
                dsum += gk * a;
                acc[k * 3] = gk * d * fu; acc[k * 3 + 1] = gk * d * fv; acc[k * 3 + 2] = gk * d;
                acc[9 + k] = gk;
            }
        }
        if (BWD) d_depth[(size_t)b * HW + off] = dsum;
    }
    if (BWD) block_sum12(acc, partials + (((size_t)b * gridDim.y + v) * gridDim.x + blockIdx.x) * 12);
}

// coords[b,v,u,:] = (2 (X/Z) / (W-1) - 1, 2 (Y/Z) / (H-1) - 1),  (X,Y,z) = A.P + t, Z = clamp(z, 1e-5)
template <bool BWD>
__global__ __launch_bounds__(TPB) void project_kernel(const float* __restrict__ pts, const float* __restrict__ A,
                                                      const float* __restrict__ t, const float* __restrict__ g, int H, int W,
                                                      float* __restrict__ coords, float* __restrict__ d_pts,
                                                      float* __restrict__ partials) {
    const int u = blockIdx.x * TPB + threadIdx.x, v = blockIdx.y, b = blockIdx.z;
    const int HW = H * W;
    const Affine m = load_affine(A, t, b);
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    if (u < W) {
        const int off = v * W + u;
        float P[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) P[k] = pts[((size_t)b * 3 + k) * HW + off];
        const float X = m.a[0] * P[0] + m.a[1] * P[1] + m.a[2] * P[2] + m.t[0];
        const float Y = m.a[3] * P[0] + m.a[4] * P[1] + m.a[5] * P[2] + m.t[1];
        const float z = m.a[6] * P[0] + m.a[7] * P[1] + m.a[8] * P[2] + m.t[2];
        const float Z = fmaxf(z, 1e-5f);
        const float kx = 2.f / (float)(W - 1), ky = 2.f / (float)(H - 1);
        float2* co = reinterpret_cast<float2*>(coords) + (size_t)b * HW + off;
        if (!BWD) {
            *co = make_float2(2.f * (X / Z) / (float)(W - 1) - 1.0f, 2.f * (Y / Z) / (float)(H - 1) - 1.0f);  // camera.py:174-175 op order
        } else {
            const float2 gg = reinterpret_cast<const float2*>(g)[(size_t)b * HW + off];
            const float rz = 1.f / Z;
            const float dX = gg.x * kx * rz, dY = gg.y * ky * rz;
            const float dz = (z >= 1e-5f) ? -(dX * X + dY * Y) * rz : 0.f;
            const float dd[3] = {dX, dY, dz};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d_pts[((size_t)b * 3 + k) * HW + off] = m.a[k] * dX + m.a[3 + k] * dY + m.a[6 + k] * dz;
                acc[k * 3] = dd[k] * P[0]; acc[k * 3 + 1] = dd[k] * P[1]; acc[k * 3 + 2] = dd[k] * P[2];
                acc[9 + k] = dd[k];
            }
        }
    }
    if (BWD) block_sum12(acc, partials + (((size_t)b * gridDim.y + v) * gridDim.x + blockIdx.x) * 12);
}

inline bool bad_shape(int B, int H, int W) { return B < 1 || H < 2 || W < 2 || (long long)H * W > (1LL << 30) || B > 65535 || H > 65535; }
inline dim3 grid_of(int B, int H, int W) { return dim3((W + TPB - 1) / TPB, H, B); }

}  // namespace

extern "C" {

int mgn_geometry_partial_rows(int B, int H, int W, size_t* rows) {
    if (bad_shape(B, H, W) || !rows) return MGN_EINVAL;
    *rows = (size_t)B * H * ((W + TPB - 1) / TPB);
    return MGN_OK;
}

int mgn_view_synthesis_fwd(const float* ref, const float* depth, const float* A, const float* t, int B, int C, int H, int W,
                           int padding_mode, float* out, void* stream) {
    if (!ref || !depth || !A || !t || !out || C < 1 || bad_shape(B, H, W)) return MGN_EINVAL;
    if (padding_mode != 0) return MGN_ENOTSUP;
    hipLaunchKernelGGL(view_synthesis_kernel<false>, grid_of(B, H, W), dim3(TPB), 0, (hipStream_t)stream, ref, depth, A, t,
                       (const float*)nullptr, C, H, W, out, (float*)nullptr, (float*)nullptr);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_view_synthesis_bwd(const float* ref, const float* depth, const float* A, const float* t, const float* g_out, int B,
                           int C, int H, int W, int padding_mode, float* d_depth, float* partials, void* stream) {
    if (!ref || !depth || !A || !t || !g_out || !d_depth || !partials || C < 1 || bad_shape(B, H, W)) return MGN_EINVAL;
    if (padding_mode != 0) return MGN_ENOTSUP;
    hipLaunchKernelGGL(view_synthesis_kernel<true>, grid_of(B, H, W), dim3(TPB), 0, (hipStream_t)stream, ref, depth, A, t, g_out,
                       C, H, W, (float*)nullptr, d_depth, partials);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_reconstruct_fwd(const float* depth, const float* A, const float* t, int B, int H, int W, float* points, void* stream) {
    if (!depth || !A || !t || !points || bad_shape(B, H, W)) return MGN_EINVAL;
    hipLaunchKernelGGL(reconstruct_kernel<false>, grid_of(B, H, W), dim3(TPB), 0, (hipStream_t)stream, depth, A, t,
                       (const float*)nullptr, H, W, points, (float*)nullptr, (float*)nullptr);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_reconstruct_bwd(const float* depth, const float* A, const float* t, const float* g_points, int B, int H, int W,
                        float* d_depth, float* partials, void* stream) {
    if (!depth || !A || !t || !g_points || !d_depth || !partials || bad_shape(B, H, W)) return MGN_EINVAL;
    hipLaunchKernelGGL(reconstruct_kernel<true>, grid_of(B, H, W), dim3(TPB), 0, (hipStream_t)stream, depth, A, t, g_points, H, W,
                       (float*)nullptr, d_depth, partials);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_project_fwd(const float* points, const float* A, const float* t, int B, int H, int W, float* coords, void* stream) {
    if (!points || !A || !t || !coords || bad_shape(B, H, W)) return MGN_EINVAL;
    hipLaunchKernelGGL(project_kernel<false>, grid_of(B, H, W), dim3(TPB), 0, (hipStream_t)stream, points, A, t,
                       (const float*)nullptr, H, W, coords, (float*)nullptr, (float*)nullptr);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_project_bwd(const float* points, const float* A, const float* t, const float* g_coords, int B, int H, int W,
                    float* d_points, float* partials, void* stream) {
    if (!points || !A || !t || !g_coords || !d_points || !partials || bad_shape(B, H, W)) return MGN_EINVAL;
    hipLaunchKernelGGL(project_kernel<true>, grid_of(B, H, W), dim3(TPB), 0, (hipStream_t)stream, points, A, t, g_coords, H, W,
                       (float*)nullptr, d_points, partials);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

}  // extern "C"

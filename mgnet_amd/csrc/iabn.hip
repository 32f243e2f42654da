// iabn.hip -- in-place activated batch norm (InPlaceABNSync) for MI355X, channels-last activations.
//
// Replaces inplace_abn.InPlaceABNSync (pip inplace-abn>=1.1.0, the only native code on the reference's training
// path; 68 call sites: mgnet/modeling/res_net.py:35,49,59,103, layers.py:63,71,117,209,242,253,291).  That package is
// not vendored in the reference; semantics restated from its published behaviour:
//     y = act( (|gamma|+eps) * (x - mean) / sqrt(var + eps) + beta ),  act = leaky_relu(slope) | identity
//     batch statistics over N*H*W (biased var; unbiased for running_var), synchronised across ranks by the caller
//     backward re-derives x_hat from the OUTPUT y by inverting the activation (no saved input -> "in place")
//
// Layout: x is [M = N*H*W, C] with C contiguous (torch channels_last), bf16 or fp32; statistics fp32.
// All kernels are HBM-streaming; 16-byte accesses per lane; per-channel reductions are column sums:
//   a block of 256 threads covers (256 / (C/VEC)) rows per pass, lanes along C => fully coalesced rows,
//   per-thread register accumulators, one LDS reduction per block, block partials in the workspace, fp64-free
//   deterministic finalize (fixed order).
// Numerics: sums are taken around a per-channel shift (the first row) so that var does not cancel (the naive
// E[x^2]-mean^2 loses the 1x1-spatial layers entirely, see tests); cross-rank combination uses Chan's formula.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgnet_hip.h"

namespace {

constexpr int TPB = 256;
constexpr int MAX_BLOCKS = 2048;

template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    using raw = float4;
    __device__ static void load(const float* p, float (&v)[4]) {
        const float4 r = *reinterpret_cast<const float4*>(p);
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    }
    __device__ static void store(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Vec<__hip_bfloat16> {
    static constexpr int N = 8;
    __device__ static void load(const __hip_bfloat16* p, float (&v)[8]) {
        const uint4 r = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = __uint_as_float(w[k] << 16);
            v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
        }
    }
    __device__ static uint32_t pack(float a, float b) {  // round-to-nearest-even bf16 x2
        auto rne = [](float f) -> uint32_t {
            uint32_t u = __float_as_uint(f);
            if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;  // NaN
            return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
        };
        return rne(a) | (rne(b) << 16);
    }
    __device__ static void store(__hip_bfloat16* p, const float (&v)[8]) {
        uint4 r;
        r.x = pack(v[0], v[1]); r.y = pack(v[2], v[3]); r.z = pack(v[4], v[5]); r.w = pack(v[6], v[7]);
        *reinterpret_cast<uint4*>(p) = r;
    }
};

// ------------------------------------------------------------------------------------------------------
// column sums of f(row) for two quantities.  partials: [gridDim.x][2][C]
// ------------------------------------------------------------------------------------------------------
template <typename T, typename F>
__device__ __forceinline__ void column_sums2(long M, int C, float* partials, F&& row_values) {
    constexpr int V = Vec<T>::N;
    __shared__ float sh[TPB * 2 * 8];
    const int tpr = C / V;            // threads per row
    const int rpb = TPB / tpr;        // rows per block pass
    const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr;
    float a[V], b[V];
#pragma unroll
    for (int k = 0; k < V; ++k) a[k] = b[k] = 0.f;
    if (ty < rpb) {
        for (long r = (long)blockIdx.x * rpb + ty; r < M; r += (long)gridDim.x * rpb) row_values(r, tx * V, a, b);
    }
    // reduce over ty
#pragma unroll
    for (int k = 0; k < V; ++k) {
        sh[(ty * tpr + tx) * 2 * V + k] = a[k];
        sh[(ty * tpr + tx) * 2 * V + V + k] = b[k];
    }
    __syncthreads();
    if (ty == 0) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float sa = 0.f, sb = 0.f;
            for (int y = 0; y < rpb; ++y) {
                sa += sh[(y * tpr + tx) * 2 * V + k];
                sb += sh[(y * tpr + tx) * 2 * V + V + k];
            }
            partials[((size_t)blockIdx.x * 2 + 0) * C + tx * V + k] = sa;
            partials[((size_t)blockIdx.x * 2 + 1) * C + tx * V + k] = sb;
        }
    }
}

// forward statistics: sum (x - shift), sum (x - shift)^2 with shift = x[0, c]
template <typename T>
__global__ __launch_bounds__(TPB) void iabn_stats_partial(const T* __restrict__ x, long M, int C, float* partials) {
    constexpr int V = Vec<T>::N;
    float s[V];
    Vec<T>::load(x + (threadIdx.x % (C / V)) * V, s);
    column_sums2<T>(M, C, partials, [&](long r, int c0, float (&a)[V], float (&b)[V]) {
        float v[V];
        Vec<T>::load(x + r * C + c0, v);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float d = v[k] - s[k];
            a[k] += d;
            b[k] += d * d;
        }
    });
}

// sum the block partials [nblk][2][C] for 32 channels per block: 8 slices of blocks in parallel, then LDS (fixed order)
__device__ __forceinline__ void reduce_partials(const float* partials, int nblk, int C, int c, int slice, float& s1, float& s2) {
    __shared__ float r1[8][32], r2[8][32];
    float a = 0.f, b = 0.f;
    if (c < C)
        for (int k = slice; k < nblk; k += 8) {
            a += partials[((size_t)k * 2 + 0) * C + c];
            b += partials[((size_t)k * 2 + 1) * C + c];
        }
    r1[slice][threadIdx.x] = a;
    r2[slice][threadIdx.x] = b;
    __syncthreads();
    s1 = s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1 += r1[k][threadIdx.x]; s2 += r2[k][threadIdx.x]; }
}

// stats[3][C] = {count, mean, M2} of THIS rank.  block (32, 8), grid C/32
template <typename T>
__global__ void iabn_stats_final(const T* __restrict__ x, const float* partials, int nblk, long M, int C, float* stats) {
    const int c = blockIdx.x * 32 + threadIdx.x;
    float s1, s2;
    reduce_partials(partials, nblk, C, c, threadIdx.y, s1, s2);
    if (c >= C || threadIdx.y != 0) return;
    float sv[Vec<T>::N];
    Vec<T>::load(x + (c / Vec<T>::N) * Vec<T>::N, sv);
    const float shift = sv[c % Vec<T>::N];
    const float n = (float)M;
    const float md = s1 / n;
    stats[c] = n;
    stats[C + c] = shift + md;
    stats[2 * C + c] = fmaxf(s2 - s1 * md, 0.f);  // sum (x-mean)^2
}

// combine R ranks (Chan), update running stats, emit scale/offset/rstd for the apply + backward kernels.
// gathered: [R][3][C];  out: scale[C], offset[C], saved[2][C] = {mean, rstd}
__global__ void iabn_combine(const float* gathered, int R, int C, const float* weight, const float* bias, float eps,
                             float momentum, float* running_mean, float* running_var, float* scale, float* offset,
                             float* saved) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int r = 0; r < R; ++r) {
        const float nr = gathered[((size_t)r * 3 + 0) * C + c], mr = gathered[((size_t)r * 3 + 1) * C + c],
                    qr = gathered[((size_t)r * 3 + 2) * C + c];
        if (nr <= 0.f) continue;
        const float nt = n + nr, d = mr - mean;
        mean += d * (nr / nt);
        m2 += qr + d * d * (n * nr / nt);
        n = nt;
    }
    const float var = m2 / n;
    const float rstd = rsqrtf(var + eps);
    const float g = fabsf(weight[c]) + eps;
    scale[c] = g * rstd;
    offset[c] = bias[c] - mean * g * rstd;
    saved[c] = mean;
    saved[C + c] = rstd;
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
    }
}

// eval mode: scale/offset from the running statistics
__global__ void iabn_eval_coeffs(int C, const float* weight, const float* bias, const float* running_mean,
                                 const float* running_var, float eps, float* scale, float* offset) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float g = (fabsf(weight[c]) + eps) * rsqrtf(running_var[c] + eps);
    scale[c] = g;
    offset[c] = bias[c] - running_mean[c] * g;
}

template <typename T>
__global__ __launch_bounds__(TPB) void iabn_apply(const T* __restrict__ x, T* __restrict__ y, long M, int C,
                                                  const float* __restrict__ scale, const float* __restrict__ offset,
                                                  int leaky, float slope) {
    constexpr int V = Vec<T>::N;
    const long nvec = M * C / V;
    const int cv = C / V;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cv) * V;
        float v[V];
        Vec<T>::load(x + i * V, v);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float z = fmaf(v[k], scale[c0 + k], offset[c0 + k]);
            if (leaky) z = z > 0.f ? z : z * slope;
            v[k] = z;
        }
        Vec<T>::store(y + i * V, v);
    }
}

// backward pass 1: per-channel sum dz and sum dz * x_hat, with z = act^-1(y), dz = dy * act'(z), x_hat = (z - beta)/gamma'
template <typename T>
__global__ __launch_bounds__(TPB) void iabn_bwd_partial(const T* __restrict__ y, const T* __restrict__ dy, long M, int C,
                                                        const float* __restrict__ weight, const float* __restrict__ bias,
                                                        float eps, int leaky, float slope, float* partials) {
    constexpr int V = Vec<T>::N;
    const float inv_slope = 1.f / slope;
    column_sums2<T>(M, C, partials, [&](long r, int c0, float (&a)[V], float (&b)[V]) {
        float yv[V], gv[V];
        Vec<T>::load(y + r * C + c0, yv);
        Vec<T>::load(dy + r * C + c0, gv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float z = yv[k], dz = gv[k];
            if (leaky && z < 0.f) { z *= inv_slope; dz *= slope; }
            const float xh = (z - bias[c0 + k]) / (fabsf(weight[c0 + k]) + eps);
            a[k] += dz;
            b[k] += dz * xh;
        }
    });
}

__global__ void iabn_bwd_final(const float* partials, int nblk, int C, const float* weight, float* sums, float* dwb) {
    const int c = blockIdx.x * 32 + threadIdx.x;
    float s1, s2;
    reduce_partials(partials, nblk, C, c, threadIdx.y, s1, s2);
    if (c >= C || threadIdx.y != 0) return;
    sums[c] = s1;
    sums[C + c] = s2;
    if (dwb) {  // local parameter gradients: d bias = sum dz ; d weight = sign(weight) * sum dz*x_hat  (gamma' = |weight| + eps)
        const float w = weight[c];
        dwb[c] = s2 * (float)((w > 0.f) - (w < 0.f));
        dwb[C + c] = s1;
    }
}

// backward pass 2: dx = gamma' * rstd * (dz - sum_dz/n - x_hat * sum_dzxh/n)     (sums are GLOBAL over ranks, n too)
template <typename T>
__global__ __launch_bounds__(TPB) void iabn_bwd_apply(const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ dx,
                                                      long M, int C, const float* __restrict__ weight,
                                                      const float* __restrict__ bias, const float* __restrict__ saved,
                                                      const float* __restrict__ sums, float inv_n, float eps, int leaky,
                                                      float slope) {
    constexpr int V = Vec<T>::N;
    const long nvec = M * C / V;
    const int cv = C / V;
    const float inv_slope = 1.f / slope;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cv) * V;
        float yv[V], gv[V];
        Vec<T>::load(y + i * V, yv);
        Vec<T>::load(dy + i * V, gv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int c = c0 + k;
            float z = yv[k], dz = gv[k];
            if (leaky && z < 0.f) { z *= inv_slope; dz *= slope; }
            const float g = fabsf(weight[c]) + eps;
            const float xh = (z - bias[c]) / g;
            gv[k] = g * saved[C + c] * (dz - sums[c] * inv_n - xh * sums[C + c] * inv_n);
        }
        Vec<T>::store(dx + i * V, gv);
    }
}

inline int grid_for(long nvec) {
    long b = (nvec + TPB - 1) / TPB;
    return (int)(b < 1 ? 1 : (b > MAX_BLOCKS ? MAX_BLOCKS : b));
}

inline int check_shape(long M, int C, int dtype) {
    if (M < 1 || C < 8 || (dtype != 0 && dtype != 1)) return MGN_EINVAL;
    const int V = dtype == 1 ? 8 : 4;
    if (C % V != 0 || C / V > TPB || TPB % (C / V) != 0) return MGN_EINVAL;
    return MGN_OK;
}

inline int stat_blocks(long M, int C, int dtype) {
    const int V = dtype == 1 ? 8 : 4;
    const int rpb = TPB / (C / V);
    long b = (M + (long)rpb * 8 - 1) / ((long)rpb * 8);  // >= 8 rows per thread before adding blocks
    return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

}  // namespace

extern "C" {

int mgn_iabn_workspace_bytes(long M, int C, int dtype, size_t* bytes) {
    int rc = check_shape(M, C, dtype);
    if (rc != MGN_OK) return rc;
    if (!bytes) return MGN_EINVAL;
    *bytes = sizeof(float) * 2 * (size_t)C * 1024;
    return MGN_OK;
}

int mgn_iabn_stats(const void* x, int dtype, long M, int C, float* stats, void* ws, size_t ws_bytes, void* stream_) {
    int rc = check_shape(M, C, dtype);
    if (rc != MGN_OK) return rc;
    if (!x || !stats || !ws) return MGN_EINVAL;
    if (ws_bytes < sizeof(float) * 2 * (size_t)C * 1024) return MGN_ENOSPC;
    hipStream_t s = (hipStream_t)stream_;
    const int nb = stat_blocks(M, C, dtype);
    if (dtype == 1) {
        hipLaunchKernelGGL(iabn_stats_partial<__hip_bfloat16>, dim3(nb), dim3(TPB), 0, s, (const __hip_bfloat16*)x, M, C, (float*)ws);
        hipLaunchKernelGGL(iabn_stats_final<__hip_bfloat16>, dim3((C + 31) / 32), dim3(32, 8), 0, s, (const __hip_bfloat16*)x, (const float*)ws, nb, M, C, stats);
    } else {
        hipLaunchKernelGGL(iabn_stats_partial<float>, dim3(nb), dim3(TPB), 0, s, (const float*)x, M, C, (float*)ws);
        hipLaunchKernelGGL(iabn_stats_final<float>, dim3((C + 31) / 32), dim3(32, 8), 0, s, (const float*)x, (const float*)ws, nb, M, C, stats);
    }
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_iabn_combine(const float* gathered, int n_ranks, int C, const float* weight, const float* bias, float eps,
                     float momentum, float* running_mean, float* running_var, float* scale, float* offset, float* saved,
                     void* stream_) {
    if (!gathered || n_ranks < 1 || C < 1 || !weight || !bias || !scale || !offset || !saved) return MGN_EINVAL;
    hipLaunchKernelGGL(iabn_combine, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream_, gathered, n_ranks, C, weight, bias, eps,
                       momentum, running_mean, running_var, scale, offset, saved);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_iabn_eval_coeffs(int C, const float* weight, const float* bias, const float* running_mean, const float* running_var,
                         float eps, float* scale, float* offset, void* stream_) {
    if (C < 1 || !weight || !bias || !running_mean || !running_var || !scale || !offset) return MGN_EINVAL;
    hipLaunchKernelGGL(iabn_eval_coeffs, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream_, C, weight, bias, running_mean,
                       running_var, eps, scale, offset);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_iabn_apply(const void* x, void* y, int dtype, long M, int C, const float* scale, const float* offset, int activation,
                   float slope, void* stream_) {
    int rc = check_shape(M, C, dtype);
    if (rc != MGN_OK) return rc;
    if (!x || !y || !scale || !offset || activation < 0 || activation > 1) return MGN_EINVAL;
    hipStream_t s = (hipStream_t)stream_;
    if (dtype == 1)
        hipLaunchKernelGGL(iabn_apply<__hip_bfloat16>, dim3(grid_for(M * C / 8)), dim3(TPB), 0, s, (const __hip_bfloat16*)x,
                           (__hip_bfloat16*)y, M, C, scale, offset, activation, slope);
    else
        hipLaunchKernelGGL(iabn_apply<float>, dim3(grid_for(M * C / 4)), dim3(TPB), 0, s, (const float*)x, (float*)y, M, C, scale,
                           offset, activation, slope);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_iabn_bwd_reduce(const void* y, const void* dy, int dtype, long M, int C, const float* weight, const float* bias,
                        float eps, int activation, float slope, float* sums, float* dwb, void* ws, size_t ws_bytes, void* stream_) {
    int rc = check_shape(M, C, dtype);
    if (rc != MGN_OK) return rc;
    if (!y || !dy || !weight || !bias || !sums || !ws) return MGN_EINVAL;
    if (ws_bytes < sizeof(float) * 2 * (size_t)C * 1024) return MGN_ENOSPC;
    hipStream_t s = (hipStream_t)stream_;
    const int nb = stat_blocks(M, C, dtype);
    if (dtype == 1)
        hipLaunchKernelGGL(iabn_bwd_partial<__hip_bfloat16>, dim3(nb), dim3(TPB), 0, s, (const __hip_bfloat16*)y,
                           (const __hip_bfloat16*)dy, M, C, weight, bias, eps, activation, slope, (float*)ws);
    else
        hipLaunchKernelGGL(iabn_bwd_partial<float>, dim3(nb), dim3(TPB), 0, s, (const float*)y, (const float*)dy, M, C, weight, bias,
                           eps, activation, slope, (float*)ws);
    hipLaunchKernelGGL(iabn_bwd_final, dim3((C + 31) / 32), dim3(32, 8), 0, s, (const float*)ws, nb, C, weight, sums, dwb);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_iabn_bwd_apply(const void* y, const void* dy, void* dx, int dtype, long M, int C, const float* weight, const float* bias,
                       const float* saved, const float* sums, float total_count, float eps, int activation, float slope,
                       void* stream_) {
    int rc = check_shape(M, C, dtype);
    if (rc != MGN_OK) return rc;
    if (!y || !dy || !dx || !weight || !bias || !saved || !sums || !(total_count > 0.f)) return MGN_EINVAL;
    hipStream_t s = (hipStream_t)stream_;
    if (dtype == 1)
        hipLaunchKernelGGL(iabn_bwd_apply<__hip_bfloat16>, dim3(grid_for(M * C / 8)), dim3(TPB), 0, s, (const __hip_bfloat16*)y,
                           (const __hip_bfloat16*)dy, (__hip_bfloat16*)dx, M, C, weight, bias, saved, sums, 1.f / total_count, eps,
                           activation, slope);
    else
        hipLaunchKernelGGL(iabn_bwd_apply<float>, dim3(grid_for(M * C / 4)), dim3(TPB), 0, s, (const float*)y, (const float*)dy,
                           (float*)dx, M, C, weight, bias, saved, sums, 1.f / total_count, eps, activation, slope);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

}  // extern "C"

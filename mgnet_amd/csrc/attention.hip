// attention.hip -- the channel-attention vectors of MGNet's decoders as single-block kernels.
//
// Replaces, for the [N, C, 1, 1] pooled vectors only,
//   mgnet/modeling/layers.py:248-258,262-267  AttentionRefinementModule.channel_attention =
//        FastGlobalAvgPool2d -> Conv2d(C, C, 1x1, norm=InPlaceABNSync(activation="identity")) -> Sigmoid
//   mgnet/modeling/layers.py:297-311,315-322  FeatureFusionModule.channel_attention =
//        FastGlobalAvgPool2d -> Conv2d(C, C, 1x1, activation=ReLU) -> Conv2d(C, C, 1x1) -> Sigmoid
// i.e. a (C x K) matrix applied to N <= 64 vectors, an optional batch norm over those N samples, an activation.  As
// separate conv / IABN / element-wise launches each site cost ~25 kernels of a few microseconds (a third of all launches
// of a training step, each followed by ~2.7 us of dispatch gap); here a layer is ONE launch forward and ONE backward.
// fp32 throughout, straight from the fp32 master weights (no bf16 layout).  Single process only for the batch-norm
// variant: with more ranks the statistics need the cross-rank exchange and the host keeps the general path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgnet_hip.h"

namespace {

constexpr int VT = 256;        // threads per block
constexpr int CB = 16;         // output channels per block (the batch norm is per channel, so blocks are independent)
constexpr int VMAXN = 64;      // samples

__device__ __forceinline__ float act_fwd(float y, int act) {
    if (act == 1) return y > 0.f ? y : 0.f;
    if (act == 2) return 1.f / (1.f + __expf(-y));
    return y;
}

// out[n][c] = act( bn( sum_k W[c][k] * in[n][k] ) ) for the block's CB channels
__global__ __launch_bounds__(VT) void vec_linear_fwd(const float* __restrict__ in, const float* __restrict__ W, int N, int K, int C, int act,
                                                     const float* __restrict__ bn_w, const float* __restrict__ bn_b, float* running_mean,
                                                     float* running_var, int training, float momentum, float eps, float* __restrict__ out,
                                                     float* __restrict__ xhat, float* __restrict__ rstd_out) {
    extern __shared__ float sm[];           // in [N][K] | z [N][CB]
    float* sin = sm;
    float* sz = sm + (size_t)N * K;
    const int c0 = blockIdx.x * CB;
    for (int i = threadIdx.x; i < N * K; i += VT) sin[i] = in[i];
    __syncthreads();
    // thread = (channel, k-part): partial dot products for all samples, combined through LDS
    // (k fastest across lanes: a wave reads 4 weight rows x 16 consecutive k = four 64-byte segments per load instead of sixteen rows)
    const int nparts = VT / CB, part = threadIdx.x % nparts, cl = threadIdx.x / nparts;   // 16 parts
    const int c = c0 + cl;
    float* sp = sz + (size_t)N * CB;        // partials [nparts][N][CB]
    {
        float acc[VMAXN / 8 > 8 ? 8 : 8];
        for (int n0 = 0; n0 < N; n0 += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
            if (c < C)
#pragma unroll 8   // (eight weight loads in flight: left rolled, every iteration waits for its own load -- 32 round trips = the kernel's 14 us)
                for (int k = part; k < K; k += nparts) {
                    const float w = W[(size_t)c * K + k];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (n0 + j < N) acc[j] = fmaf(w, sin[(n0 + j) * K + k], acc[j]);
                }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (n0 + j < N) sp[((size_t)part * N + n0 + j) * CB + cl] = acc[j];
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < N * CB; o += VT) {
        float z = 0.f;
        for (int q = 0; q < nparts; ++q) z += sp[(size_t)q * N * CB + o];
        sz[o] = z;
    }
    __syncthreads();
    if (bn_w && threadIdx.x < CB && c0 + threadIdx.x < C) {   // InPlaceABNSync semantics over the N samples
        const int cc = threadIdx.x, cg = c0 + cc;
        float mean, var;
        if (training) {
            float s1 = 0.f;
            for (int n = 0; n < N; ++n) s1 += sz[n * CB + cc];
            mean = s1 / N;
            float s2 = 0.f;
            for (int n = 0; n < N; ++n) { const float d = sz[n * CB + cc] - mean; s2 += d * d; }
            var = s2 / N;
            if (running_mean) {
                running_mean[cg] = (1.f - momentum) * running_mean[cg] + momentum * mean;
                running_var[cg] = (1.f - momentum) * running_var[cg] + momentum * var * ((float)N / fmaxf((float)N - 1.f, 1.f));
            }
        } else {
            mean = running_mean[cg];
            var = running_var[cg];
        }
        const float rstd = rsqrtf(var + eps), g = fabsf(bn_w[cg]) + eps, b = bn_b[cg];
        if (rstd_out) rstd_out[cg] = rstd;
        for (int n = 0; n < N; ++n) {
            const float xh = (sz[n * CB + cc] - mean) * rstd;
            if (xhat) xhat[n * C + cg] = xh;
            sz[n * CB + cc] = g * xh + b;
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < N * CB; o += VT) {
        const int cc = o % CB, n = o / CB;
        if (c0 + cc < C) out[n * C + c0 + cc] = act_fwd(sz[o], act);
    }
}

// backward of the block's CB channels: dW rows, d bn weight / bias, and this block's PARTIAL din [blockIdx][N][K]
__global__ __launch_bounds__(VT) void vec_linear_bwd(const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ in,
                                                     const float* __restrict__ W, int N, int K, int C, int act, const float* __restrict__ bn_w,
                                                     const float* __restrict__ xhat, const float* __restrict__ rstd, float eps,
                                                     float* __restrict__ dW, float* __restrict__ din_parts, float* __restrict__ dbn_w,
                                                     float* __restrict__ dbn_b) {
    extern __shared__ float sm[];           // in [N][K] | dz [N][CB]
    float* sin = sm;
    float* dz = sm + (size_t)N * K;
    const int c0 = blockIdx.x * CB;
    for (int i = threadIdx.x; i < N * K; i += VT) sin[i] = in[i];
    for (int o = threadIdx.x; o < N * CB; o += VT) {
        const int cc = o % CB, n = o / CB;
        float d = 0.f;
        if (c0 + cc < C) {
            const float y = out[n * C + c0 + cc];
            d = dout[n * C + c0 + cc];
            if (act == 1) d = y > 0.f ? d : 0.f;
            else if (act == 2) d *= y * (1.f - y);
        }
        dz[o] = d;
    }
    __syncthreads();
    if (bn_w && threadIdx.x < CB && c0 + threadIdx.x < C) {   // training-mode batch norm backward over the N samples
        const int cc = threadIdx.x, cg = c0 + cc;
        float s1 = 0.f, s2 = 0.f;
        for (int n = 0; n < N; ++n) { const float d = dz[n * CB + cc]; s1 += d; s2 += d * xhat[n * C + cg]; }
        const float w = bn_w[cg], g = fabsf(w) + eps, r = rstd[cg];
        dbn_b[cg] = s1;
        dbn_w[cg] = s2 * (float)((w > 0.f) - (w < 0.f));
        const float m1 = s1 / N, m2 = s2 / N;
        for (int n = 0; n < N; ++n) dz[n * CB + cc] = g * r * (dz[n * CB + cc] - m1 - xhat[n * C + cg] * m2);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < CB * K; o += VT) {           // dW[c][k] = sum_n dz[n][c] * in[n][k]
        const int k = o % K, cc = o / K;
        if (c0 + cc >= C) continue;
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc = fmaf(dz[n * CB + cc], sin[n * K + k], acc);
        dW[(size_t)(c0 + cc) * K + k] = acc;
    }
    float* dp = din_parts + (size_t)blockIdx.x * N * K;
    // partial din[n][k] = sum_{c in block} dz[n][c] * W[c][k]: a thread owns column k -- its CB weights are loaded once (all in flight
    // together) and reused for the N samples (per (n, k) output they were N x CB dependent round trips: ~16 of the kernel's 25 us)
    for (int k = threadIdx.x; k < K; k += VT) {
        float w[CB];
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) w[cc] = c0 + cc < C ? W[(size_t)(c0 + cc) * K + k] : 0.f;
        for (int n = 0; n < N; ++n) {
            float acc = 0.f;
#pragma unroll
            for (int cc = 0; cc < CB; ++cc) acc = fmaf(dz[n * CB + cc], w[cc], acc);
            dp[(size_t)n * K + k] = acc;
        }
    }
}

// din = scale * sum over the blocks' partials (fixed order)
__global__ void vec_sum_parts(const float* __restrict__ parts, int nparts, int n, float scale, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // (independent chains: the loads of all parts in flight at once; fixed order)
    int q = 0;
#pragma unroll 4
    for (; q + 3 < nparts; q += 4) {
        a0 += parts[(size_t)q * n + i]; a1 += parts[(size_t)(q + 1) * n + i]; a2 += parts[(size_t)(q + 2) * n + i]; a3 += parts[(size_t)(q + 3) * n + i];
    }
    for (; q < nparts; ++q) a0 += parts[(size_t)q * n + i];
    out[i] = ((a0 + a1) + (a2 + a3)) * scale;
}

inline bool shape_ok(int N, int K, int C) { return N >= 1 && N <= VMAXN && K >= 1 && C >= 1 && (size_t)N * K * 4 <= 96 * 1024; }
inline size_t fwd_lds(int N, int K) { return sizeof(float) * ((size_t)N * K + (size_t)N * CB + (size_t)(VT / CB) * N * CB); }
inline size_t bwd_lds(int N, int K) { return sizeof(float) * ((size_t)N * K + (size_t)N * CB); }

void set_attrs() {
    static bool attr = false;
    if (attr) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vec_linear_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vec_linear_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
}

}  // namespace

extern "C" {

int mgn_vec_linear_fwd(const float* in, const float* W, int N, int K, int C, int act, const float* bn_weight, const float* bn_bias,
                       float* running_mean, float* running_var, int training, float momentum, float eps, float* out, float* xhat,
                       float* rstd, void* stream) {
    if (!in || !W || !out || act < 0 || act > 2) return MGN_EINVAL;
    if (!shape_ok(N, K, C)) return MGN_ENOTSUP;
    if (bn_weight && (!bn_bias || (!training && (!running_mean || !running_var)) || (training && (!xhat || !rstd)))) return MGN_EINVAL;
    set_attrs();
    hipLaunchKernelGGL(vec_linear_fwd, dim3((C + CB - 1) / CB), dim3(VT), fwd_lds(N, K), (hipStream_t)stream, in, W, N, K, C, act, bn_weight,
                       bn_bias, running_mean, running_var, training, momentum, eps, out, xhat, rstd);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_vec_linear_bwd_workspace_bytes(int N, int K, int C, size_t* bytes) {
    if (!bytes || !shape_ok(N, K, C)) return MGN_EINVAL;
    *bytes = sizeof(float) * (size_t)((C + CB - 1) / CB) * N * K;
    return MGN_OK;
}

int mgn_vec_linear_bwd(const float* dout, const float* out, const float* in, const float* W, int N, int K, int C, int act,
                       const float* bn_weight, const float* xhat, const float* rstd, float eps, float din_scale, float* dW, float* din,
                       float* dbn_weight, float* dbn_bias, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dout || !out || !in || !W || !dW || !din || !workspace || act < 0 || act > 2) return MGN_EINVAL;
    if (!shape_ok(N, K, C)) return MGN_ENOTSUP;
    if (bn_weight && (!xhat || !rstd || !dbn_weight || !dbn_bias)) return MGN_EINVAL;
    const int nblk = (C + CB - 1) / CB;
    if (workspace_bytes < sizeof(float) * (size_t)nblk * N * K) return MGN_ENOSPC;
    set_attrs();
    hipLaunchKernelGGL(vec_linear_bwd, dim3(nblk), dim3(VT), bwd_lds(N, K), (hipStream_t)stream, dout, out, in, W, N, K, C, act, bn_weight, xhat,
                       rstd, eps, dW, (float*)workspace, dbn_weight, dbn_bias);
    hipLaunchKernelGGL(vec_sum_parts, dim3((N * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nblk, N * K, din_scale,
                       din);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

}  // extern "C"

// optim.hip -- full-model gradient clipping + Adam as two streaming passes over flat fp32 buckets.
//
// Replaces tools/train_net.py:108-154 (torch.optim.Adam wrapped in FullModelGradientClippingOptimizer:
// torch.nn.utils.clip_grad_norm_(all params, 0.01) then Adam.step) -- ~350 tensors x ~8 small kernels per step in the
// reference.  Parameters, gradients and both moments live in flat buckets (the same buckets the RCCL all-reduce
// uses); per-tensor learning rate / weight decay come from a table with one entry per 1024-element chunk (tensors are
// padded to chunk multiples inside a bucket).
//   pass 1  mgn_sqnorm        : sum g^2 per bucket -> block partials (deterministic two-stage)
//   pass 1b mgn_clip_coef     : total_norm = sqrt(sum) * grad_scale ; coef = min(1, max_norm / (total_norm + 1e-6))
//   pass 2  mgn_adam_step     : g' = g*grad_scale*coef (+ wd*p); m,v update; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// HBM: pass 1 reads 4 B/param, pass 2 reads 16 + writes 12 B/param: 32 B/param -> 31 M params = 0.99 GB per step.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgnet_hip.h"

namespace {

constexpr int TPB = 256;
constexpr int CHUNK = 1024;  // elements per lr/wd table entry

__global__ __launch_bounds__(TPB) void sqnorm_kernel(const float* __restrict__ g, long n, float* partials) {
    __shared__ float sh[TPB / 64];
    float acc = 0.f;
    const long nv = n / 4;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nv; i += (long)gridDim.x * TPB) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0)
        for (long i = nv * 4 + threadIdx.x; i < n; i += TPB) acc += g[i] * g[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ void clip_coef_kernel(const float* partials, int n, float max_norm, float grad_scale, float* out) {
    __shared__ double sh[TPB];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += TPB) acc += (double)partials[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = TPB / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float total = (float)sqrt(sh[0]) * grad_scale;
        float coef = max_norm > 0.f ? max_norm / (total + 1e-6f) : 1.f;  // torch.nn.utils.clip_grad_norm_
        out[0] = fminf(coef, 1.f);
        out[1] = total;
    }
}

// The same with dynamic loss scaling (torch.cuda.amp.GradScaler semantics, all on the device: no host synchronisation, and the
// launch can live in a captured graph).  The gradients in the buckets are S times the true ones:
//   total    = ||g|| * grad_scale / S;  found_inf = !isfinite(total)
//   out      = { clip coefficient / S (so the optimizer's g * grad_scale * out[0] is the clipped TRUE gradient), total, found_inf }
//   scaler   = { S, growth tracker, t }:  found_inf -> S *= 0.5, tracker = 0, t unchanged (the optimizer skips the step);
//              else t += 1, tracker += 1, and after `growth_interval` clean steps S *= 2
//   hyper    = Adam's bias corrections for the new t
__global__ void clip_coef_scaled_kernel(const float* partials, int n, float max_norm, float grad_scale, float beta1, float beta2,
                                        int growth_interval, float* scaler, float* hyper, float* out) {
    __shared__ double sh[TPB];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += TPB) acc += (double)partials[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = TPB / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float S = scaler[0];
        const float total = (float)sqrt(sh[0]) * grad_scale / S;
        const bool bad = !(fabsf(total) <= 3.0e38f);     // inf or NaN anywhere in the gradients
        float coef = max_norm > 0.f ? max_norm / (total + 1e-6f) : 1.f;
        out[0] = bad ? 0.f : fminf(coef, 1.f) / S;
        out[1] = total;
        out[2] = bad ? 1.f : 0.f;
        float tracker = scaler[1], t = scaler[2];
        if (bad) {
            scaler[0] = fmaxf(S * 0.5f, 1.f);
            tracker = 0.f;
        } else {
            t += 1.f;
            tracker += 1.f;
            if (tracker >= (float)growth_interval) {
                scaler[0] = fminf(S * 2.f, 16777216.f);
                tracker = 0.f;
            }
        }
        scaler[1] = tracker;
        scaler[2] = t;
        const double tt = t < 1.f ? 1.0 : (double)t;
        hyper[0] = (float)(1.0 / (1.0 - pow((double)beta1, tt)));
        hyper[1] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, tt)));
    }
}

__global__ __launch_bounds__(TPB) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, const float* __restrict__ chunk_lr,
                                                   const float* __restrict__ chunk_wd, float beta1, float beta2, float eps,
                                                   float inv_bc1, float inv_sqrt_bc2, const float* __restrict__ hyper,
                                                   const float* __restrict__ clip, float grad_scale) {
    if (hyper) {   // bias corrections of the current step from device memory (a captured launch is replayed for every step)
        inv_bc1 = hyper[0];
        inv_sqrt_bc2 = hyper[1];
        if (clip[2] != 0.f) return;   // dynamic loss scaling found inf/NaN gradients: the step is skipped (GradScaler.step)
    }
    const float gs = grad_scale * clip[0];
    const long nv = n / 4;  // buckets are padded to CHUNK multiples, so n % 4 == 0
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nv; i += (long)gridDim.x * TPB) {
        const int ch = (int)((i * 4) / CHUNK);
        const float lr = chunk_lr[ch], wd = chunk_wd[ch];
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gk = gp[k] * gs;
            if (wd != 0.f) gk = fmaf(wd, pp[k], gk);
            mp[k] = fmaf(beta1, mp[k], (1.f - beta1) * gk);
            vp[k] = fmaf(beta2, vp[k], (1.f - beta2) * gk * gk);
            const float denom = sqrtf(vp[k]) * inv_sqrt_bc2 + eps;
            pp[k] -= lr * inv_bc1 * (mp[k] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
}

// AdamW (torch.optim.AdamW, tools/train_net.py:150-154): decoupled weight decay p <- p (1 - lr wd), then Adam on the plain gradient.
// SGD (torch.optim.SGD, tools/train_net.py:136-141; dampening 0): g <- g + wd p; buf <- mu buf + g; p <- p - lr (nesterov ? g + mu buf : buf).
// Same flat buckets, per-chunk lr / wd tables, clip coefficient and skip-on-overflow as adam_kernel.
template <int KIND>   // 1 AdamW, 2 SGD
__global__ __launch_bounds__(TPB) void optim_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, const float* __restrict__ chunk_lr,
                                                    const float* __restrict__ chunk_wd, float beta1, float beta2, float eps, int nesterov,
                                                    const float* __restrict__ hyper, const float* __restrict__ clip, float grad_scale) {
    const float inv_bc1 = hyper[0], inv_sqrt_bc2 = hyper[1];
    if (clip[2] != 0.f) return;   // dynamic loss scaling found inf/NaN gradients: the step is skipped
    const float gs = grad_scale * clip[0];
    const long nv = n / 4;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nv; i += (long)gridDim.x * TPB) {
        const int ch = (int)((i * 4) / CHUNK);
        const float lr = chunk_lr[ch], wd = chunk_wd[ch];
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x;
        if (KIND == 1) {
            float4 vv = reinterpret_cast<float4*>(v)[i];
            float* vp = &vv.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gk = gp[k] * gs;
                pp[k] *= 1.f - lr * wd;
                mp[k] = fmaf(beta1, mp[k], (1.f - beta1) * gk);
                vp[k] = fmaf(beta2, vp[k], (1.f - beta2) * gk * gk);
                const float denom = sqrtf(vp[k]) * inv_sqrt_bc2 + eps;
                pp[k] -= lr * inv_bc1 * (mp[k] / denom);
            }
            reinterpret_cast<float4*>(v)[i] = vv;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float gk = gp[k] * gs;
                if (wd != 0.f) gk = fmaf(wd, pp[k], gk);
                mp[k] = fmaf(beta1, mp[k], gk);                    // beta1 = momentum
                pp[k] -= lr * (nesterov ? fmaf(beta1, mp[k], gk) : mp[k]);
            }
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
    }
}

inline int blocks_for(long n) {
    long b = (n / 4 + TPB - 1) / TPB;
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace

extern "C" {

int mgn_optim_chunk(void) { return CHUNK; }

int mgn_sqnorm(const float* g, long n, float* partials, int max_partials, int* n_partials, void* stream) {
    if (!g || n < 1 || !partials || !n_partials) return MGN_EINVAL;
    const int nb = blocks_for(n);
    if (nb > max_partials) return MGN_ENOSPC;
    hipLaunchKernelGGL(sqnorm_kernel, dim3(nb), dim3(TPB), 0, (hipStream_t)stream, g, n, partials);
    *n_partials = nb;
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_clip_coef(const float* partials, int n_partials, float max_norm, float grad_scale, float* coef_and_norm, void* stream) {
    if (!partials || n_partials < 1 || !coef_and_norm) return MGN_EINVAL;
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(TPB), 0, (hipStream_t)stream, partials, n_partials, max_norm, grad_scale,
                       coef_and_norm);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_clip_coef_scaled(const float* partials, int n_partials, float max_norm, float grad_scale, float beta1, float beta2,
                         int growth_interval, float* scaler_state, float* hyper, float* coef_norm_found, void* stream) {
    if (!partials || n_partials < 1 || !scaler_state || !hyper || !coef_norm_found || growth_interval < 1) return MGN_EINVAL;
    hipLaunchKernelGGL(clip_coef_scaled_kernel, dim3(1), dim3(TPB), 0, (hipStream_t)stream, partials, n_partials, max_norm, grad_scale,
                       beta1, beta2, growth_interval, scaler_state, hyper, coef_norm_found);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_adam_step(float* p, const float* g, float* m, float* v, long n, const float* chunk_lr, const float* chunk_wd,
                  float beta1, float beta2, float eps, int step, const float* clip_coef, float grad_scale, void* stream) {
    if (!p || !g || !m || !v || n < 1 || n % CHUNK != 0 || !chunk_lr || !chunk_wd || !clip_coef || step < 1) return MGN_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n)), dim3(TPB), 0, (hipStream_t)stream, p, g, m, v, n, chunk_lr, chunk_wd, beta1,
                       beta2, eps, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), (const float*)nullptr, clip_coef, grad_scale);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_adam_step_dev(float* p, const float* g, float* m, float* v, long n, const float* chunk_lr, const float* chunk_wd,
                      float beta1, float beta2, float eps, const float* hyper, const float* clip_coef, float grad_scale,
                      void* stream) {
    if (!p || !g || !m || !v || n < 1 || n % CHUNK != 0 || !chunk_lr || !chunk_wd || !clip_coef || !hyper) return MGN_EINVAL;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n)), dim3(TPB), 0, (hipStream_t)stream, p, g, m, v, n, chunk_lr, chunk_wd, beta1,
                       beta2, eps, 0.f, 0.f, hyper, clip_coef, grad_scale);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_optim_step_dev(int kind, float* p, const float* g, float* m, float* v, long n, const float* chunk_lr, const float* chunk_wd,
                       float beta1, float beta2, float eps, int nesterov, const float* hyper, const float* clip_coef, float grad_scale,
                       void* stream) {
    if (kind == 0) return mgn_adam_step_dev(p, g, m, v, n, chunk_lr, chunk_wd, beta1, beta2, eps, hyper, clip_coef, grad_scale, stream);
    if ((kind != 1 && kind != 2) || !p || !g || !m || (kind == 1 && !v) || n < 1 || n % CHUNK != 0 || !chunk_lr || !chunk_wd || !clip_coef || !hyper)
        return MGN_EINVAL;
    if (kind == 1)
        hipLaunchKernelGGL(optim_kernel<1>, dim3(blocks_for(n)), dim3(TPB), 0, (hipStream_t)stream, p, g, m, v, n, chunk_lr, chunk_wd, beta1, beta2,
                           eps, 0, hyper, clip_coef, grad_scale);
    else
        hipLaunchKernelGGL(optim_kernel<2>, dim3(blocks_for(n)), dim3(TPB), 0, (hipStream_t)stream, p, g, m, v, n, chunk_lr, chunk_wd, beta1, 0.f,
                           0.f, nesterov ? 1 : 0, hyper, clip_coef, grad_scale);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

}  // extern "C"

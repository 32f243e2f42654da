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
#include "h16.h"

namespace {

constexpr int TPB = 256;
constexpr int MAX_BLOCKS = 2048;

template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    using Raw = float4;
    __device__ static Raw load_raw(const float* p) { return *reinterpret_cast<const float4*>(p); }
    __device__ static void unpack(const Raw& r, float (&v)[4]) { v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w; }
    __device__ static void load(const float* p, float (&v)[4]) { unpack(load_raw(p), v); }
    __device__ static void store(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
// 16-bit activations: `__hip_bfloat16` is only the pointer tag of "the 16-bit format of this translation unit" (h16.h:
// bf16 in iabn.hip, IEEE fp16 in iabn_f16.hip); the conversions below go through h16.h
template <> struct Vec<__hip_bfloat16> {
    static constexpr int N = 8;
    using Raw = uint4;
    __device__ static Raw load_raw(const __hip_bfloat16* p) { return *reinterpret_cast<const uint4*>(p); }
    __device__ static void unpack(const Raw& r, float (&v)[8]) {
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = mgn_lo2f(w[k]);
            v[2 * k + 1] = mgn_hi2f(w[k]);
        }
    }
    __device__ static void load(const __hip_bfloat16* p, float (&v)[8]) { unpack(load_raw(p), v); }
    __device__ static uint32_t pack(float a, float b) { return mgn_pack2(a, b); }   // round-to-nearest-even x2
    __device__ static void store(__hip_bfloat16* p, const float (&v)[8]) {
        uint4 r;
        r.x = pack(v[0], v[1]); r.y = pack(v[2], v[3]); r.z = pack(v[4], v[5]); r.w = pack(v[6], v[7]);
        *reinterpret_cast<uint4*>(p) = r;
    }
};

// ------------------------------------------------------------------------------------------------------
// column sums of two per-row quantities + finalize, in ONE launch.
//   * every block accumulates its rows (4 independent 16-byte loads in flight per thread) and writes its partial
//     sums [2][C] to the workspace;
//   * the block that finishes last (device-scope ticket counter; release/acquire fences around it) sums the block
//     partials in a fixed order -- deterministic whichever block that is -- and calls finalize(c, s1, s2) per channel.
// workspace: partials [gridDim.x][2][C] | final sums [2][C] at float offset 2*C*MAX_STAT_BLOCKS
// ------------------------------------------------------------------------------------------------------
constexpr int MAX_STAT_BLOCKS = 512;

// Channels are processed in slabs of SC (blockIdx.y): wide layers then have enough blocks to fill the chip and their
// finalize runs on one block PER SLAB in parallel.  C below = channels of the slab, ch0 = first channel of the slab.
template <typename T, bool DEEP, typename L, typename A, typename G>
__device__ __forceinline__ void column_sums2(long M, int C, float* ws_all, unsigned* counter_all, L&& load_row, A&& add_row, G&& finalize) {
    constexpr int V = Vec<T>::N;
    const int ch0 = blockIdx.y * C;
    float* ws = ws_all + (size_t)blockIdx.y * 2 * C * (MAX_STAT_BLOCKS + 1);
    unsigned* counter = counter_all + blockIdx.y;
    __shared__ __attribute__((aligned(16))) float sh[TPB * 2 * 8];
    __shared__ int is_last;
    const int tpr = C / V;            // threads per row
    const int rpb = TPB / tpr;        // rows per block pass
    const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr;
    float a[V], b[V];
#pragma unroll
    for (int k = 0; k < V; ++k) a[k] = b[k] = 0.f;
    {
        const long stride = (long)gridDim.x * rpb;
        long r = (long)blockIdx.x * rpb + ty;
        typename Vec<T>::Raw q0[2], q1[2], q2[2], q3[2], q4[2], q5[2], q6[2], q7[2];
        if (DEEP && r + 7 * stride < M) {
            // 8 independent rows in flight (single-tensor reductions), software-pipelined: the next group of 8 loads is issued
            // BEFORE the current group is reduced, so that a block with only a few groups per thread (the 30-130 MB layers)
            // does not drain its memory pipeline between groups
            typename Vec<T>::Raw p0[2], p1[2], p2[2], p3[2], p4[2], p5[2], p6[2], p7[2];
#define MGN_LOAD8(Q, R) load_row(R, ch0 + tx * V, Q##0); load_row(R + stride, ch0 + tx * V, Q##1); load_row(R + 2 * stride, ch0 + tx * V, Q##2); \
            load_row(R + 3 * stride, ch0 + tx * V, Q##3); load_row(R + 4 * stride, ch0 + tx * V, Q##4); load_row(R + 5 * stride, ch0 + tx * V, Q##5); \
            load_row(R + 6 * stride, ch0 + tx * V, Q##6); load_row(R + 7 * stride, ch0 + tx * V, Q##7)
#define MGN_ADD8(Q) add_row(Q##0, a, b); add_row(Q##1, a, b); add_row(Q##2, a, b); add_row(Q##3, a, b); add_row(Q##4, a, b); \
            add_row(Q##5, a, b); add_row(Q##6, a, b); add_row(Q##7, a, b)
            MGN_LOAD8(q, r);
            r += 8 * stride;
            for (;;) {
                if (!(r + 7 * stride < M)) { MGN_ADD8(q); break; }
                MGN_LOAD8(p, r);
                r += 8 * stride;
                MGN_ADD8(q);
                if (!(r + 7 * stride < M)) { MGN_ADD8(p); break; }
                MGN_LOAD8(q, r);
                r += 8 * stride;
                MGN_ADD8(p);
            }
#undef MGN_LOAD8
#undef MGN_ADD8
        }
        for (; r + 3 * stride < M; r += 4 * stride) {
            load_row(r, ch0 + tx * V, q0);
            load_row(r + stride, ch0 + tx * V, q1);
            load_row(r + 2 * stride, ch0 + tx * V, q2);
            load_row(r + 3 * stride, ch0 + tx * V, q3);
            add_row(q0, a, b);
            add_row(q1, a, b);
            add_row(q2, a, b);
            add_row(q3, a, b);
        }
        for (; r < M; r += stride) {
            load_row(r, ch0 + tx * V, q0);
            add_row(q0, a, b);
        }
    }
    // reduce over ty
#pragma unroll
    for (int k = 0; k < V; ++k) {
        sh[(ty * tpr + tx) * 2 * V + k] = a[k];
        sh[(ty * tpr + tx) * 2 * V + V + k] = b[k];
    }
    __syncthreads();
    float* partials = ws;
    // partials travel through device-coherent accesses (sc0 sc1: written through to / read from memory), so the
    // finalizing block on another XCD sees them without any L2 write-back / invalidate
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int COHERENT = 17;  // cache policy sc0 | sc1
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(partials, 0, (uint32_t)(sizeof(float) * 2 * (size_t)C * (MAX_STAT_BLOCKS + 1)), 0x00020000);
    if (ty == 0) {
        float sa[V], sb[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            sa[k] = sb[k] = 0.f;
            for (int y = 0; y < rpb; ++y) {
                sa[k] += sh[(y * tpr + tx) * 2 * V + k];
                sb[k] += sh[(y * tpr + tx) * 2 * V + V + k];
            }
        }
#pragma unroll
        for (int k = 0; k < V; k += 4) {
            const f32x4 va = {sa[k], sa[k + 1], sa[k + 2], sa[k + 3]}, vb = {sb[k], sb[k + 1], sb[k + 2], sb[k + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, va), rs,
                                                   (int)((((size_t)blockIdx.x * 2 + 0) * C + tx * V + k) * 4), 0, COHERENT);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, vb), rs,
                                                   (int)((((size_t)blockIdx.x * 2 + 1) * C + tx * V + k) * 4), 0, COHERENT);
        }
    }
    // ticket: the last block to arrive finalizes.  No agent-scope fences (a release would write back the whole L2, an
    // acquire invalidate it): the partials are device-coherent (write-through) stores, ordered before the ticket by
    // an EXPLICIT wait for their completion -- a workgroup-scope release fence does not emit s_waitcnt vmcnt(0) in
    // non-tgsplit mode, so without this the ticket could be taken while the stores are still in flight -- and the barrier.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (t == gridDim.x - 1);
    }
    __syncthreads();
    if (!is_last) return;
    const int nblk = gridDim.x;
    const int ncol4 = 2 * C / 4;                       // float4 columns of one partial row [2][C]
    const int cols = ncol4 < TPB ? ncol4 : TPB;
    const int slices = TPB / cols, sl = threadIdx.x / cols, cc = threadIdx.x % cols;
    float* fin = ws + (size_t)2 * C * MAX_STAT_BLOCKS;
    float4* sh4 = reinterpret_cast<float4*>(sh);
    for (int base = 0; base < ncol4; base += cols) {
        const int col = base + cc;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sl < slices && col < ncol4)
#pragma unroll 16   // these device-coherent loads come from memory (~2 us each): the tail is their round trips / loads in flight
            for (int k = sl; k < nblk; k += slices) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((size_t)k * ncol4 + col) * 16), 0, COHERENT));
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        __syncthreads();
        sh4[threadIdx.x] = acc;
        __syncthreads();
        if (sl == 0 && col < ncol4) {
            float4 t = sh4[cc];
            for (int k = 1; k < slices; ++k) {
                const float4 v = sh4[k * cols + cc];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            reinterpret_cast<float4*>(fin)[col] = t;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += TPB) finalize(ch0 + c, fin[c], fin[C + c]);
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
}

// forward statistics: sum (x - shift), sum (x - shift)^2 with shift = x[0, c]; the last block emits
//   stats[3][C] = {count, mean, M2} of THIS rank, and -- single-rank training (coef != null) -- directly the
//   scale/offset/mean/rstd of the apply + backward kernels and the running-statistics update (= iabn_combine with R=1)
struct StatsOut {
    float* stats;          // [3][C] or null
    float* coef;           // [4][C] = scale, offset, mean, rstd, or null
    const float* weight; const float* bias;
    float* running_mean; float* running_var;
    float eps, momentum;
};
MGN_PLAN_RO(StatsOut, MGN_RO(weight) MGN_RO(bias))   // pointers the kernels only read through (launch-plan dependency analysis, csrc/mgn_launch.h)

template <typename T>
__global__ __launch_bounds__(TPB) void iabn_stats_kernel(const T* __restrict__ x, long M, int C, int SC, float* ws, unsigned* counter, StatsOut o) {
    constexpr int V = Vec<T>::N;
    float s[V];
    Vec<T>::load(x + blockIdx.y * SC + (threadIdx.x % (SC / V)) * V, s);   // shift = first row of this thread's channels
    column_sums2<T, true>(M, SC, ws, counter,
        [&](long r, int c0, typename Vec<T>::Raw (&q)[2]) { q[0] = Vec<T>::load_raw(x + r * C + c0); },
        [&](const typename Vec<T>::Raw (&q)[2], float (&a)[V], float (&b)[V]) {
            float v[V];
            Vec<T>::unpack(q[0], v);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float d = v[k] - s[k];
                a[k] += d;
                b[k] += d * d;
            }
        },
        [&](int c, float s1, float s2) {
            float sv[V];
            Vec<T>::load(x + (c / V) * V, sv);
            const float shift = sv[c % V];
            const float n = (float)M;
            const float md = s1 / n;
            const float mean = shift + md, m2 = fmaxf(s2 - s1 * md, 0.f);  // sum (x-mean)^2
            if (o.stats) {
                o.stats[c] = n;
                o.stats[C + c] = mean;
                o.stats[2 * C + c] = m2;
            }
            if (o.coef) {
                const float var = m2 / n;
                const float rstd = rsqrtf(var + o.eps);
                const float g = fabsf(o.weight[c]) + o.eps;
                o.coef[c] = g * rstd;
                o.coef[C + c] = o.bias[c] - mean * g * rstd;
                o.coef[2 * C + c] = mean;
                o.coef[3 * C + c] = rstd;
                if (o.running_mean) {
                    o.running_mean[c] = (1.f - o.momentum) * o.running_mean[c] + o.momentum * mean;
                    o.running_var[c] = (1.f - o.momentum) * o.running_var[c] + o.momentum * var * (n / fmaxf(n - 1.f, 1.f));
                }
            }
        });
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

// The grid stride (gridDim.x * TPB vectors) is a multiple of C/V, so a thread always meets the same V channels: their
// coefficients are loaded once; two independent vectors are in flight per thread.
template <typename T>
__global__ __launch_bounds__(TPB) void iabn_apply(const T* __restrict__ x, T* __restrict__ y, long M, int C,
                                                  const float* __restrict__ scale, const float* __restrict__ offset,
                                                  int leaky, float slope) {
    constexpr int V = Vec<T>::N;
    const long nvec = M * C / V;
    const int cv = C / V;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    const int c0 = (int)(i0 % cv) * V;
    float sc[V], of[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { sc[k] = scale[c0 + k]; of[k] = offset[c0 + k]; }
    auto body = [&](const typename Vec<T>::Raw& q, long i) {
        float v[V];
        Vec<T>::unpack(q, v);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float z = fmaf(v[k], sc[k], of[k]);
            if (leaky) z = z > 0.f ? z : z * slope;
            v[k] = z;
        }
        Vec<T>::store(y + i * V, v);
    };
    long i = i0;
    for (; i + stride < nvec; i += 2 * stride) {
        const typename Vec<T>::Raw q0 = Vec<T>::load_raw(x + i * V), q1 = Vec<T>::load_raw(x + (i + stride) * V);
        body(q0, i);
        body(q1, i + stride);
    }
    if (i < nvec) body(Vec<T>::load_raw(x + i * V), i);
}

// backward pass 1: per-channel sum dz and sum dz * x_hat, with z = act^-1(y), dz = dy * act'(z), x_hat = (z - beta)/gamma';
// the last block writes sums[2][C] and the local parameter gradients
//   d bias = sum dz ; d weight = sign(weight) * sum dz*x_hat  (gamma' = |weight| + eps)
// dy where the 16-bit y is > 0, else 0, on the packed words (two values per dword): sign clear and magnitude non-zero
__device__ __forceinline__ uint32_t relu_mask2(uint32_t g, uint32_t yv) {
    const uint32_t t = yv & 0x7fff7fffu;
    const uint32_t nz = ((t + 0x7fff7fffu) | t) & 0x80008000u;
    const uint32_t pos = nz & ~yv;
    return g & (((pos >> 15) & 0x00010001u) * 0xffffu);
}

// two mask bits (bit 0: low, bit 1: high 16-bit value) -> the AND mask of the packed word
__device__ __forceinline__ uint32_t bits_mask2(uint32_t e) { return ((e & 1u) * 0xffffu) | (((e >> 1) & 1u) * 0xffff0000u); }

// MASK (block tail relu(norm(x) + shortcut), 16-bit only): `dy` is the gradient g of the ReLU's OUTPUT and `yrelu` that output; the
// gradient of both summands dm = g * (yrelu > 0) is formed here, stored to `dm_out` (the shortcut's gradient, and what the apply pass
// reads) and reduced in the same pass -- the separate mask pass (read g, y; write dm) and this pass's re-read of dm disappear.
// MASK = 2: the mask comes from `rbits`, one byte per 8 values (bit k = yrelu[k] > 0, written by eltwise.hip abn_add_relu_fwd<true>):
// 1/16 of the map is read instead of the map -- the same dm bit for bit.
template <typename T, bool FROMX = false, int MASK = 0>
__global__ __launch_bounds__(TPB) void iabn_bwd_reduce_kernel(const T* __restrict__ y, const T* __restrict__ dy, long M, int C,
                                                              const float* __restrict__ weight, const float* __restrict__ bias,
                                                              float eps, int leaky, float slope, int SC, float* ws, unsigned* counter,
                                                              float* sums, float* dwb, const float* __restrict__ psc = nullptr,
                                                              const float* __restrict__ pof = nullptr, const T* __restrict__ yrelu = nullptr,
                                                              T* __restrict__ dm_out = nullptr, const unsigned char* __restrict__ rbits = nullptr) {
    // FROMX: `y` holds the norm's INPUT x and z = psc * x + pof is recomputed instead of inverted from the activated output
    // (used where the normalised map is not kept: fused norm + add + ReLU of the residual blocks).  Compile-time: a run-time
    // test inside the streaming loop cost the ordinary path 30 %.
    constexpr int V = Vec<T>::N;
    const float inv_slope = FROMX ? 1.f : 1.f / slope;
    const int c0t = blockIdx.y * SC + (threadIdx.x % (SC / V)) * V;   // this thread's channels never change
    float bk[V], igk[V], sck[V], ofk[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        bk[k] = bias[c0t + k];
        igk[k] = 1.f / (fabsf(weight[c0t + k]) + eps);
        sck[k] = FROMX ? psc[c0t + k] : 1.f;
        ofk[k] = FROMX ? pof[c0t + k] : 0.f;
    }
    column_sums2<T, false>(M, SC, ws, counter,
        [&](long r, int c0, typename Vec<T>::Raw (&q)[2]) {
            q[0] = Vec<T>::load_raw(y + r * C + c0);
            q[1] = Vec<T>::load_raw(dy + r * C + c0);
            if constexpr (MASK == 1) {
                const uint4 yr = *reinterpret_cast<const uint4*>(yrelu + r * C + c0);
                uint4 m;
                m.x = relu_mask2(q[1].x, yr.x); m.y = relu_mask2(q[1].y, yr.y); m.z = relu_mask2(q[1].z, yr.z); m.w = relu_mask2(q[1].w, yr.w);
                *reinterpret_cast<uint4*>(dm_out + r * C + c0) = m;
                q[1] = m;
            }
            if constexpr (MASK == 2) {
                const uint32_t e = rbits[(r * C + c0) >> 3];
                uint4 m;
                m.x = q[1].x & bits_mask2(e); m.y = q[1].y & bits_mask2(e >> 2); m.z = q[1].z & bits_mask2(e >> 4); m.w = q[1].w & bits_mask2(e >> 6);
                *reinterpret_cast<uint4*>(dm_out + r * C + c0) = m;
                q[1] = m;
            }
        },
        [&](const typename Vec<T>::Raw (&q)[2], float (&a)[V], float (&b)[V]) {
            float yv[V], gv[V];
            Vec<T>::unpack(q[0], yv);
            Vec<T>::unpack(q[1], gv);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float z = FROMX ? fmaf(yv[k], sck[k], ofk[k]) : yv[k], dz = gv[k];
                // (sign BIT: a negative pre-activation whose 0.01-fold underflows the 16-bit format is stored as -0 and must still
                //  take the negative branch -- fp16: |z| < 3e-6; `z < 0.f` is false for -0)
                if (leaky && (__float_as_uint(z) >> 31)) { z *= inv_slope; dz *= slope; }
                a[k] += dz;
                b[k] += dz * ((z - bk[k]) * igk[k]);
            }
        },
        [&](int c, float s1, float s2) {
            sums[c] = s1;
            sums[C + c] = s2;
            if (dwb) {
                const float w = weight[c];
                dwb[c] = s2 * (float)((w > 0.f) - (w < 0.f));
                dwb[C + c] = s1;
            }
        });
}

// backward pass 2: dx = gamma' * rstd * (dz - sum_dz/n - x_hat * sum_dzxh/n)     (sums are GLOBAL over ranks, n too)
template <typename T, bool FROMX = false>
__global__ __launch_bounds__(TPB) void iabn_bwd_apply(const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ dx,
                                                      long M, int C, const float* __restrict__ weight,
                                                      const float* __restrict__ bias, const float* __restrict__ saved,
                                                      const float* __restrict__ sums, float inv_n, float eps, int leaky,
                                                      float slope, const float* __restrict__ psc = nullptr,
                                                      const float* __restrict__ pof = nullptr) {
    constexpr int V = Vec<T>::N;
    const long nvec = M * C / V;
    const int cv = C / V;
    const float inv_slope = FROMX ? 1.f : 1.f / slope;   // "from x" (see iabn_bwd_reduce_kernel): z comes from the affine map
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    const int c0 = (int)(i0 % cv) * V;   // fixed per thread (the stride is a multiple of C/V)
    // dx = A * (dz - m1) - (z - beta) * B   with A = gamma' * rstd, m1 = sum_dz / n, B = rstd * sum_dzxh / n
    float A[V], m1[V], Bc[V], bk[V], sck[V], ofk[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int c = c0 + k;
        const float g = fabsf(weight[c]) + eps, rstd = saved[C + c];
        A[k] = g * rstd;
        m1[k] = sums[c] * inv_n;
        Bc[k] = rstd * sums[C + c] * inv_n;
        bk[k] = bias[c];
        sck[k] = FROMX ? psc[c] : 1.f;
        ofk[k] = FROMX ? pof[c] : 0.f;
    }
    auto body = [&](const typename Vec<T>::Raw& qy, const typename Vec<T>::Raw& qg, long i) {
        float yv[V], gv[V];
        Vec<T>::unpack(qy, yv);
        Vec<T>::unpack(qg, gv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float z = FROMX ? fmaf(yv[k], sck[k], ofk[k]) : yv[k], dz = gv[k];
            if (leaky && (__float_as_uint(z) >> 31)) { z *= inv_slope; dz *= slope; }   // (sign bit: -0 is negative, see the reduction)
            gv[k] = A[k] * (dz - m1[k]) - (z - bk[k]) * Bc[k];
        }
        Vec<T>::store(dx + i * V, gv);
    };
    long i = i0;
    for (; i + 3 * stride < nvec; i += 4 * stride) {   // 8 independent 16-byte loads in flight per thread
        const typename Vec<T>::Raw y0 = Vec<T>::load_raw(y + i * V), g0 = Vec<T>::load_raw(dy + i * V);
        const typename Vec<T>::Raw y1 = Vec<T>::load_raw(y + (i + stride) * V), g1 = Vec<T>::load_raw(dy + (i + stride) * V);
        const typename Vec<T>::Raw y2 = Vec<T>::load_raw(y + (i + 2 * stride) * V), g2 = Vec<T>::load_raw(dy + (i + 2 * stride) * V);
        const typename Vec<T>::Raw y3 = Vec<T>::load_raw(y + (i + 3 * stride) * V), g3 = Vec<T>::load_raw(dy + (i + 3 * stride) * V);
        body(y0, g0, i);
        body(y1, g1, i + stride);
        body(y2, g2, i + 2 * stride);
        body(y3, g3, i + 3 * stride);
    }
    for (; i < nvec; i += stride) body(Vec<T>::load_raw(y + i * V), Vec<T>::load_raw(dy + i * V), i);
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

// channel slab of the reduction kernels: wide layers (few rows) get a block grid of (row blocks) x (C / 128)
inline int slab_channels(int C) { return (C > 128 && C % 128 == 0) ? 128 : C; }

inline int stat_blocks(long M, int SC, int dtype) {
    const int V = dtype == 1 ? 8 : 4;
    const int rpb = TPB / (SC / V);
    long b = (M + (long)rpb * 16 - 1) / ((long)rpb * 16);  // >= 16 rows (two 8-deep iterations) per thread before adding blocks
    return (int)(b < 1 ? 1 : (b > MAX_STAT_BLOCKS ? MAX_STAT_BLOCKS : b));
}

// Ticket counters of the single-launch reductions: a library-owned, zero-initialised pool per device; every launch
// takes the next slot (concurrent launches on different streams never share one) and the kernel leaves it at zero.
unsigned* next_counter() {
    static unsigned* pool[64] = {};
    static unsigned next[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!pool[dev]) {
        if (hipMalloc((void**)&pool[dev], 4096 * sizeof(unsigned)) != hipSuccess) return nullptr;
        if (hipMemset(pool[dev], 0, 4096 * sizeof(unsigned)) != hipSuccess) return nullptr;
    }
    const unsigned slot = (next[dev] & 255u) * 16u;   // 16 consecutive counters per launch (one per channel slab)
    ++next[dev];
    return pool[dev] + slot;
}


// ------------------------------------------------------------------------------------------------------
// statistics from the per-tile partial sums a convolution left behind (csrc/conv_win.hip: stat_part[rows][C][2] = sums of
// (r - shift), (r - shift)^2 of its rounded outputs): a block per 4 channels adds the rows in a fixed order (fp64) and writes
// the same outputs as iabn_stats_kernel's last block: stats[3][C] = {count, mean, M2} and / or the coefficient block + running
// statistics.  Replaces the statistics pass over the tensor (one read of the whole activation) by a read of rows*C*8 bytes.
// ------------------------------------------------------------------------------------------------------
#ifndef MGN_F16
// first stage for layers with tens of thousands of partial rows (the stems: 32768 pixel tiles): block (x, y) adds the rows y, y + Y, ...
// of 4 channels (fp64) and writes ONE row of a [Y][C][2] table, which iabn_from_partials_kernel then finishes
__global__ __launch_bounds__(256) void iabn_partials_reduce_kernel(const float* __restrict__ part, int rows, int C, float* __restrict__ out) {
    __shared__ double sh[32][8];
    const int c0 = blockIdx.x * 4, col = threadIdx.x & 7, rg = threadIdx.x >> 3;
    const float* src = part + (size_t)c0 * 2 + col;
    double a0 = 0.0, a1 = 0.0;
    const int step = 32 * gridDim.y;
    int r = blockIdx.y + rg * gridDim.y;
    for (; r + step < rows; r += 2 * step) {
        a0 += (double)src[(size_t)r * C * 2];
        a1 += (double)src[(size_t)(r + step) * C * 2];
    }
    if (r < rows) a0 += (double)src[(size_t)r * C * 2];
    sh[rg][col] = a0 + a1;
    __syncthreads();
    if (threadIdx.x < 8) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += sh[k][threadIdx.x];
        out[((size_t)blockIdx.y * C + c0) * 2 + threadIdx.x] = (float)t;
    }
}

__global__ __launch_bounds__(1024) void iabn_from_partials_kernel(const float* __restrict__ part, int rows, int C, long M, const float* __restrict__ shift, StatsOut o) {
    // a block per 4 channels (C / 4 blocks: 16 .. 128 of them): 8 floats = 32 bytes per partial row, blockDim / 8 row groups
    // (256 threads; 1024 for the layers with tens of thousands of pixel tiles: the stems)
    __shared__ double sh[128][8];
    const int c0 = blockIdx.x * 4, nrg = blockDim.x >> 3;
    const int col = threadIdx.x & 7, rg = threadIdx.x >> 3;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const float* src = part + (size_t)c0 * 2 + col;
    int r = rg;
    for (; r + 3 * nrg < rows; r += 4 * nrg) {
        a0 += (double)src[(size_t)r * C * 2];
        a1 += (double)src[(size_t)(r + nrg) * C * 2];
        a2 += (double)src[(size_t)(r + 2 * nrg) * C * 2];
        a3 += (double)src[(size_t)(r + 3 * nrg) * C * 2];
    }
    for (; r < rows; r += nrg) a0 += (double)src[(size_t)r * C * 2];
    sh[rg][col] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x < 4) {
        const int c = c0 + threadIdx.x;
        double t1 = 0.0, t2 = 0.0;
        for (int k = 0; k < nrg; ++k) {
            t1 += sh[k][2 * threadIdx.x];
            t2 += sh[k][2 * threadIdx.x + 1];
        }
        const double n = (double)M, md = t1 / n;
        const float mean = (float)((shift ? (double)shift[c] : 0.0) + md);
        const float m2 = (float)fmax(t2 - t1 * md, 0.0);
        const float nf = (float)M;
        if (o.stats) {
            o.stats[c] = nf;
            o.stats[C + c] = mean;
            o.stats[2 * C + c] = m2;
        }
        if (o.coef) {
            const float var = m2 / nf;
            const float rstd = rsqrtf(var + o.eps);
            const float g = fabsf(o.weight[c]) + o.eps;
            o.coef[c] = g * rstd;
            o.coef[C + c] = o.bias[c] - mean * g * rstd;
            o.coef[2 * C + c] = mean;
            o.coef[3 * C + c] = rstd;
            if (o.running_mean) {
                o.running_mean[c] = (1.f - o.momentum) * o.running_mean[c] + o.momentum * mean;
                o.running_var[c] = (1.f - o.momentum) * o.running_var[c] + o.momentum * var * (nf / fmaxf(nf - 1.f, 1.f));
            }
        }
    }
}
#endif
}  // namespace

extern "C" {

#ifndef MGN_F16
int mgn_iabn_workspace_bytes(long M, int C, int dtype, size_t* bytes) {
    int rc = check_shape(M, C, dtype);
    if (rc != MGN_OK) return rc;
    if (!bytes) return MGN_EINVAL;
    *bytes = sizeof(float) * 2 * (size_t)C * 1024;
    return MGN_OK;
}
#endif

int MGN_SYM(mgn_iabn_stats)(const void* x, int dtype, long M, int C, float* stats, void* ws, size_t ws_bytes, void* stream_) {
    int rc = check_shape(M, C, dtype);
    if (rc != MGN_OK) return rc;
    if (!x || !stats || !ws) return MGN_EINVAL;
    if (ws_bytes < sizeof(float) * 2 * (size_t)C * 1024) return MGN_ENOSPC;
    hipStream_t s = (hipStream_t)stream_;
    const int SC = slab_channels(C), nb = stat_blocks(M, SC, dtype);
    unsigned* ctr = next_counter();
    if (!ctr || C / SC > 16) return MGN_ELAUNCH;
    StatsOut o = {stats, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f};
    if (dtype == 1) hipLaunchKernelGGL(iabn_stats_kernel<__hip_bfloat16>, dim3(nb, C / SC), dim3(TPB), 0, s, (const __hip_bfloat16*)x, M, C, SC, (float*)ws, ctr, o);
    else hipLaunchKernelGGL(iabn_stats_kernel<float>, dim3(nb, C / SC), dim3(TPB), 0, s, (const float*)x, M, C, SC, (float*)ws, ctr, o);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int MGN_SYM(mgn_iabn_train_coeffs)(const void* x, int dtype, long M, int C, const float* weight, const float* bias, float eps, float momentum,
                          float* running_mean, float* running_var, float* coef, void* ws, size_t ws_bytes, void* stream_) {
    int rc = check_shape(M, C, dtype);
    if (rc != MGN_OK) return rc;
    if (!x || !weight || !bias || !coef || !ws) return MGN_EINVAL;
    if (ws_bytes < sizeof(float) * 2 * (size_t)C * 1024) return MGN_ENOSPC;
    hipStream_t s = (hipStream_t)stream_;
    const int SC = slab_channels(C), nb = stat_blocks(M, SC, dtype);
    unsigned* ctr = next_counter();
    if (!ctr || C / SC > 16) return MGN_ELAUNCH;
    StatsOut o = {nullptr, coef, weight, bias, running_mean, running_var, eps, momentum};
    if (dtype == 1) hipLaunchKernelGGL(iabn_stats_kernel<__hip_bfloat16>, dim3(nb, C / SC), dim3(TPB), 0, s, (const __hip_bfloat16*)x, M, C, SC, (float*)ws, ctr, o);
    else hipLaunchKernelGGL(iabn_stats_kernel<float>, dim3(nb, C / SC), dim3(TPB), 0, s, (const float*)x, M, C, SC, (float*)ws, ctr, o);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

#ifndef MGN_F16
int mgn_iabn_partials_reduce(const float* partials, int rows, int C, int rows_out, float* out, void* stream_) {
    if (!partials || !out || rows < 1 || rows_out < 1 || rows_out > 1024 || C < 4 || C % 4 != 0) return MGN_EINVAL;
    hipLaunchKernelGGL(iabn_partials_reduce_kernel, dim3(C / 4, rows_out), dim3(256), 0, (hipStream_t)stream_, partials, rows, C, out);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_iabn_coeffs_from_partials(const float* partials, int rows, int C, long M, const float* shift, const float* weight, const float* bias,
                                  float eps, float momentum, float* running_mean, float* running_var, float* coef, float* stats,
                                  void* stream_) {
    if (!partials || rows < 1 || C < 4 || C % 4 != 0 || M < 1 || (!coef && !stats)) return MGN_EINVAL;
    if (coef && (!weight || !bias)) return MGN_EINVAL;
    StatsOut o = {stats, coef, weight, bias, running_mean, running_var, eps, momentum};
    hipLaunchKernelGGL(iabn_from_partials_kernel, dim3(C / 4), dim3(rows > 2048 ? 1024 : 256), 0, (hipStream_t)stream_, partials, rows, C, M, shift, o);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
#endif

#ifndef MGN_F16
int mgn_iabn_combine(const float* gathered, int n_ranks, int C, const float* weight, const float* bias, float eps,
                     float momentum, float* running_mean, float* running_var, float* scale, float* offset, float* saved,
                     void* stream_) {
    if (!gathered || n_ranks < 1 || C < 1 || !weight || !bias || !scale || !offset || !saved) return MGN_EINVAL;
    hipLaunchKernelGGL(iabn_combine, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream_, gathered, n_ranks, C, weight, bias, eps,
                       momentum, running_mean, running_var, scale, offset, saved);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
#endif

#ifndef MGN_F16
int mgn_iabn_eval_coeffs(int C, const float* weight, const float* bias, const float* running_mean, const float* running_var,
                         float eps, float* scale, float* offset, void* stream_) {
    if (C < 1 || !weight || !bias || !running_mean || !running_var || !scale || !offset) return MGN_EINVAL;
    hipLaunchKernelGGL(iabn_eval_coeffs, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream_, C, weight, bias, running_mean,
                       running_var, eps, scale, offset);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
#endif

int MGN_SYM(mgn_iabn_apply)(const void* x, void* y, int dtype, long M, int C, const float* scale, const float* offset, int activation,
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

int MGN_SYM(mgn_iabn_bwd_reduce)(const void* y, const void* dy, int dtype, long M, int C, const float* weight, const float* bias,
                        float eps, int activation, float slope, float* sums, float* dwb, void* ws, size_t ws_bytes, void* stream_) {
    return MGN_SYM(mgn_iabn_bwd_reduce_x)(y, dy, dtype, M, C, weight, bias, nullptr, nullptr, eps, activation, slope, sums, dwb, ws, ws_bytes,
                                 stream_);
}

int MGN_SYM(mgn_iabn_bwd_reduce_x)(const void* y, const void* dy, int dtype, long M, int C, const float* weight, const float* bias,
                          const float* scale, const float* offset, float eps, int activation, float slope, float* sums, float* dwb,
                          void* ws, size_t ws_bytes, void* stream_) {
    int rc = check_shape(M, C, dtype);
    if (rc != MGN_OK) return rc;
    if (!y || !dy || !weight || !bias || !sums || !ws || ((scale == nullptr) != (offset == nullptr))) return MGN_EINVAL;
    if (ws_bytes < sizeof(float) * 2 * (size_t)C * 1024) return MGN_ENOSPC;
    hipStream_t s = (hipStream_t)stream_;
    const int SC = slab_channels(C), nb = stat_blocks(M, SC, dtype);
    unsigned* ctr = next_counter();
    if (!ctr || C / SC > 16) return MGN_ELAUNCH;
    const dim3 grid(nb, C / SC);
#define MGN_LAUNCH_REDUCE(T, CT, FX) hipLaunchKernelGGL((iabn_bwd_reduce_kernel<T, FX>), grid, dim3(TPB), 0, s, (const CT*)y, (const CT*)dy, M, \
        C, weight, bias, eps, activation, slope, SC, (float*)ws, ctr, sums, dwb, scale, offset)
    if (dtype == 1) { if (scale) MGN_LAUNCH_REDUCE(__hip_bfloat16, __hip_bfloat16, true); else MGN_LAUNCH_REDUCE(__hip_bfloat16, __hip_bfloat16, false); }
    else { if (scale) MGN_LAUNCH_REDUCE(float, float, true); else MGN_LAUNCH_REDUCE(float, float, false); }
#undef MGN_LAUNCH_REDUCE
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

/* block tail: the reduction of mgn_iabn_bwd_reduce_x (identity activation) with the ReLU mask folded in -- g: gradient of relu(norm(x) +
 * shortcut); the mask (output > 0) from `relu_bits` (one byte per 8 values, written by mgn_abn_add_relu_fwd) when given, else from
 * `yrelu`, that output; writes dm = g * mask (the gradient of both summands) and reduces it in the same pass */
int MGN_SYM(mgn_iabn_bwd_reduce_x_relu)(const void* x, const void* g, const void* yrelu, const void* relu_bits, void* dm, long M, int C,
                               const float* weight, const float* bias, const float* scale, const float* offset, float eps, float* sums, float* dwb,
                               void* ws, size_t ws_bytes, void* stream_) {
    int rc = check_shape(M, C, 1);
    if (rc != MGN_OK) return rc;
    if (!x || !g || (!yrelu && !relu_bits) || !dm || !weight || !bias || !scale || !offset || !sums || !ws) return MGN_EINVAL;
    if (ws_bytes < sizeof(float) * 2 * (size_t)C * 1024) return MGN_ENOSPC;
    const int SC = slab_channels(C), nb = stat_blocks(M, SC, 1);
    unsigned* ctr = next_counter();
    if (!ctr || C / SC > 16) return MGN_ELAUNCH;
    if (relu_bits)
        hipLaunchKernelGGL((iabn_bwd_reduce_kernel<__hip_bfloat16, true, 2>), dim3(nb, C / SC), dim3(TPB), 0, (hipStream_t)stream_,
                           (const __hip_bfloat16*)x, (const __hip_bfloat16*)g, M, C, weight, bias, eps, 0, 0.01f, SC, (float*)ws, ctr, sums, dwb,
                           scale, offset, (const __hip_bfloat16*)nullptr, (__hip_bfloat16*)dm, (const unsigned char*)relu_bits);
    else
        hipLaunchKernelGGL((iabn_bwd_reduce_kernel<__hip_bfloat16, true, 1>), dim3(nb, C / SC), dim3(TPB), 0, (hipStream_t)stream_,
                           (const __hip_bfloat16*)x, (const __hip_bfloat16*)g, M, C, weight, bias, eps, 0, 0.01f, SC, (float*)ws, ctr, sums, dwb,
                           scale, offset, (const __hip_bfloat16*)yrelu, (__hip_bfloat16*)dm, (const unsigned char*)nullptr);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int MGN_SYM(mgn_iabn_bwd_apply)(const void* y, const void* dy, void* dx, int dtype, long M, int C, const float* weight, const float* bias,
                       const float* saved, const float* sums, float total_count, float eps, int activation, float slope,
                       void* stream_) {
    return MGN_SYM(mgn_iabn_bwd_apply_x)(y, dy, dx, dtype, M, C, weight, bias, nullptr, nullptr, saved, sums, total_count, eps, activation, slope,
                                stream_);
}

int MGN_SYM(mgn_iabn_bwd_apply_x)(const void* y, const void* dy, void* dx, int dtype, long M, int C, const float* weight, const float* bias,
                         const float* scale, const float* offset, const float* saved, const float* sums, float total_count, float eps,
                         int activation, float slope, void* stream_) {
    int rc = check_shape(M, C, dtype);
    if (rc != MGN_OK) return rc;
    if (!y || !dy || !dx || !weight || !bias || !saved || !sums || !(total_count > 0.f)) return MGN_EINVAL;
    if ((scale == nullptr) != (offset == nullptr)) return MGN_EINVAL;
    hipStream_t s = (hipStream_t)stream_;
#define MGN_LAUNCH_APPLY(T, V, FX) hipLaunchKernelGGL((iabn_bwd_apply<T, FX>), dim3(grid_for(M * C / V)), dim3(TPB), 0, s, (const T*)y, \
        (const T*)dy, (T*)dx, M, C, weight, bias, saved, sums, 1.f / total_count, eps, activation, slope, scale, offset)
    if (dtype == 1) { if (scale) MGN_LAUNCH_APPLY(__hip_bfloat16, 8, true); else MGN_LAUNCH_APPLY(__hip_bfloat16, 8, false); }
    else { if (scale) MGN_LAUNCH_APPLY(float, 4, true); else MGN_LAUNCH_APPLY(float, 4, false); }
#undef MGN_LAUNCH_APPLY
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

}  // extern "C"

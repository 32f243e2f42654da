// eltwise.hip -- the element-wise / broadcast / pooling glue of the MGNet blocks on channels-last bf16 activations.
//
// Replaces the torch ops at (reference file:line)
//   res_net.py:77-78        out = out + shortcut ; F.relu_(out)                          -> add_relu (+ backward mask)
//   layers.py:170-184       FastGlobalAvgPool2d (mean over H*W)                          -> gap (two-stage column mean) + backward
//   layers.py:90, :217      F.interpolate(mode="nearest") (decoder x2, GCM 1x1 -> h x w) -> nearest upsampling + adjoint
//   layers.py:262-267       fm * sigmoid-attention  (AttentionRefinementModule)          -> scale_channels (mode 0)
//   layers.py:315-322       fm + fm * attention     (FeatureFusionModule)                -> scale_channels (mode 1)
//   layers.py:316           torch.cat([fsp, fcp], dim=1)                                 -> concat2 + split backward
// All kernels stream 16 bytes per lane; x is [N, H*W, C] with C % 8 == 0.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgnet_hip.h"
#include "h16.h"

namespace {

constexpr int TPB = 256;

__device__ __forceinline__ void unpack8(const uint4& r, float (&v)[8]) {   // 8 activations of this TU's 16-bit format (h16.h)
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[2 * k] = mgn_lo2f(w[k]);
        v[2 * k + 1] = mgn_hi2f(w[k]);
    }
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    uint4 r;
    r.x = mgn_pack2(v[0], v[1]); r.y = mgn_pack2(v[2], v[3]);
    r.z = mgn_pack2(v[4], v[5]); r.w = mgn_pack2(v[6], v[7]);
    return r;
}
inline unsigned blocks_for(long nvec) {
    long b = (nvec + TPB - 1) / TPB;
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// y = relu(a + b)
__global__ __launch_bounds__(TPB) void add_relu_fwd(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ y, long nvec) {
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
        float va[8], vb[8];
        unpack8(a[i], va);
        unpack8(b[i], vb);
#pragma unroll
        for (int k = 0; k < 8; ++k) va[k] = fmaxf(va[k] + vb[k], 0.f);
        y[i] = pack8(va);
    }
}
// sum of the gradients of a tensor consumed by three branches (the backbone features feed the three heads): one pass, one rounding
// (autograd's own accumulation is two passes with an intermediate 16-bit tensor); c may be null
__global__ __launch_bounds__(TPB) void sum3_h16(const uint4* __restrict__ a, const uint4* __restrict__ b, const uint4* __restrict__ c,
                                                uint4* __restrict__ y, long nvec) {
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
        float va[8], vb[8], vc[8];
        unpack8(a[i], va);
        unpack8(b[i], vb);
        if (c) {
            unpack8(c[i], vc);
#pragma unroll
            for (int k = 0; k < 8; ++k) va[k] = (va[k] + vb[k]) + vc[k];
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) va[k] += vb[k];
        }
        y[i] = pack8(va);
    }
}
// residual-block tail with the second norm folded in: y = relu(bf16(scale[c] * x + offset[c]) + b)  -- the value the separate
// in-place norm + add_relu_fwd produce, without writing the normalised map (x, the conv output, is kept for the backward).
// BITS: one byte per 8 outputs is written beside y, bit k = (y[k] > 0) taken from the ROUNDED 16-bit value -- the backward's ReLU mask
// (iabn.hip iabn_bwd_reduce_kernel<.., MASK = 2>) then reads 1/16 of the map instead of the map.
__device__ __forceinline__ uint32_t positive_bits2(uint32_t w) {   // bit 0 / bit 1: low / high 16-bit value is > 0 (sign clear, magnitude non-zero)
    const uint32_t t = w & 0x7fff7fffu;
    const uint32_t pos = (((t + 0x7fff7fffu) | t) & 0x80008000u) & ~w;
    return ((pos >> 15) & 1u) | ((pos >> 30) & 2u);
}
template <bool BITS>
__global__ __launch_bounds__(TPB) void abn_add_relu_fwd(const uint4* __restrict__ x, const float* __restrict__ scale,
                                                        const float* __restrict__ offset, const uint4* __restrict__ b, uint4* __restrict__ y,
                                                        long nvec, int cv, unsigned char* __restrict__ bits) {
    const long i0 = (long)blockIdx.x * TPB + threadIdx.x, stride = (long)gridDim.x * TPB;   // stride % cv == 0 (host)
    const int c0 = (int)(i0 % cv) * 8;
    float sc[8], of[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = scale[c0 + k]; of[k] = offset[c0 + k]; }
    for (long i = i0; i < nvec; i += stride) {
        float va[8], vb[8];
        unpack8(x[i], va);
        unpack8(b[i], vb);
#pragma unroll
        for (int k = 0; k < 8; ++k) va[k] = fmaf(va[k], sc[k], of[k]);
        unpack8(pack8(va), va);   // the bf16 rounding of the stored normalised value
#pragma unroll
        for (int k = 0; k < 8; ++k) va[k] = fmaxf(va[k] + vb[k], 0.f);
        const uint4 r = pack8(va);
        y[i] = r;
        if constexpr (BITS)
            bits[i] = (unsigned char)(positive_bits2(r.x) | (positive_bits2(r.y) << 2) | (positive_bits2(r.z) << 4) | (positive_bits2(r.w) << 6));
    }
}
// dx = dy where y > 0 (the same tensor is the gradient of both summands)
__global__ __launch_bounds__(TPB) void relu_mask_bwd(const uint4* __restrict__ dy, const uint4* __restrict__ y, uint4* __restrict__ dx, long nvec) {
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
        float g[8], v[8];
        unpack8(dy[i], g);
        unpack8(y[i], v);
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = v[k] > 0.f ? g[k] : 0.f;
        dx[i] = pack8(g);
    }
}

// ---- per-image column sums over HW of f(row): partial[n][chunk][C] then final ------------------------------------
// block = (C/8 column vectors) x (256/(C/8) row lanes); grid = (chunks, N)
template <bool PRODUCT>
__global__ __launch_bounds__(TPB) void colsum_partial(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2, long HW, int C,
                                                      float* partial) {
    __shared__ float sh[TPB * 8];
    const int cv = C / 8, rl = TPB / cv;
    const int tx = threadIdx.x % cv, ty = threadIdx.x / cv;
    const int n = blockIdx.y, chunks = gridDim.x;
    const long r0 = HW * blockIdx.x / chunks, r1 = HW * (blockIdx.x + 1) / chunks;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    if (ty < rl)
        for (long r = r0 + ty; r < r1; r += rl) {
            float v[8];
            unpack8(*reinterpret_cast<const uint4*>(x + ((long)n * HW + r) * C + tx * 8), v);
            if (PRODUCT) {
                float w[8];
                unpack8(*reinterpret_cast<const uint4*>(x2 + ((long)n * HW + r) * C + tx * 8), w);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += v[k] * w[k];
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += v[k];
            }
        }
#pragma unroll
    for (int k = 0; k < 8; ++k) sh[(ty * cv + tx) * 8 + k] = acc[k];
    __syncthreads();
    if (ty == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float s = 0.f;
            for (int y = 0; y < rl; ++y) s += sh[(y * cv + tx) * 8 + k];
            partial[((long)n * chunks + blockIdx.x) * C + tx * 8 + k] = s;
        }
    }
}
__global__ void colsum_final(const float* partial, int chunks, int C, int N, float scale, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i % C;
    // four independent chains: up to 64 dependent adds behind as many dependent load issues cost ~10 us per call (25 calls per step, each
    // at the head of an attention vector's chain); fixed order, so still deterministic
    const float* q = partial + (long)n * chunks * C + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
#pragma unroll 4
    for (; k + 3 < chunks; k += 4) {
        s0 += q[(long)k * C]; s1 += q[(long)(k + 1) * C]; s2 += q[(long)(k + 2) * C]; s3 += q[(long)(k + 3) * C];
    }
    for (; k < chunks; ++k) s0 += q[(long)k * C];
    out[i] = ((s0 + s1) + (s2 + s3)) * scale;
}

// dx[n, r, c] = g[n, c] * scale  (adjoint of the global average pool)
__global__ __launch_bounds__(TPB) void bcast_rows(const float* __restrict__ g, long HW, int C, float scale, uint4* __restrict__ dx, long nvec) {
    const int cv = C / 8;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
        const int c8 = (int)(i % cv);
        const long n = i / cv / HW;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = g[n * C + c8 * 8 + k] * scale;
        dx[i] = pack8(v);
    }
}

// y = x * s[n, c]            (mode 0)      y = x + x * s[n, c] = x * (1 + s)   (mode 1)
// (+ add[n, c], + addt[n, r, c]: a full tensor of x's shape -- the decoder's `arm(x) + last`, layers.py:87 -- fp32 sum, one rounding)
__global__ __launch_bounds__(TPB) void scale_channels(const uint4* __restrict__ x, const float* __restrict__ s, long HW, int C, int mode,
                                                      const float* __restrict__ add, const uint4* __restrict__ addt, uint4* __restrict__ y,
                                                      long nvec) {
    const int cv = C / 8;
    const float base = mode ? 1.f : 0.f;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
        const int c8 = (int)(i % cv);
        const long n = i / cv / HW;
        float v[8];
        unpack8(x[i], v);
        const float4* sp = reinterpret_cast<const float4*>(s + n * C + c8 * 8);
        const float4 s0 = sp[0], s1 = sp[1];
        const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        float av[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (add) {
            const float4* ap = reinterpret_cast<const float4*>(add + n * C + c8 * 8);
            const float4 a0 = ap[0], a1 = ap[1];
            av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
        }
        if (addt) {
            float tv[8];
            unpack8(addt[i], tv);
#pragma unroll
            for (int k = 0; k < 8; ++k) av[k] += tv[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fmaf(v[k], base + sv[k], av[k]);
        y[i] = pack8(v);
    }
}

// nearest-neighbour upsampling (torch: src = floor(dst * in / out)) and its adjoint (sum over the preimage)
__global__ __launch_bounds__(TPB) void nearest_fwd(const uint16_t* __restrict__ x, int N, int h, int w, int H, int W, int C, uint4* __restrict__ y) {
    const int cv = C / 8;
    const long nvec = (long)N * H * W * cv;
    const float sy = (float)h / H, sx = (float)w / W;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
        const int c8 = (int)(i % cv);
        long r = i / cv;
        const int X = (int)(r % W); r /= W;
        const int Y = (int)(r % H);
        const int n = (int)(r / H);
        const int ys = min((int)floorf(Y * sy), h - 1), xs = min((int)floorf(X * sx), w - 1);
        y[i] = *reinterpret_cast<const uint4*>(x + (((long)n * h + ys) * w + xs) * C + c8 * 8);
    }
}
__global__ __launch_bounds__(TPB) void nearest_bwd(const uint16_t* __restrict__ dy, int N, int h, int w, int H, int W, int C, uint4* __restrict__ dx) {
    const int cv = C / 8;
    const long nvec = (long)N * h * w * cv;
    const float sy = (float)h / H, sx = (float)w / W;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
        const int c8 = (int)(i % cv);
        long r = i / cv;
        const int xs = (int)(r % w); r /= w;
        const int ys = (int)(r % h);
        const int n = (int)(r / h);
        // destination rows/cols whose source is (ys, xs): a contiguous range around ys*H/h
        int Y0 = (int)((long)ys * H / h), X0 = (int)((long)xs * W / w);
        while (Y0 > 0 && min((int)floorf((Y0 - 1) * sy), h - 1) == ys) --Y0;
        while (X0 > 0 && min((int)floorf((X0 - 1) * sx), w - 1) == xs) --X0;
        while (Y0 < H && min((int)floorf(Y0 * sy), h - 1) < ys) ++Y0;
        while (X0 < W && min((int)floorf(X0 * sx), w - 1) < xs) ++X0;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int Y = Y0; Y < H && min((int)floorf(Y * sy), h - 1) == ys; ++Y)
            for (int X = X0; X < W && min((int)floorf(X * sx), w - 1) == xs; ++X) {
                float v[8];
                unpack8(*reinterpret_cast<const uint4*>(dy + (((long)n * H + Y) * W + X) * C + c8 * 8), v);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += v[k];
            }
        dx[i] = pack8(acc);
    }
}

// out[n, r, :] = [a[n, r, :Ca], b[n, r, :Cb]]   and the split of its gradient
__global__ __launch_bounds__(TPB) void concat2(const uint4* __restrict__ a, const uint4* __restrict__ b, int Ca, int Cb, uint4* __restrict__ y, long rows) {
    const int va = Ca / 8, vb = Cb / 8, vt = va + vb;
    const long nvec = rows * vt;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
        const long r = i / vt;
        const int c = (int)(i % vt);
        y[i] = c < va ? a[r * va + c] : b[r * vb + (c - va)];
    }
}
__global__ __launch_bounds__(TPB) void split2(const uint4* __restrict__ dy, int Ca, int Cb, uint4* __restrict__ da, uint4* __restrict__ db, long rows) {
    const int va = Ca / 8, vb = Cb / 8, vt = va + vb;
    const long nvec = rows * vt;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
        const long r = i / vt;
        const int c = (int)(i % vt);
        if (c < va) da[r * va + c] = dy[i];
        else db[r * vb + (c - va)] = dy[i];
    }
}


// ---- InPlaceABNSync followed by a channel-attention module (layers.py:221-322: ARM / FFM on the output of conv + norm), fused ------------
// The attention factor s[n, c] depends on the global average of the normalised map z, and the pooled branch's gradient dpool[n, c] on
// sum(g * z): two-pass problems like the norm itself.  Done as separate ops the chain costs, per module,
//   forward : apply (R y, W z) + pool (R z) + scale (R z, W out)                                  = 3 R + 2 W
//   backward: sum g*z (R g, R z) + scale (R g, W dx) + norm reduce (R dx, R z) + norm apply (R dx, R z, W dy) = 7 R + 2 W
// of activation-sized traffic.  dx = g * base + dpool (base = s or 1 + s) is linear in g with per-(image, channel) coefficients, so the
// norm's two sums follow from five per-(image, channel) sums taken in ONE pass over (g, z),
//   P = sum g z,  B1 = sum g m,  B2 = sum g m xh,  M1 = sum m,  M2 = sum m xh      (m = act'(z), xh = (act^-1(z) - beta) / gamma')
//   sum dz = sum_n base B1 + dpool M1,     sum dz xh = sum_n base B2 + dpool M2
// and the norm's input gradient dy from a second pass over (g, z): 4 R + 1 W; forward 2 R + 2 W with the pool taken while z is written.
template <int NQ>
__device__ __forceinline__ void block_colsums(float (&acc)[NQ][8], float* sh, int tx, int ty, int cv, int rl, float* dst, int C) {
    // sh: [rl][cv][8] floats per quantity, one quantity after the other (TPB * 8 floats)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) sh[(ty * cv + tx) * 8 + k] = acc[q][k];
        __syncthreads();
        if (ty == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float t = 0.f;
                for (int y = 0; y < rl; ++y) t += sh[(y * cv + tx) * 8 + k];
                dst[(long)q * C + tx * 8 + k] = t;
            }
        }
    }
}

// z = act(scale * x + offset) (x and z may be the same tensor) + partial[n][chunk][C] = sums of the ROUNDED z over the chunk's rows
__global__ __launch_bounds__(TPB) void abn_apply_pool(const uint16_t* x, uint16_t* z, const float* __restrict__ scale,
                                                      const float* __restrict__ offset, int leaky, float slope, long HW, int C, float* partial) {
    __shared__ float sh[TPB * 8];
    const int cv = C / 8, rl = TPB / cv;
    const int tx = threadIdx.x % cv, ty = threadIdx.x / cv;
    const int n = blockIdx.y, chunks = gridDim.x;
    const long r0 = HW * blockIdx.x / chunks, r1 = HW * (blockIdx.x + 1) / chunks;
    float sc[8], of[8], acc[1][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = scale[tx * 8 + k]; of[k] = offset[tx * 8 + k]; acc[0][k] = 0.f; }
    auto body = [&](const uint4& q, long r) {
        float v[8];
        unpack8(q, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t = fmaf(v[k], sc[k], of[k]);
            v[k] = leaky ? (t > 0.f ? t : t * slope) : t;   // (the expression of iabn_apply, csrc/iabn.hip)
        }
        const uint4 o = pack8(v);
        *reinterpret_cast<uint4*>(z + ((long)n * HW + r) * C + tx * 8) = o;
        unpack8(o, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[0][k] += v[k];
    };
    long r = r0 + ty;
    for (; r + 3 * rl < r1; r += 4 * rl) {
        const uint16_t* b = x + ((long)n * HW + r) * C + tx * 8;
        const uint4 q0 = *reinterpret_cast<const uint4*>(b), q1 = *reinterpret_cast<const uint4*>(b + (long)rl * C);
        const uint4 q2 = *reinterpret_cast<const uint4*>(b + 2L * rl * C), q3 = *reinterpret_cast<const uint4*>(b + 3L * rl * C);
        body(q0, r); body(q1, r + rl); body(q2, r + 2 * rl); body(q3, r + 3 * rl);
    }
    for (; r < r1; r += rl) body(*reinterpret_cast<const uint4*>(x + ((long)n * HW + r) * C + tx * 8), r);
    block_colsums<1>(acc, sh, tx, ty, cv, rl, partial + ((long)n * chunks + blockIdx.x) * C, C);
}

// the five per-(image, channel) sums of the backward (see above): partial[n][chunk][5][C]
__global__ __launch_bounds__(TPB) void att_abn_bwd_stats(const uint16_t* __restrict__ g, const uint16_t* __restrict__ z,
                                                         const float* __restrict__ weight, const float* __restrict__ bias, float eps, int leaky,
                                                         float slope, long HW, int C, float* partial) {
    __shared__ float sh[TPB * 8];
    const int cv = C / 8, rl = TPB / cv;
    const int tx = threadIdx.x % cv, ty = threadIdx.x / cv;
    const int n = blockIdx.y, chunks = gridDim.x;
    const long r0 = HW * blockIdx.x / chunks, r1 = HW * (blockIdx.x + 1) / chunks;
    const float inv_slope = 1.f / slope;
    float bk[8], igk[8], acc[5][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        bk[k] = bias[tx * 8 + k];
        igk[k] = 1.f / (fabsf(weight[tx * 8 + k]) + eps);
#pragma unroll
        for (int q = 0; q < 5; ++q) acc[q][k] = 0.f;
    }
    auto body = [&](const uint4& qg, const uint4& qz) {
        float gv[8], zv[8];
        unpack8(qg, gv);
        unpack8(qz, zv);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float zl = zv[k], m = 1.f;
            if (leaky && (__float_as_uint(zl) >> 31)) { zl *= inv_slope; m = slope; }   // (sign bit: -0 is a negative pre-activation, csrc/iabn.hip)
            const float xh = (zl - bk[k]) * igk[k], gm = gv[k] * m;
            acc[0][k] = fmaf(gv[k], zv[k], acc[0][k]);
            acc[1][k] += gm;
            acc[2][k] = fmaf(gm, xh, acc[2][k]);
            acc[3][k] += m;
            acc[4][k] = fmaf(m, xh, acc[4][k]);
        }
    };
    long r = r0 + ty;
    for (; r + rl < r1; r += 2 * rl) {
        const long o = ((long)n * HW + r) * C + tx * 8;
        const uint4 g0 = *reinterpret_cast<const uint4*>(g + o), z0 = *reinterpret_cast<const uint4*>(z + o);
        const uint4 g1 = *reinterpret_cast<const uint4*>(g + o + (long)rl * C), z1 = *reinterpret_cast<const uint4*>(z + o + (long)rl * C);
        body(g0, z0);
        body(g1, z1);
    }
    if (r < r1) {
        const long o = ((long)n * HW + r) * C + tx * 8;
        body(*reinterpret_cast<const uint4*>(g + o), *reinterpret_cast<const uint4*>(z + o));
    }
    block_colsums<5>(acc, sh, tx, ty, cv, rl, partial + ((long)n * chunks + blockIdx.x) * 5 * C, C);
}

// dy = A (dz - m1) - (zlin - beta) Bc   with dz = (g base[n, c] + dpool[n, c]) m      (iabn_bwd_apply of csrc/iabn.hip on the fused dz)
__global__ __launch_bounds__(TPB) void att_abn_bwd_apply(const uint16_t* __restrict__ g, const uint16_t* __restrict__ z, uint16_t* __restrict__ dy,
                                                         const float* __restrict__ s, const float* __restrict__ dpool, int mode,
                                                         const float* __restrict__ weight, const float* __restrict__ bias,
                                                         const float* __restrict__ rstd, const float* __restrict__ sums, float inv_n, float eps,
                                                         int leaky, float slope, long HW, int C) {
    const int cv = C / 8, rl = TPB / cv;
    const int tx = threadIdx.x % cv, ty = threadIdx.x / cv;
    const int n = blockIdx.y, chunks = gridDim.x;
    const long r0 = HW * blockIdx.x / chunks, r1 = HW * (blockIdx.x + 1) / chunks;
    const float inv_slope = 1.f / slope;
    float A[8], m1[8], Bc[8], bk[8], bs[8], dp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = tx * 8 + k;
        const float gm = fabsf(weight[c]) + eps, rs = rstd[c];
        A[k] = gm * rs;
        m1[k] = sums[c] * inv_n;
        Bc[k] = rs * sums[C + c] * inv_n;
        bk[k] = bias[c];
        bs[k] = (mode ? 1.f : 0.f) + s[(long)n * C + c];
        dp[k] = dpool ? dpool[(long)n * C + c] : 0.f;
    }
    auto body = [&](const uint4& qg, const uint4& qz, long r) {
        float gv[8], zv[8];
        unpack8(qg, gv);
        unpack8(qz, zv);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float zl = zv[k], dz = fmaf(gv[k], bs[k], dp[k]);
            if (leaky && (__float_as_uint(zl) >> 31)) { zl *= inv_slope; dz *= slope; }
            gv[k] = A[k] * (dz - m1[k]) - (zl - bk[k]) * Bc[k];
        }
        *reinterpret_cast<uint4*>(dy + ((long)n * HW + r) * C + tx * 8) = pack8(gv);
    };
    long r = r0 + ty;
    for (; r + 3 * rl < r1; r += 4 * rl) {
        const long o = ((long)n * HW + r) * C + tx * 8, st = (long)rl * C;
        const uint4 g0 = *reinterpret_cast<const uint4*>(g + o), z0 = *reinterpret_cast<const uint4*>(z + o);
        const uint4 g1 = *reinterpret_cast<const uint4*>(g + o + st), z1 = *reinterpret_cast<const uint4*>(z + o + st);
        const uint4 g2 = *reinterpret_cast<const uint4*>(g + o + 2 * st), z2 = *reinterpret_cast<const uint4*>(z + o + 2 * st);
        const uint4 g3 = *reinterpret_cast<const uint4*>(g + o + 3 * st), z3 = *reinterpret_cast<const uint4*>(z + o + 3 * st);
        body(g0, z0, r); body(g1, z1, r + rl); body(g2, z2, r + 2 * rl); body(g3, z3, r + 3 * rl);
    }
    for (; r < r1; r += rl) {
        const long o = ((long)n * HW + r) * C + tx * 8;
        body(*reinterpret_cast<const uint4*>(g + o), *reinterpret_cast<const uint4*>(z + o), r);
    }
}

// partial[n][chunk][5][C] -> S[5][N][C] (quantity-major: S[0] = sum g z is the [N][C] matrix the attention branch's backward consumes)
__global__ void att_abn_stats_final(const float* __restrict__ partial, int chunks, int C, int N, float* __restrict__ S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * 5 * C) return;
    const int n = i / (5 * C), qc = i % (5 * C), q = qc / C, c = qc % C;
    const float* p = partial + (long)n * chunks * 5 * C + qc;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < chunks; k += 4) {
        s0 += p[(long)k * 5 * C]; s1 += p[(long)(k + 1) * 5 * C]; s2 += p[(long)(k + 2) * 5 * C]; s3 += p[(long)(k + 3) * 5 * C];
    }
    for (; k < chunks; ++k) s0 += p[(long)k * 5 * C];
    S[((long)q * N + n) * C + c] = (s0 + s1) + (s2 + s3);
}
#ifndef MGN_F16

// the norm's two sums (and its parameter gradients) from the per-(image, channel) sums S[5][N][C], s and dpool: one thread per channel,
// images in a fixed order
__global__ void att_abn_bwd_sums(const float* __restrict__ S, const float* __restrict__ s, const float* __restrict__ dpool, int mode, int N, int C,
                                 const float* __restrict__ weight, float* __restrict__ sums, float* __restrict__ dwb) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, b = 0.f;
    for (int n = 0; n < N; ++n) {
        const float* q = S + (long)n * C + c;
        const long Q = (long)N * C;
        const float base = (mode ? 1.f : 0.f) + s[(long)n * C + c], dp = dpool ? dpool[(long)n * C + c] : 0.f;
        a += fmaf(base, q[Q], dp * q[3 * Q]);
        b += fmaf(base, q[2 * Q], dp * q[4 * Q]);
    }
    sums[c] = a;
    sums[C + c] = b;
    if (dwb) {
        const float w = weight[c];
        dwb[c] = b * (float)((w > 0.f) - (w < 0.f));
        dwb[C + c] = a;
    }
}
#endif

inline bool c_ok(int C) { return C >= 8 && C % 8 == 0 && C / 8 <= TPB && TPB % (C / 8) == 0; }
inline int chunks_for(long HW) { long c = HW / 512; return (int)(c < 1 ? 1 : (c > 64 ? 64 : c)); }
// passes that also WRITE the tensor want the whole chip: up to 256 chunks per image (>= 128 rows each)
inline int chunks_wide(long HW) { long c = HW / 128; return (int)(c < 1 ? 1 : (c > 256 ? 256 : c)); }

}  // namespace

extern "C" {

int MGN_SYM(mgn_add_relu_fwd)(const void* a, const void* b, void* y, long n_elems, void* stream) {
    if (!a || !b || !y || n_elems < 8 || n_elems % 8) return MGN_EINVAL;
    hipLaunchKernelGGL(add_relu_fwd, dim3(blocks_for(n_elems / 8)), dim3(TPB), 0, (hipStream_t)stream, (const uint4*)a, (const uint4*)b, (uint4*)y, n_elems / 8);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
int MGN_SYM(mgn_sum3)(const void* a, const void* b, const void* c, void* y, long n_elems, void* stream) {
    if (!a || !b || !y || n_elems < 8 || n_elems % 8) return MGN_EINVAL;
    hipLaunchKernelGGL(sum3_h16, dim3(blocks_for(n_elems / 8)), dim3(TPB), 0, (hipStream_t)stream, (const uint4*)a, (const uint4*)b, (const uint4*)c,
                       (uint4*)y, n_elems / 8);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
int MGN_SYM(mgn_abn_add_relu_fwd)(const void* x, const float* scale, const float* offset, const void* b, void* y, void* relu_bits, long M, int C,
                                  void* stream) {
    if (!x || !scale || !offset || !b || !y || M < 1 || !c_ok(C)) return MGN_EINVAL;
    const long nvec = M * C / 8;
    const int cv = C / 8;
    // c_ok guarantees TPB % cv == 0: with any block count the grid stride stays a multiple of cv (fixed channels per thread)
    if (relu_bits)
        hipLaunchKernelGGL(abn_add_relu_fwd<true>, dim3(blocks_for(nvec)), dim3(TPB), 0, (hipStream_t)stream, (const uint4*)x, scale, offset,
                           (const uint4*)b, (uint4*)y, nvec, cv, (unsigned char*)relu_bits);
    else
        hipLaunchKernelGGL(abn_add_relu_fwd<false>, dim3(blocks_for(nvec)), dim3(TPB), 0, (hipStream_t)stream, (const uint4*)x, scale, offset,
                           (const uint4*)b, (uint4*)y, nvec, cv, (unsigned char*)nullptr);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
int MGN_SYM(mgn_relu_mask_bwd)(const void* dy, const void* y, void* dx, long n_elems, void* stream) {
    if (!dy || !y || !dx || n_elems < 8 || n_elems % 8) return MGN_EINVAL;
    hipLaunchKernelGGL(relu_mask_bwd, dim3(blocks_for(n_elems / 8)), dim3(TPB), 0, (hipStream_t)stream, (const uint4*)dy, (const uint4*)y, (uint4*)dx, n_elems / 8);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

/* out[n, c] = scale * sum_r x[n, r, c] * (x2 ? x2[n, r, c] : 1);  workspace: N * chunks * C floats (chunks <= 64) */
int MGN_SYM(mgn_colsum)(const void* x, const void* x2, int N, long HW, int C, float scale, float* out, float* workspace, size_t workspace_bytes,
               void* stream) {
    if (!x || !out || !workspace || N < 1 || HW < 1 || !c_ok(C)) return MGN_EINVAL;
    const int chunks = chunks_for(HW);
    if (workspace_bytes < sizeof(float) * (size_t)N * chunks * C) return MGN_ENOSPC;
    hipStream_t s = (hipStream_t)stream;
    if (x2) hipLaunchKernelGGL(colsum_partial<true>, dim3(chunks, N), dim3(TPB), 0, s, (const uint16_t*)x, (const uint16_t*)x2, HW, C, workspace);
    else hipLaunchKernelGGL(colsum_partial<false>, dim3(chunks, N), dim3(TPB), 0, s, (const uint16_t*)x, (const uint16_t*)nullptr, HW, C, workspace);
    hipLaunchKernelGGL(colsum_final, dim3((N * C + 255) / 256), dim3(256), 0, s, (const float*)workspace, chunks, C, N, scale, out);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
int MGN_SYM(mgn_bcast_rows)(const float* g, int N, long HW, int C, float scale, void* dx, void* stream) {
    if (!g || !dx || N < 1 || HW < 1 || !c_ok(C)) return MGN_EINVAL;
    const long nvec = (long)N * HW * (C / 8);
    hipLaunchKernelGGL(bcast_rows, dim3(blocks_for(nvec)), dim3(TPB), 0, (hipStream_t)stream, g, HW, C, scale, (uint4*)dx, nvec);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
int MGN_SYM(mgn_scale_channels)(const void* x, const float* s, int N, long HW, int C, int mode, const float* add, const void* addt, void* y,
                                void* stream) {
    if (!x || !s || !y || N < 1 || HW < 1 || !c_ok(C) || mode < 0 || mode > 1) return MGN_EINVAL;
    const long nvec = (long)N * HW * (C / 8);
    hipLaunchKernelGGL(scale_channels, dim3(blocks_for(nvec)), dim3(TPB), 0, (hipStream_t)stream, (const uint4*)x, s, HW, C, mode, add, (const uint4*)addt, (uint4*)y, nvec);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
int MGN_SYM(mgn_nearest_fwd)(const void* x, int N, int h, int w, int H, int W, int C, void* y, void* stream) {
    if (!x || !y || N < 1 || h < 1 || w < 1 || H < h || W < w || !c_ok(C)) return MGN_EINVAL;
    hipLaunchKernelGGL(nearest_fwd, dim3(blocks_for((long)N * H * W * (C / 8))), dim3(TPB), 0, (hipStream_t)stream, (const uint16_t*)x, N, h, w, H, W, C, (uint4*)y);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
int MGN_SYM(mgn_nearest_bwd)(const void* dy, int N, int h, int w, int H, int W, int C, void* dx, void* stream) {
    if (!dy || !dx || N < 1 || h < 1 || w < 1 || H < h || W < w || !c_ok(C)) return MGN_EINVAL;
    hipLaunchKernelGGL(nearest_bwd, dim3(blocks_for((long)N * h * w * (C / 8))), dim3(TPB), 0, (hipStream_t)stream, (const uint16_t*)dy, N, h, w, H, W, C, (uint4*)dx);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
#ifndef MGN_F16
int mgn_concat2(const void* a, const void* b, long rows, int Ca, int Cb, void* y, void* stream) {
    if (!a || !b || !y || rows < 1 || Ca < 8 || Cb < 8 || Ca % 8 || Cb % 8) return MGN_EINVAL;
    hipLaunchKernelGGL(concat2, dim3(blocks_for(rows * ((Ca + Cb) / 8))), dim3(TPB), 0, (hipStream_t)stream, (const uint4*)a, (const uint4*)b, Ca, Cb, (uint4*)y, rows);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
#endif
#ifndef MGN_F16
int mgn_split2(const void* dy, long rows, int Ca, int Cb, void* da, void* db, void* stream) {
    if (!dy || !da || !db || rows < 1 || Ca < 8 || Cb < 8 || Ca % 8 || Cb % 8) return MGN_EINVAL;
    hipLaunchKernelGGL(split2, dim3(blocks_for(rows * ((Ca + Cb) / 8))), dim3(TPB), 0, (hipStream_t)stream, (const uint4*)dy, Ca, Cb, (uint4*)da, (uint4*)db, rows);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
#endif

/* InPlaceABNSync + channel attention, fused passes (see the kernels' header comment).
 * mgn_abn_apply_pool: z = act(scale[c] * x + offset[c]) (x == z allowed: in place) and pooled[n][c] = pool_scale * sum over the image's rows
 *   of the rounded z.  workspace: N * 256 * C floats.
 * mgn_att_abn_bwd_stats: S[5][N][C] = per-(image, channel) sums {g z, g m, g m xh, m, m xh} over (g, z) (m = act'(z), xh from z, weight, bias).
 *   workspace: N * 64 * 5 * C floats.
 * mgn_att_abn_bwd_sums (fp32 only): sums[2][C] = {sum dz, sum dz xh} of the norm's backward (+ dwb[2][C] = {d weight, d bias}) from S, the
 *   attention factor s[N][C], the pooled branch's gradient dpool[N][C] (1 / HW included; NULL = 0) and mode (0: z s, 1: z (1 + s)).
 * mgn_att_abn_bwd_apply: dy = the norm's input gradient for dz = (g base + dpool) m; sums are GLOBAL over the ranks, inv_n = 1 / (total rows). */
int MGN_SYM(mgn_abn_apply_pool)(const void* x, void* z, const float* scale, const float* offset, int act, float slope, int N, long HW, int C,
                                float pool_scale, float* pooled, float* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !z || !scale || !offset || !pooled || !workspace || N < 1 || HW < 1 || !c_ok(C) || act < 0 || act > 1) return MGN_EINVAL;
    const int chunks = chunks_wide(HW);
    if (workspace_bytes < sizeof(float) * (size_t)N * chunks * C) return MGN_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(abn_apply_pool, dim3(chunks, N), dim3(TPB), 0, st, (const uint16_t*)x, (uint16_t*)z, scale, offset, act, slope, HW, C, workspace);
    hipLaunchKernelGGL(colsum_final, dim3((N * C + 255) / 256), dim3(256), 0, st, (const float*)workspace, chunks, C, N, pool_scale, pooled);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
int MGN_SYM(mgn_att_abn_bwd_stats)(const void* g, const void* z, const float* weight, const float* bias, float eps, int act, float slope, int N,
                                   long HW, int C, float* S, float* workspace, size_t workspace_bytes, void* stream) {
    if (!g || !z || !weight || !bias || !S || !workspace || N < 1 || HW < 1 || !c_ok(C) || act < 0 || act > 1) return MGN_EINVAL;
    const int chunks = chunks_for(HW);
    if (workspace_bytes < sizeof(float) * (size_t)N * chunks * 5 * C) return MGN_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(att_abn_bwd_stats, dim3(chunks, N), dim3(TPB), 0, st, (const uint16_t*)g, (const uint16_t*)z, weight, bias, eps, act, slope, HW, C,
                       workspace);
    hipLaunchKernelGGL(att_abn_stats_final, dim3((N * 5 * C + 255) / 256), dim3(256), 0, st, (const float*)workspace, chunks, C, N, S);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
int MGN_SYM(mgn_att_abn_bwd_apply)(const void* g, const void* z, void* dy, const float* s, const float* dpool, int mode, const float* weight,
                                   const float* bias, const float* rstd, const float* sums, float inv_n, float eps, int act, float slope, int N,
                                   long HW, int C, void* stream) {
    if (!g || !z || !dy || !s || !weight || !bias || !rstd || !sums || N < 1 || HW < 1 || !c_ok(C) || mode < 0 || mode > 1 || act < 0 || act > 1)
        return MGN_EINVAL;
    hipLaunchKernelGGL(att_abn_bwd_apply, dim3(chunks_wide(HW), N), dim3(TPB), 0, (hipStream_t)stream, (const uint16_t*)g, (const uint16_t*)z,
                       (uint16_t*)dy, s, dpool, mode, weight, bias, rstd, sums, inv_n, eps, act, slope, HW, C);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
#ifndef MGN_F16
int mgn_att_abn_bwd_sums(const float* S, const float* s, const float* dpool, int mode, int N, int C, const float* weight, float* sums, float* dwb,
                         void* stream) {
    if (!S || !s || !weight || !sums || N < 1 || C < 1 || mode < 0 || mode > 1) return MGN_EINVAL;
    hipLaunchKernelGGL(att_abn_bwd_sums, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, S, s, dpool, mode, N, C, weight, sums, dwb);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
#endif

}  // extern "C"

// pool.hip -- 3x3 stride-2 pad-1 max pooling on channels-last bf16 activations (ResNet stem, res_net.py:109:
// F.max_pool2d(x, kernel_size=3, stride=2, padding=1)), forward with a 1-byte arg-max tap per element and a gather-style
// backward (each input pixel collects from the <= 4 windows that contain it: no atomics, deterministic).
//
// abn_maxpool_*: the same two kernels with the stem's activated batch norm folded in (BasicStem, res_net.py:82-110:
// conv -> InPlaceABNSync(leaky) -> max_pool).  The 537 MB stem activation (8 x 64 x 512 x 1024 bf16) is then read twice in
// the forward pass (statistics, pooling) and once in the backward pass instead of being rewritten in place, re-read by the
// pooling, re-read twice and rewritten by the norm's backward and written once more by the pooling's backward:
//   forward : y = leaky(scale * x + offset) is evaluated on the fly inside the pooling window (rounded to bf16 exactly as
//             the separate in-place pass would have stored it) -- the normalised map is never written;
//   backward: the channel sums of the norm's backward run over the POOLED tensors (d pooled is the only non-zero part of
//             d y, and pooled = y at the arg-max, so x_hat there follows from it: mgn_iabn_bwd_reduce on (pooled, d
//             pooled)); the second kernel gathers d y per 2x2 input patch like maxpool_bwd and applies
//             dx = A (dz - sum_dz / n) - (z - beta) B with z = scale * x + offset recomputed from the saved conv output.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mgnet_hip.h"
#include "h16.h"

namespace {

__device__ __forceinline__ float bf2f(uint16_t v) { return mgn_h2f(v); }   // this TU's 16-bit format (h16.h)

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
    for (int k = 0; k < 8; ++k) { best[k] = -3.4e38f; bits[k] = MGN_H16_LOWEST; arg[k] = 0; }
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

// one thread: 8 channels of a 2x2 INPUT patch (rows 2i, 2i+1; columns 2j, 2j+1).  The patch is covered by the four windows
// (i | i+1, j | j+1): pixel (2i+a, 2j+b) lies in window oh = i at tap row a+1 and, when a = 1, in window i+1 at tap row 0
// (columns alike), so 4 (dy, argmax) loads serve 4 pixels instead of 4 loads per pixel.
__global__ __launch_bounds__(256) void maxpool_bwd(const uint16_t* __restrict__ dy, const uint8_t* __restrict__ idx, uint16_t* __restrict__ dx,
                                                   int N, int IH, int IW, int C, int OH, int OW) {
    const int cv = C / 8, PH = (IH + 1) / 2, PW = (IW + 1) / 2;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)N * PH * PW * cv) return;
    const int c8 = (int)(t % cv);
    long r = t / cv;
    const int j = (int)(r % PW); r /= PW;
    const int i = (int)(r % PH);
    const int n = (int)(r / PH);
    float acc[2][2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[a][b][k] = 0.f;
#pragma unroll
    for (int wo = 0; wo < 2; ++wo) {
        const int oh = i + wo;
        if (oh >= OH) continue;
#pragma unroll
        for (int wx = 0; wx < 2; ++wx) {
            const int ow = j + wx;
            if (ow >= OW) continue;
            const long op = (((long)n * OH + oh) * OW + ow) * C + c8 * 8;
            const uint4 g = *reinterpret_cast<const uint4*>(dy + op);
            const uint2 am = *reinterpret_cast<const uint2*>(idx + op);
            const uint32_t gw[4] = {g.x, g.y, g.z, g.w};
            const uint32_t aw[2] = {am.x, am.y};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int tap = (int)((aw[k >> 2] >> ((k & 3) * 8)) & 0xff);
                const float v = bf2f((uint16_t)(gw[k >> 1] >> ((k & 1) * 16)));
                const int kh = tap / 3, kw = tap - kh * 3;
                // window (oh, ow) tap (kh, kw) is input pixel (2*oh - 1 + kh, 2*ow - 1 + kw) = patch offset (a, b):
                const int a = 2 * wo - 1 + kh, b = 2 * wx - 1 + kw;   // relative to (2i, 2j)
#pragma unroll
                for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb)
                        if (a == aa && b == bb) acc[aa][bb][k] += v;
            }
        }
    }
    auto f2bf = [](float f) -> uint32_t { return mgn_f2h(f); };
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int ih = 2 * i + a;
        if (ih >= IH) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int iw = 2 * j + b;
            if (iw >= IW) continue;
            uint4 o;
            o.x = f2bf(acc[a][b][0]) | (f2bf(acc[a][b][1]) << 16); o.y = f2bf(acc[a][b][2]) | (f2bf(acc[a][b][3]) << 16);
            o.z = f2bf(acc[a][b][4]) | (f2bf(acc[a][b][5]) << 16); o.w = f2bf(acc[a][b][6]) | (f2bf(acc[a][b][7]) << 16);
            *reinterpret_cast<uint4*>(dx + (((long)n * IH + ih) * IW + iw) * C + c8 * 8) = o;
        }
    }
}

// ---- fused with the activated batch norm of the stem --------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2bf_rne(float f) { return mgn_f2h(f); }

// A thread owns one 8-channel vector of TWO adjacent output pixels: their windows share the middle input column, so 15 instead of
// 18 vectors are loaded per pair -- all of them up front (282 -> 270 us at 8 x 512 x 1024 x 64; an XCD-banded block order on
// top of it measured neutral, so the re-read of the shared input row is not what bounds it) -- and the activation is evaluated once per loaded value.  Scan order and
// the strict comparison are those of the separate kernels (first maximum in (kh, kw) order wins).
__global__ __launch_bounds__(256) void abn_maxpool_fwd(const uint16_t* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ offset, int leaky, float slope,
                                                       uint16_t* __restrict__ y, uint8_t* __restrict__ idx, int N, int IH, int IW, int C,
                                                       int OH, int OW) {
    const int cv = C / 8, OWP = (OW + 1) / 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * OH * OWP * cv) return;
    const int c8 = (int)(i % cv);
    long r = i / cv;
    const int ow0 = (int)(r % OWP) * 2; r /= OWP;
    const int oh = (int)(r % OH);
    const int n = (int)(r / OH);
    uint4 v[3][5];
    bool ok[3][5];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh * 2 - 1 + kh;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int iw = ow0 * 2 - 1 + j;
            ok[kh][j] = ih >= 0 && ih < IH && iw >= 0 && iw < IW;
            v[kh][j] = ok[kh][j] ? *reinterpret_cast<const uint4*>(x + (((long)n * IH + ih) * IW + iw) * C + c8 * 8) : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    float sc[8], of[8], best[2][8];
    uint8_t arg[2][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = scale[c8 * 8 + k]; of[k] = offset[c8 * 8 + k];
        best[0][k] = best[1][k] = -3.4e38f; arg[0][k] = arg[1][k] = 0;
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (!ok[kh][j]) continue;
            const uint32_t w[4] = {v[kh][j].x, v[kh][j].y, v[kh][j].z, v[kh][j].w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float z = fmaf(bf2f((uint16_t)(w[k >> 1] >> ((k & 1) * 16))), sc[k], of[k]);   // iabn_apply's arithmetic
                if (leaky) z = z > 0.f ? z : z * slope;
                const float f = bf2f((uint16_t)f2bf_rne(z));
                // column j is tap kw = j of the left pixel (j <= 2) and tap kw = j - 2 of the right one (j >= 2)
                if (j <= 2 && f > best[0][k]) { best[0][k] = f; arg[0][k] = (uint8_t)(kh * 3 + j); }
                if (j >= 2 && f > best[1][k]) { best[1][k] = f; arg[1][k] = (uint8_t)(kh * 3 + j - 2); }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (ow0 + u >= OW) break;
        uint4 o;
        o.x = mgn_pack2(best[u][0], best[u][1]); o.y = mgn_pack2(best[u][2], best[u][3]);   // (exact: the maxima are 16-bit values)
        o.z = mgn_pack2(best[u][4], best[u][5]); o.w = mgn_pack2(best[u][6], best[u][7]);
        const long op = (((long)n * OH + oh) * OW + ow0 + u) * C + c8 * 8;
        *reinterpret_cast<uint4*>(y + op) = o;
        uint2 a;
        a.x = arg[u][0] | (arg[u][1] << 8) | (arg[u][2] << 16) | ((uint32_t)arg[u][3] << 24);
        a.y = arg[u][4] | (arg[u][5] << 8) | (arg[u][6] << 16) | ((uint32_t)arg[u][7] << 24);
        *reinterpret_cast<uint2*>(idx + op) = a;
    }
}

// The same, marching down POOL_ROWS output rows per thread: of the three input rows of a window the top one (2 oh - 1) was the bottom one of
// the previous output row, so its activated values are kept in registers and each input row is loaded and activated ONCE per column pair
// instead of 1.5 times (round 4).  Same arithmetic per value and the same comparison order (kh outer, column inner): bit-identical outputs
// and arg-max bytes.
constexpr int POOL_ROWS = 8;
__global__ __launch_bounds__(256) void abn_maxpool_fwd_march(const uint16_t* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ offset, int leaky, float slope,
                                                             uint16_t* __restrict__ y, uint8_t* __restrict__ idx, int N, int IH, int IW, int C,
                                                             int OH, int OW) {
    const int cv = C / 8, OWP = (OW + 1) / 2, OHC = (OH + POOL_ROWS - 1) / POOL_ROWS;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * OHC * OWP * cv) return;
    const int c8 = (int)(i % cv);
    long r = i / cv;
    const int ow0 = (int)(r % OWP) * 2; r /= OWP;
    const int oh0 = (int)(r % OHC) * POOL_ROWS;
    const int n = (int)(r / OHC);
    float sc[8], of[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = scale[c8 * 8 + k]; of[k] = offset[c8 * 8 + k]; }
    bool okc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) { const int iw = ow0 * 2 - 1 + j; okc[j] = iw >= 0 && iw < IW; }
    auto load_row = [&](int ih, uint4 (&v)[5]) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int iw = ow0 * 2 - 1 + j;
            v[j] = (okc[j] && ih >= 0 && ih < IH) ? *reinterpret_cast<const uint4*>(x + (((long)n * IH + ih) * IW + iw) * C + c8 * 8)
                                                  : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto activate = [&](const uint4 (&v)[5], float (&f)[5][8]) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float z = fmaf(bf2f((uint16_t)(w[k >> 1] >> ((k & 1) * 16))), sc[k], of[k]);   // iabn_apply's arithmetic
                if (leaky) z = z > 0.f ? z : z * slope;
                f[j][k] = bf2f((uint16_t)f2bf_rne(z));
            }
        }
    };
    float top[5][8];          // activated row 2 oh - 1 (kh = 0)
    bool top_ok;
    {
        uint4 v[5];
        load_row(2 * oh0 - 1, v);
        activate(v, top);
        top_ok = 2 * oh0 - 1 >= 0;
    }
    for (int oh = oh0; oh < oh0 + POOL_ROWS && oh < OH; ++oh) {
        uint4 v1[5], v2[5];
        load_row(2 * oh, v1);
        load_row(2 * oh + 1, v2);
        float f1[5][8], f2[5][8];
        activate(v1, f1);
        activate(v2, f2);
        const bool ok1 = 2 * oh < IH, ok2 = 2 * oh + 1 < IH;
        float best[2][8];
        uint8_t arg[2][8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { best[0][k] = best[1][k] = -3.4e38f; arg[0][k] = arg[1][k] = 0; }
        auto scan = [&](const float (&f)[5][8], bool rowok, int kh) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                if (!(rowok && okc[j])) continue;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    // column j is tap kw = j of the left pixel (j <= 2) and tap kw = j - 2 of the right one (j >= 2)
                    if (j <= 2 && f[j][k] > best[0][k]) { best[0][k] = f[j][k]; arg[0][k] = (uint8_t)(kh * 3 + j); }
                    if (j >= 2 && f[j][k] > best[1][k]) { best[1][k] = f[j][k]; arg[1][k] = (uint8_t)(kh * 3 + j - 2); }
                }
            }
        };
        scan(top, top_ok, 0);
        scan(f1, ok1, 1);
        scan(f2, ok2, 2);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (ow0 + u >= OW) break;
            uint4 o;
            o.x = mgn_pack2(best[u][0], best[u][1]); o.y = mgn_pack2(best[u][2], best[u][3]);   // (exact: the maxima are 16-bit values)
            o.z = mgn_pack2(best[u][4], best[u][5]); o.w = mgn_pack2(best[u][6], best[u][7]);
            const long op = (((long)n * OH + oh) * OW + ow0 + u) * C + c8 * 8;
            *reinterpret_cast<uint4*>(y + op) = o;
            uint2 a;
            a.x = arg[u][0] | (arg[u][1] << 8) | (arg[u][2] << 16) | ((uint32_t)arg[u][3] << 24);
            a.y = arg[u][4] | (arg[u][5] << 8) | (arg[u][6] << 16) | ((uint32_t)arg[u][7] << 24);
            *reinterpret_cast<uint2*>(idx + op) = a;
        }
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) top[j][k] = f2[j][k];
        top_ok = ok2;
    }
}

struct AbnBwd {
    const float* scale; const float* offset; const float* weight; const float* bias; const float* rstd; const float* sums;
    float inv_n, eps, slope;
    int leaky;
};
MGN_PLAN_RO(AbnBwd, MGN_RO(scale) MGN_RO(offset) MGN_RO(weight) MGN_RO(bias) MGN_RO(rstd) MGN_RO(sums))   // pointers the kernels only read through (launch-plan dependency analysis, csrc/mgn_launch.h)

__global__ __launch_bounds__(256) void abn_maxpool_bwd(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                       const uint8_t* __restrict__ idx, uint16_t* __restrict__ dx, AbnBwd q, int N, int IH,
                                                       int IW, int C, int OH, int OW) {
    const int cv = C / 8, PH = (IH + 1) / 2, PW = (IW + 1) / 2;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)N * PH * PW * cv) return;
    const int c8 = (int)(t % cv);
    long r = t / cv;
    const int j = (int)(r % PW); r /= PW;
    const int i = (int)(r % PH);
    const int n = (int)(r / PH);
    float acc[2][2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[a][b][k] = 0.f;
    // pooled (i + wo, j + wx) looks at input (2 (i + wo) - 1 + kh, 2 (j + wx) - 1 + kw), so with tap = 3 kh + kw the patch cells it can feed are
    //   (wo, wx) = (0, 0): taps 4, 5, 7, 8 -> cells (0,0), (0,1), (1,0), (1,1)      (0, 1): taps 3, 6 -> cells (0,1), (1,1)
    //   (wo, wx) = (1, 0): taps 1, 2       -> cells (1,0), (1,1)                    (1, 1): tap 0     -> cell (1,1)
    // = nine compare-selects per channel (round 4; the generic form decoded kh, kw by division and tested all 16 (window, cell) pairs)
#pragma unroll
    for (int wo = 0; wo < 2; ++wo) {
        const int oh = i + wo;
        if (oh >= OH) continue;
#pragma unroll
        for (int wx = 0; wx < 2; ++wx) {
            const int ow = j + wx;
            if (ow >= OW) continue;
            const long op = (((long)n * OH + oh) * OW + ow) * C + c8 * 8;
            const uint4 g = *reinterpret_cast<const uint4*>(dy + op);
            const uint2 am = *reinterpret_cast<const uint2*>(idx + op);
            const uint32_t gw[4] = {g.x, g.y, g.z, g.w};
            const uint32_t aw[2] = {am.x, am.y};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int tap = (int)((aw[k >> 2] >> ((k & 3) * 8)) & 0xff);
                const float v = bf2f((uint16_t)(gw[k >> 1] >> ((k & 1) * 16)));
                if (wo == 0 && wx == 0) {
                    acc[0][0][k] += tap == 4 ? v : 0.f; acc[0][1][k] += tap == 5 ? v : 0.f;
                    acc[1][0][k] += tap == 7 ? v : 0.f; acc[1][1][k] += tap == 8 ? v : 0.f;
                } else if (wo == 0) {
                    acc[0][1][k] += tap == 3 ? v : 0.f; acc[1][1][k] += tap == 6 ? v : 0.f;
                } else if (wx == 0) {
                    acc[1][0][k] += tap == 1 ? v : 0.f; acc[1][1][k] += tap == 2 ? v : 0.f;
                } else {
                    acc[1][1][k] += tap == 0 ? v : 0.f;
                }
            }
        }
    }
    // dx = A (dz - m1) - (z - beta) B,  A = gamma' rstd, m1 = sum_dz / n, B = rstd sum_dzxh / n   (iabn_bwd_apply)
    float sc[8], of[8], A[8], m1[8], Bc[8], bk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = c8 * 8 + k;
        const float rs = q.rstd[c];
        sc[k] = q.scale[c]; of[k] = q.offset[c];
        A[k] = (fabsf(q.weight[c]) + q.eps) * rs;
        m1[k] = q.sums[c] * q.inv_n;
        Bc[k] = rs * q.sums[C + c] * q.inv_n;
        bk[k] = q.bias[c];
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int ih = 2 * i + a;
        if (ih >= IH) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int iw = 2 * j + b;
            if (iw >= IW) continue;
            const long ip = (((long)n * IH + ih) * IW + iw) * C + c8 * 8;
            const uint4 xv = *reinterpret_cast<const uint4*>(x + ip);
            const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w};
            uint32_t o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float z = fmaf(bf2f((uint16_t)(xw[k >> 1] >> ((k & 1) * 16))), sc[k], of[k]);
                float dz = acc[a][b][k];
                if (q.leaky && z < 0.f) dz *= q.slope;
                o[k] = f2bf_rne(A[k] * (dz - m1[k]) - (z - bk[k]) * Bc[k]);
            }
            *reinterpret_cast<uint4*>(dx + ip) = make_uint4(o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16));
        }
    }
}

}  // namespace

extern "C" {

int MGN_SYM(mgn_abn_maxpool_fwd)(const void* x_bf16, const float* scale, const float* offset, int activation, float slope, void* y_bf16,
                        uint8_t* argmax, int N, int IH, int IW, int C, void* stream) {
    if (!x_bf16 || !scale || !offset || !y_bf16 || !argmax || N < 1 || IH < 1 || IW < 1 || C < 8 || C % 8) return MGN_EINVAL;
    if (activation != 0 && activation != 1) return MGN_EINVAL;
    const int OH = (IH + 2 - 3) / 2 + 1, OW = (IW + 2 - 3) / 2 + 1;
    static const bool plain = getenv("MGN_POOL_NO_MARCH") != nullptr;   // (A/B switch: the one-row-per-thread kernel)
    if (plain || OH < 4 * POOL_ROWS) {
        const long n = (long)N * OH * ((OW + 1) / 2) * (C / 8);   // a thread per 8-channel vector of two adjacent output pixels
        hipLaunchKernelGGL(abn_maxpool_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x_bf16, scale,
                           offset, activation, slope, (uint16_t*)y_bf16, argmax, N, IH, IW, C, OH, OW);
    } else {   // the same pair of pixels for POOL_ROWS consecutive output rows: every input row is loaded and activated once
        const long n = (long)N * ((OH + POOL_ROWS - 1) / POOL_ROWS) * ((OW + 1) / 2) * (C / 8);
        hipLaunchKernelGGL(abn_maxpool_fwd_march, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x_bf16,
                           scale, offset, activation, slope, (uint16_t*)y_bf16, argmax, N, IH, IW, C, OH, OW);
    }
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int MGN_SYM(mgn_abn_maxpool_bwd)(const void* x_bf16, const void* dpool_bf16, const uint8_t* argmax, void* dx_bf16, const float* scale,
                        const float* offset, const float* weight, const float* bias, const float* rstd, const float* sums,
                        float total_count, float eps, int activation, float slope, int N, int IH, int IW, int C, void* stream) {
    if (!x_bf16 || !dpool_bf16 || !argmax || !dx_bf16 || !scale || !offset || !weight || !bias || !rstd || !sums) return MGN_EINVAL;
    if (N < 1 || IH < 1 || IW < 1 || C < 8 || C % 8 || !(total_count > 0.f) || (activation != 0 && activation != 1)) return MGN_EINVAL;
    const int OH = (IH + 2 - 3) / 2 + 1, OW = (IW + 2 - 3) / 2 + 1;
    const long n = (long)N * ((IH + 1) / 2) * ((IW + 1) / 2) * (C / 8);
    AbnBwd q = {scale, offset, weight, bias, rstd, sums, 1.f / total_count, eps, slope, activation};
    hipLaunchKernelGGL(abn_maxpool_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x_bf16,
                       (const uint16_t*)dpool_bf16, argmax, (uint16_t*)dx_bf16, q, N, IH, IW, C, OH, OW);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int MGN_SYM(mgn_maxpool3x3s2_fwd)(const void* x_bf16, void* y_bf16, uint8_t* argmax, int N, int IH, int IW, int C, void* stream) {
    if (!x_bf16 || !y_bf16 || !argmax || N < 1 || IH < 1 || IW < 1 || C < 8 || C % 8) return MGN_EINVAL;
    const int OH = (IH + 2 - 3) / 2 + 1, OW = (IW + 2 - 3) / 2 + 1;
    const long n = (long)N * OH * OW * (C / 8);
    hipLaunchKernelGGL(maxpool_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x_bf16,
                       (uint16_t*)y_bf16, argmax, N, IH, IW, C, OH, OW);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int MGN_SYM(mgn_maxpool3x3s2_bwd)(const void* dy_bf16, const uint8_t* argmax, void* dx_bf16, int N, int IH, int IW, int C, void* stream) {
    if (!dy_bf16 || !dx_bf16 || !argmax || N < 1 || IH < 1 || IW < 1 || C < 8 || C % 8) return MGN_EINVAL;
    const int OH = (IH + 2 - 3) / 2 + 1, OW = (IW + 2 - 3) / 2 + 1;
    const long n = (long)N * ((IH + 1) / 2) * ((IW + 1) / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dy_bf16, argmax,
                       (uint16_t*)dx_bf16, N, IH, IW, C, OH, OW);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

}  // extern "C"

// headloss.hip -- losses of the panoptic heads fused with the x8 bilinear upsampling of their low-resolution outputs.
//
// Replaces (reference file:line)
//   mgnet/modeling/mg_net.py:597-610  MGNetSemSegHead.forward/.losses: F.interpolate(logits, x8, bilinear,
//       align_corners=True) -> [B,20,H,W] fp32 (1.34 GB at B=8 1024x2048) -> loss.py:45-81 OhemCE (per-pixel CE x weights,
//       FULL torch.sort over B*H*W values, threshold / top-n_min selection, mean)
//   mgnet/modeling/mg_net.py:676-715  MGNetInsEmbedHead.forward/.losses: x8 upsampling of centre (after sigmoid) and
//       offset (then x8), weighted MSE / L1 sums divided by the weight sums.
// Here the full-resolution maps are never materialised: every output pixel interpolates its 4 low-resolution
// neighbours on the fly (the low-res maps are 64x smaller and stay in L2), both in the forward and in the backward.
//
// OHEM without a sort (result-identical selection): sorted[n_min] > thr  <=>  count(loss > thr) > n_min, then the mean of
// {loss > thr}; otherwise the mean of the n_min largest losses, which only needs the n_min-th largest VALUE (host side:
// torch.topk on the per-pixel loss map this kernel writes; rare late-training branch).
//
// Backward = gather-free two-phase tile kernel: phase 1 recomputes softmax / residuals of a 16x32 pixel tile into LDS,
// phase 2 lets each (low-res cell, channel) of the tile's footprint sum its bilinear-weighted contributions in a fixed
// order; only the footprint cells shared between tiles are combined with global fp32 atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mgnet_hip.h"
#include "h16.h"

namespace {

constexpr int TPB = 256;
constexpr int TX = 32;               // backward pixel tile: TX x TYT, TYT = 8 for the semantic head (34 KB of LDS per block: four blocks per
                                     // CU hide the corner-load latency; 16 rows ran at two) and 16 for the 1..3-channel adjoints
constexpr int TYU = 8, TYS = 16;

struct UpGeom {
    int B, h, w, H, W;
    int bx0, by0;      // tile offset of this launch (0 unless MGN_SERIAL_SCATTER: one tile per launch, see serial_scatter())
    float* foot;       // footprint table [B][tiles y][tiles x][fr][fc][KK] of the backward tile kernels, or null (float atomics)
    int fr, fc;        // footprint bound of this geometry (rows, columns)
    int ty;            // tile height of the kernel that fills the table
    int kt;            // channels per table entry
    long sb, sh, sw;   // element strides of the low-res map (channel stride 1)
    float ry, rx;      // (h-1)/(H-1), (w-1)/(W-1)   (align_corners=True)
};

__device__ __forceinline__ float bf2f(uint16_t v) { return mgn_h2f(v); }   // this TU's 16-bit format (h16.h)

struct Corner {
    long o00, o10, o01, o11;
    float w00, w10, w01, w11;
    int x0, y0, x1, y1;
    float tx, ty;
};

__device__ __forceinline__ Corner corners(const UpGeom& g, int b, int Y, int X) {
    Corner c;
    const float sy = Y * g.ry, sx = X * g.rx;  // torch area_pixel_compute_source_index(align_corners=True)
    c.y0 = min((int)sy, g.h - 1);
    c.x0 = min((int)sx, g.w - 1);
    c.y1 = min(c.y0 + 1, g.h - 1);
    c.x1 = min(c.x0 + 1, g.w - 1);
    c.ty = sy - c.y0;
    c.tx = sx - c.x0;
    c.w00 = (1.f - c.ty) * (1.f - c.tx);
    c.w10 = (1.f - c.ty) * c.tx;
    c.w01 = c.ty * (1.f - c.tx);
    c.w11 = c.ty * c.tx;
    const long base = (long)b * g.sb;
    c.o00 = base + c.y0 * g.sh + c.x0 * g.sw;
    c.o10 = base + c.y0 * g.sh + c.x1 * g.sw;
    c.o01 = base + c.y1 * g.sh + c.x0 * g.sw;
    c.o11 = base + c.y1 * g.sh + c.x1 * g.sw;
    return c;
}

// interpolated logits of one pixel (K <= 32 classes, bf16 low-res map); returns log-sum-exp
template <int K8, bool LSE = true>
__device__ __forceinline__ float interp_logits(const uint16_t* __restrict__ lg, const Corner& c, int K, float (&z)[K8 * 8]) {
    float m = -3.0e38f;
#pragma unroll
    for (int v = 0; v < K8; ++v) {
        const uint4 a = *reinterpret_cast<const uint4*>(lg + c.o00 + v * 8), b = *reinterpret_cast<const uint4*>(lg + c.o10 + v * 8);
        const uint4 d = *reinterpret_cast<const uint4*>(lg + c.o01 + v * 8), e = *reinterpret_cast<const uint4*>(lg + c.o11 + v * 8);
        const uint32_t A[4] = {a.x, a.y, a.z, a.w}, Bv[4] = {b.x, b.y, b.z, b.w}, D[4] = {d.x, d.y, d.z, d.w}, E[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int sh = (q & 1) * 16;
            const float va = bf2f((uint16_t)(A[q >> 1] >> sh)), vb = bf2f((uint16_t)(Bv[q >> 1] >> sh));
            const float vd = bf2f((uint16_t)(D[q >> 1] >> sh)), ve = bf2f((uint16_t)(E[q >> 1] >> sh));
            const float val = c.w00 * va + c.w10 * vb + c.w01 * vd + c.w11 * ve;
            z[v * 8 + q] = val;
            if (v * 8 + q < K) m = fmaxf(m, val);
        }
    }
    if (!LSE) return m;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K8 * 8; ++k)
        if (k < K) s += __expf(z[k] - m);
    return m + __logf(s);
}

// Phase 2 of the backward tile kernels, separable: rows first (sum over the tile's x with the horizontal bilinear weight
// of every footprint column), then columns.  res: [KK][TY][RP] residuals in LDS; rowsum: [TY][MAXC][KK] scratch in LDS.
constexpr int MAXC = 8, MAXR = 6;  // footprint bound of a 32x16 tile for scale >= 4 (checked on the host)
constexpr int RP = TX + 4;         // pitch of a residual row in LDS: res[(k*TY + yy)*RP + xx].  36 words: rows are 16-byte aligned
                                   // (the row pass reads them as float4) and eight consecutive rows start in eight different bank
                                   // quads, so those reads -- lanes along yy -- are conflict-free
template <int TY> __device__ __forceinline__ int res_idx(int k, int yy, int xx) { return (k * TY + yy) * RP + xx; }
// rowsum[yy][cxi][k] with odd pitches: the column pass reads it with lanes along k (consecutive banks), the row pass writes it
// with lanes along yy (stride YP, odd -> 16 different banks).  (The [k][yy][cxi] layout this replaces put all 24 lanes of a
// column-pass read on ONE bank: SQ_LDS_BANK_CONFLICT was 52 % of the LDS cycles of upce_bwd, profiles/r02_upce_pmc.txt.)
template <int KK, int TY> struct ScatterLds {
    static constexpr int KKP = KK | 1, YP = MAXC * KKP + 1;
    static constexpr int FLOATS = KK * TY * RP + TY * YP;   // res + rowsum
};

template <int KK, int TY>
__device__ __forceinline__ void scatter_tile(const UpGeom& g, int b, int X0, int Y0, int K, const float* res, float* rowsum,
                                             float* out, int out_stride) {
    __shared__ float wxs[TX * MAXC], wys[TY * MAXR];   // bilinear weight of tile column xx (row yy) on footprint column cxi (row cyi)
    const int ly0 = min((int)(Y0 * g.ry), g.h - 1), lx0 = min((int)(X0 * g.rx), g.w - 1);
    const int Yl = min(Y0 + TY, g.H) - 1, Xl = min(X0 + TX, g.W) - 1;
    const int ly1 = min(min((int)(Yl * g.ry), g.h - 1) + 1, g.h - 1), lx1 = min(min((int)(Xl * g.rx), g.w - 1) + 1, g.w - 1);
    const int nr = min(ly1 - ly0 + 1, MAXR), nc = min(lx1 - lx0 + 1, MAXC);
    const int ny = Yl - Y0 + 1, nx = Xl - X0 + 1;
    for (int o = threadIdx.x; o < TX * MAXC; o += TPB) {
        const int xx = o / MAXC, cxi = o % MAXC;
        float wgt = 0.f;
        if (xx < nx && cxi < nc) {
            const float sx = (X0 + xx) * g.rx;
            const int x0 = min((int)sx, g.w - 1), x1 = min(x0 + 1, g.w - 1), cx = lx0 + cxi;
            const float tx = sx - x0;
            wgt = (x0 == cx ? 1.f - tx : 0.f) + (x1 == cx ? tx : 0.f);
        }
        wxs[o] = wgt;
    }
    for (int o = threadIdx.x; o < TY * MAXR; o += TPB) {
        const int yy = o / MAXR, cyi = o % MAXR;
        float wgt = 0.f;
        if (yy < ny && cyi < nr) {
            const float sy = (Y0 + yy) * g.ry;
            const int y0 = min((int)sy, g.h - 1), y1 = min(y0 + 1, g.h - 1), cy = ly0 + cyi;
            const float ty = sy - y0;
            wgt = (y0 == cy ? 1.f - ty : 0.f) + (y1 == cy ? ty : 0.f);
        }
        wys[o] = wgt;
    }
    __syncthreads();
    // rows: rowsum[yy][cxi][k] = sum_xx wxs[xx][cxi] * res[k][yy][xx]      (lanes: yy fastest, then k)
    using L = ScatterLds<KK, TY>;
    constexpr bool UNI = (TY * KK) % 64 == 0;   // cxi is the same for a whole wave: its 32 weights travel through ONE LDS read per
                                                // lane and v_readlane (scalar operands of the FMAs) instead of 32 broadcast reads
    for (int o = threadIdx.x; o < TY * KK * nc; o += TPB) {
        const int yy = o % TY, k = (o / TY) % KK, cxi = o / (TY * KK);
        float acc = 0.f;
        const float* rr = res + res_idx<TY>(k, yy, 0);
        if (UNI) {
            const float wl = wxs[(threadIdx.x & 31) * MAXC + cxi];
            float wv[TX];
#pragma unroll
            for (int xx = 0; xx < TX; ++xx) wv[xx] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl), xx));
            if (k < K) {   // (rows k >= K of res are not written by every caller)
#pragma unroll
                for (int q = 0; q < TX / 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(rr + 4 * q);   // rows beyond ny / columns beyond nx hold zeros
                    acc = fmaf(wv[4 * q], v.x, acc); acc = fmaf(wv[4 * q + 1], v.y, acc);   // (the order of the scalar loop)
                    acc = fmaf(wv[4 * q + 2], v.z, acc); acc = fmaf(wv[4 * q + 3], v.w, acc);
                }
            }
        } else if (k < K && yy < ny) {
#pragma unroll 8
            for (int xx = 0; xx < TX; ++xx) acc += wxs[xx * MAXC + cxi] * rr[xx];   // columns beyond nx carry weight 0
        }
        rowsum[yy * L::YP + cxi * L::KKP + k] = acc;
    }
    __syncthreads();
    if (g.foot) {
        // reproducible form: the tile's footprint goes to its own slot of the table with plain stores (zeros outside nr x nc and
        // for k >= K); adjoint_gather() then sums, per low-res element, the slots that cover it in a FIXED order
        const size_t ti = ((size_t)b * ((g.H + TY - 1) / TY) + Y0 / TY) * ((g.W + TX - 1) / TX) + X0 / TX;
        const int KT = g.kt;                              // table channels: K rounded up to a multiple of 4 (1 for one channel)
        float* slot = g.foot + ti * g.fr * g.fc * KT;     // [fr][fc][KT]
        for (int o = threadIdx.x; o < g.fr * g.fc * KT; o += TPB) {
            const int k = o % KT, cxi = (o / KT) % g.fc, cyi = o / (KT * g.fc);
            float acc = 0.f;
            if (k < K && cxi < nc && cyi < nr) {
#pragma unroll
                for (int yy = 0; yy < TY; ++yy) acc += wys[yy * MAXR + cyi] * rowsum[yy * L::YP + cxi * L::KKP + k];
            }
            slot[o] = acc;
        }
        return;
    }
    for (int o = threadIdx.x; o < nr * nc * KK; o += TPB) {
        const int k = o % KK, cxi = (o / KK) % nc, cyi = o / (KK * nc);
        if (k >= K) continue;
        float acc = 0.f;
#pragma unroll
        for (int yy = 0; yy < TY; ++yy) acc += wys[yy * MAXR + cyi] * rowsum[yy * L::YP + cxi * L::KKP + k];
        if (acc != 0.f) atomicAdd(out + (((long)b * g.h + ly0 + cyi) * g.w + lx0 + cxi) * out_stride + k, acc);
    }
}

// Second half of the reproducible adjoint: out[b, y, x, :] = sum over the tiles whose footprint covers (y, x), tile rows then tile
// columns ascending, of their table entries.  The footprint origin of a tile is recomputed with scatter_tile's own expressions.
// A thread per low-res pixel (the search for the covering tiles is done once, not per channel); the table keeps KT = 4 * KV channels
// per entry (K rounded up: 16-byte loads); destination channels >= KT are zero.  KV = 0: single-channel tables, scalar.
template <int KV>
__global__ __launch_bounds__(TPB) void adjoint_gather(UpGeom g, float* __restrict__ out, int out_stride) {
    constexpr int KT = KV ? 4 * KV : 1;
    const long i = (long)blockIdx.x * TPB + threadIdx.x;
    if (i >= (long)g.B * g.h * g.w) return;
    long r = i;
    const int x = (int)(r % g.w); r /= g.w;
    const int y = (int)(r % g.h);
    const int b = (int)(r / g.h);
    float acc[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[k] = 0.f;
    const int TY = g.ty, tny = (g.H + TY - 1) / TY, tnx = (g.W + TX - 1) / TX;
    // candidate tiles: those whose pixel rows / columns map near y / x (a generous bracket, membership is tested exactly)
    int ty0 = 0, ty1 = tny - 1, tx0 = 0, tx1 = tnx - 1;
    if (g.ry > 0.f) { ty0 = max(0, (int)((y - 1) / g.ry) / TY - 1); ty1 = min(tny - 1, (int)((y + 1) / g.ry) / TY + 1); }
    if (g.rx > 0.f) { tx0 = max(0, (int)((x - 1) / g.rx) / TX - 1); tx1 = min(tnx - 1, (int)((x + 1) / g.rx) / TX + 1); }
    for (int ty = ty0; ty <= ty1; ++ty) {
        const int Y0 = ty * TY, Yl = min(Y0 + TY, g.H) - 1;
        const int ly0 = min((int)(Y0 * g.ry), g.h - 1), ly1 = min(min((int)(Yl * g.ry), g.h - 1) + 1, g.h - 1);
        if (y < ly0 || y > ly1 || y - ly0 >= g.fr) continue;
        for (int tx = tx0; tx <= tx1; ++tx) {
            const int X0 = tx * TX, Xl = min(X0 + TX, g.W) - 1;
            const int lx0 = min((int)(X0 * g.rx), g.w - 1), lx1 = min(min((int)(Xl * g.rx), g.w - 1) + 1, g.w - 1);
            if (x < lx0 || x > lx1 || x - lx0 >= g.fc) continue;
            const size_t ti = ((size_t)b * tny + ty) * tnx + tx;
            const float* e = g.foot + (ti * g.fr * g.fc + (size_t)(y - ly0) * g.fc + (x - lx0)) * KT;
            if (KV) {
#pragma unroll
                for (int v = 0; v < (KV ? KV : 1); ++v) {
                    const float4 q = *reinterpret_cast<const float4*>(e + 4 * v);
                    acc[4 * v] += q.x; acc[4 * v + 1] += q.y; acc[4 * v + 2] += q.z; acc[4 * v + 3] += q.w;
                }
            } else {
                acc[0] += e[0];
            }
        }
    }
    float* o = out + i * out_stride;
    if (KV && out_stride % 4 == 0) {
#pragma unroll
        for (int v = 0; v < 8; ++v) {   // (out_stride <= 32)
            if (4 * v >= out_stride) break;
            const float4 q = v < KV ? make_float4(acc[4 * (v < KV ? v : 0)], acc[4 * (v < KV ? v : 0) + 1], acc[4 * (v < KV ? v : 0) + 2], acc[4 * (v < KV ? v : 0) + 3])
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(o + 4 * v) = q;
        }
    } else {
        for (int k = 0; k < out_stride; ++k) {
            float v = 0.f;
#pragma unroll
            for (int kk = 0; kk < KT; ++kk)
                if (kk == k) v = acc[kk];
            o[k] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// semantic head forward: per-pixel weighted CE map + block partials {count(ce>thr), sum(ce | ce>thr), sum(ce)}
// ---------------------------------------------------------------------------------------------------------------
template <int K8>
__global__ __launch_bounds__(TPB) void upce_fwd(const uint16_t* __restrict__ lg, UpGeom g, int K, const long* __restrict__ labels,
                                                const float* __restrict__ weights, int ignore, float thr,
                                                float* __restrict__ ce_map, float* partials) {
    __shared__ float red[3][TPB / 64];
    const int X = blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    float ce = 0.f;
    const bool in = X < g.W && Y < g.H;
    if (in) {
        const long p = ((long)b * g.H + Y) * g.W + X;
        const long lab = labels[p];
        if (lab != ignore) {
            float z[K8 * 8];
            const Corner c = corners(g, b, Y, X);
            const float lse = interp_logits<K8>(lg, c, K, z);
            float zl = 0.f;
#pragma unroll
            for (int k = 0; k < K8 * 8; ++k)
                if (k == (int)lab) zl = z[k];
            ce = (lse - zl) * (weights ? weights[p] : 1.f);
        }
        ce_map[p] = ce;
    }
    float cnt = (in && ce > thr) ? 1.f : 0.f, sh = (in && ce > thr) ? ce : 0.f, sa = in ? ce : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o); sh += __shfl_xor(sh, o); sa += __shfl_xor(sa, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = cnt; red[1][threadIdx.x >> 6] = sh; red[2][threadIdx.x >> 6] = sa; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partials[(size_t)blk * 3 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    }
}

// Final sums over the per-block partials [nblk][NK] (65536 blocks at 8 x 1024 x 2048): ONE block of 1024 threads, four
// independent fp64 accumulator sets per thread so that the loads overlap (the 256-thread dependent-chain version took 72 us),
// then a fixed-order LDS tree.  Result in sh[k][0].
constexpr int SUMT = 1024;
template <int NK>
__device__ __forceinline__ void block_sums(const float* __restrict__ partials, int nblk, double (*sh)[SUMT]) {
    double a[4][NK];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < NK; ++k) a[u][k] = 0.0;
    int i = threadIdx.x;
    for (; i + 3 * SUMT < nblk; i += 4 * SUMT) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < NK; ++k) a[u][k] += (double)partials[(size_t)(i + u * SUMT) * NK + k];
    }
    for (; i < nblk; i += SUMT)
#pragma unroll
        for (int k = 0; k < NK; ++k) a[0][k] += (double)partials[(size_t)i * NK + k];
#pragma unroll
    for (int k = 0; k < NK; ++k) sh[k][threadIdx.x] = (a[0][k] + a[1][k]) + (a[2][k] + a[3][k]);
    __syncthreads();
    for (int o = SUMT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
#pragma unroll
            for (int k = 0; k < NK; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
        __syncthreads();
    }
}

// out[0]=n_hard, out[1]=sum_hard, out[2]=sum_all  (fp64 accumulation, fixed order)
__global__ __launch_bounds__(SUMT) void sum3_kernel(const float* partials, int nblk, float* out) {
    __shared__ double sh[3][SUMT];
    block_sums<3>(partials, nblk, sh);
    if (threadIdx.x < 3) out[threadIdx.x] = (float)sh[threadIdx.x][0];
}

// ---------------------------------------------------------------------------------------------------------------
// semantic head backward.  sel = {tau, tie_weight, scale}: d loss / d ce(p) = scale * (ce > tau ? 1 : (ce == tau ? tie_weight : 0))
// dlg: [B, h, w, Kp] fp32 (zero-initialised), accumulates d loss / d low-res logits
// ---------------------------------------------------------------------------------------------------------------
template <int K8>
__global__ __launch_bounds__(TPB) void upce_bwd(const uint16_t* __restrict__ lg, UpGeom g, int K, int Kp, const long* __restrict__ labels,
                                                const float* __restrict__ weights, int ignore, const float* __restrict__ ce_map,
                                                const float* __restrict__ sel, const float* __restrict__ gout, float* dlg) {
    extern __shared__ __attribute__((aligned(16))) float res[];  // [K8*8][TY][RP] residuals g*(p_k - onehot), then rowsum
    constexpr int KK = K8 * 8, TY = TYU;
    const int X0 = (blockIdx.x + g.bx0) * TX, Y0 = (blockIdx.y + g.by0) * TY, b = blockIdx.z;
    const float tau = sel[0], tie_w = sel[1], scale = sel[2] * gout[0];
    // ---- phase 1: residual vectors of the tile's pixels.  A thread owns TWO pixels (rows yy and yy + 8 of column xx): their
    // scalars are loaded first, then -- under ONE wave-uniform branch, so that the twelve 16-byte corner loads of both pixels
    // sit in one basic block and are in flight together (2 waves per SIMD are all the LDS footprint leaves for hiding them) --
    // the interpolated soft-max of both
    constexpr int NPX = TX * TY / TPB;   // pixels per thread
    static_assert(TX * TY % TPB == 0, "whole pixels per thread");
    float gpx[NPX];
    int lab[NPX], Xc[NPX], Yc[NPX];
#pragma unroll
    for (int u = 0; u < NPX; ++u) {
        const int t = threadIdx.x + u * TPB;
        const int X = X0 + (t % TX), Y = Y0 + (t / TX);
        gpx[u] = 0.f;
        lab[u] = -1;
        Xc[u] = min(X, g.W - 1); Yc[u] = min(Y, g.H - 1);
        if (X < g.W && Y < g.H) {
            const long p = ((long)b * g.H + Y) * g.W + X;
            const long l = labels[p];
            const float ce = ce_map[p];
            const float sw = ce > tau ? 1.f : (ce == tau ? tie_w : 0.f);
            if (l != ignore && sw != 0.f) { gpx[u] = sw * scale * (weights ? weights[p] : 1.f); lab[u] = (int)l; }
        }
    }
    bool any_px = false;
#pragma unroll
    for (int u = 0; u < NPX; ++u) any_px = any_px || gpx[u] != 0.f;
    if (__any(any_px)) {
        float z[NPX][KK], inv[NPX];
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
            const Corner c = corners(g, b, Yc[u], Xc[u]);
            (void)interp_logits<K8, false>(lg, c, K, z[u]);
        }
#pragma unroll
        for (int u = 0; u < NPX; ++u) {   // soft-max as exp(z - max) / sum: one exponential per class (not a second one against the lse)
            float m = -3.0e38f;
#pragma unroll
            for (int k = 0; k < KK; ++k)
                if (k < K) m = fmaxf(m, z[u][k]);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < KK; ++k) {
                z[u][k] = k < K ? __expf(z[u][k] - m) : 0.f;
                sum += z[u][k];
            }
            inv[u] = gpx[u] / sum;
        }
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
            const int t = threadIdx.x + u * TPB;
#pragma unroll
            for (int k = 0; k < KK; ++k) res[res_idx<TY>(k, t / TX, t % TX)] = k < K ? fmaf(z[u][k], inv[u], k == lab[u] ? -gpx[u] : 0.f) : 0.f;
        }
    } else {
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
            const int t = threadIdx.x + u * TPB;
#pragma unroll
            for (int k = 0; k < KK; ++k) res[res_idx<TY>(k, t / TX, t % TX)] = 0.f;
        }
    }
    __syncthreads();
    // ---- phase 2: separable bilinear adjoint of the tile into its low-res footprint
    scatter_tile<KK, TY>(g, b, X0, Y0, K, res, res + KK * TY * RP, dlg, Kp);
}

// ---------------------------------------------------------------------------------------------------------------
// instance head: centre (sigmoid at low res, then x8) weighted MSE and offset (x8, then *scale) weighted L1
//   forward : partial sums {sum w_c*(c-t)^2, sum w_c, sum w_o*|o-t| (both channels), sum w_o}
//   backward: same two-phase tile scheme; dco [B,h,w,4] fp32 = {d/d centre_lowres(after sigmoid), d/d off0, d/d off1, 0}
// ---------------------------------------------------------------------------------------------------------------
struct InsMaps {
    const uint16_t* center;  // low-res, bf16 or null if f32
    const float* center_f;   // low-res fp32 (after sigmoid)
    const uint16_t* offset;  // low-res [.., 2+] bf16
    UpGeom gc, go;
    const float* ct;   // [B,1,H,W]
    const float* cw;   // [B,1,H,W]
    const float* ot;   // [B,2,H,W]
    const float* ow;   // [B,1,H,W]
    float oscale;      // common_stride multiplier applied after the upsampling (mg_net.py:682-694)
};
MGN_PLAN_RO(InsMaps, MGN_RO(center) MGN_RO(center_f) MGN_RO(offset) MGN_RO(ct) MGN_RO(cw) MGN_RO(ot) MGN_RO(ow))   // pointers the kernels only read through (launch-plan dependency analysis, csrc/mgn_launch.h)

__device__ __forceinline__ float interp1f(const float* m, const Corner& c) {
    return c.w00 * m[c.o00] + c.w10 * m[c.o10] + c.w01 * m[c.o01] + c.w11 * m[c.o11];
}
__device__ __forceinline__ float interp1b(const uint16_t* m, const Corner& c, int ch) {
    return c.w00 * bf2f(m[c.o00 + ch]) + c.w10 * bf2f(m[c.o10 + ch]) + c.w01 * bf2f(m[c.o01 + ch]) + c.w11 * bf2f(m[c.o11 + ch]);
}

__global__ __launch_bounds__(TPB) void ins_fwd(InsMaps m, float* partials) {
    __shared__ float red[4][TPB / 64];
    const int X = blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const int H = m.gc.H, W = m.gc.W;
    if (X < W && Y < H) {
        const long p = ((long)b * H + Y) * W + X;
        const Corner cc = corners(m.gc, b, Y, X);
        const float c = interp1f(m.center_f, cc);
        const float wc = m.cw[p], d = c - m.ct[p];
        v[0] = wc * d * d;
        v[1] = wc;
        const Corner co = corners(m.go, b, Y, X);
        const float wo = m.ow[p];
        const long p2 = ((long)b * 2 * H + Y) * W + X;
        const float o0 = interp1b(m.offset, co, 0) * m.oscale, o1 = interp1b(m.offset, co, 1) * m.oscale;
        v[2] = wo * (fabsf(o0 - m.ot[p2]) + fabsf(o1 - m.ot[p2 + (long)H * W]));
        v[3] = wo;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partials[(size_t)blk * 4 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    }
}

__global__ __launch_bounds__(SUMT) void sum4_kernel(const float* partials, int nblk, float* out) {
    __shared__ double sh[4][SUMT];
    block_sums<4>(partials, nblk, sh);
    // out = {loss_center (sum/wsum or 0), loss_offset, wsum_c, wsum_o}   (mg_net.py:697-715)
    if (threadIdx.x == 0) {
        out[0] = sh[1][0] > 0 ? (float)(sh[0][0] / sh[1][0]) : 0.f;
        out[1] = sh[3][0] > 0 ? (float)(sh[2][0] / sh[3][0]) : 0.f;
        out[2] = (float)sh[1][0];
        out[3] = (float)sh[3][0];
    }
}

// gout = {d/d loss_center, d/d loss_offset}; sums = output of sum4_kernel
__global__ __launch_bounds__(TPB) void ins_bwd(InsMaps m, const float* __restrict__ sums, const float* __restrict__ gout, float* dco) {
    constexpr int TY = TYS;
    __shared__ __attribute__((aligned(16))) float res[ScatterLds<4, TY>::FLOATS];
    const int X0 = (blockIdx.x + m.gc.bx0) * TX, Y0 = (blockIdx.y + m.gc.by0) * TY, b = blockIdx.z;
    const UpGeom& g = m.gc;  // centre and offset maps share the geometry
    const float sc = sums[2] > 0.f ? gout[0] / sums[2] : 0.f, so = sums[3] > 0.f ? gout[1] / sums[3] : 0.f;
    for (int t = threadIdx.x; t < TX * TY; t += TPB) {
        const int X = X0 + (t % TX), Y = Y0 + (t / TX);
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
        if (X < g.W && Y < g.H) {
            const long p = ((long)b * g.H + Y) * g.W + X;
            const Corner cc = corners(m.gc, b, Y, X);
            r0 = sc * m.cw[p] * 2.f * (interp1f(m.center_f, cc) - m.ct[p]);
            const Corner co = corners(m.go, b, Y, X);
            const long p2 = ((long)b * 2 * g.H + Y) * g.W + X;
            const float d0 = interp1b(m.offset, co, 0) * m.oscale - m.ot[p2], d1 = interp1b(m.offset, co, 1) * m.oscale - m.ot[p2 + (long)g.H * g.W];
            const float wo = so * m.ow[p] * m.oscale;
            r1 = wo * (float)((d0 > 0.f) - (d0 < 0.f));
            r2 = wo * (float)((d1 > 0.f) - (d1 < 0.f));
        }
        res[res_idx<TY>(0, t / TX, t % TX)] = r0; res[res_idx<TY>(1, t / TX, t % TX)] = r1; res[res_idx<TY>(2, t / TX, t % TX)] = r2;
    }
    __syncthreads();
    scatter_tile<4, TY>(g, b, X0, Y0, 3, res, res + 4 * TY * RP, dco, 4);
}

// a 32 x 16 pixel tile (the tallest) must fall into at most MAXC x MAXR low-res cells
inline bool footprint_ok(int h, int w, int H, int W) {
    constexpr int TY = TYS;
    const double rx = W > 1 ? (double)(w - 1) / (W - 1) : 0.0, ry = H > 1 ? (double)(h - 1) / (H - 1) : 0.0;
    return (TX - 1) * rx + 3 <= MAXC && (TY - 1) * ry + 3 <= MAXR;
}

// ---------------------------------------------------------------------------------------------------------------
// plain bilinear (align_corners=True) upsampling of single-channel fp32 maps and its adjoint (depth head,
// mg_net.py:804-807: x8 / x16 / x32 of the three inverse-depth predictions)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void up1_fwd(const float* __restrict__ lr, UpGeom g, float* __restrict__ out) {
    const int X = blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (X >= g.W || Y >= g.H) return;
    const Corner c = corners(g, b, Y, X);
    out[((long)b * g.H + Y) * g.W + X] = interp1f(lr, c);
}

__global__ __launch_bounds__(TPB) void up1_bwd(const float* __restrict__ dfull, UpGeom g, float* dlr) {
    constexpr int TY = TYS;
    __shared__ __attribute__((aligned(16))) float res[ScatterLds<1, TY>::FLOATS];
    const int X0 = (blockIdx.x + g.bx0) * TX, Y0 = (blockIdx.y + g.by0) * TY, b = blockIdx.z;
    for (int t = threadIdx.x; t < TX * TY; t += TPB) {
        const int X = X0 + (t % TX), Y = Y0 + (t / TX);
        res[res_idx<TY>(0, t / TX, t % TX)] = (X < g.W && Y < g.H) ? dfull[((long)b * g.H + Y) * g.W + X] : 0.f;
    }
    __syncthreads();
    scatter_tile<1, TY>(g, b, X0, Y0, 1, res, res + TY * RP, dlr, 1);
}

// ---------------------------------------------------------------------------------------------------------------
// OHEM / top-k selection entirely on the device (loss.py:67-81 without the full sort AND without the host round trip of
// `if pixel_losses[n_min] > thresh`): result-identical selection
//     count(ce > thr) > n_sel  ->  mean of {ce > thr}                       sel = {thr, 0, 1/count}
//     otherwise               ->  mean of the n_sel largest values         sel = {v_k, (n_sel - n_gt)/n_eq, 1/n_sel}
// v_k (the n_sel-th largest value) by a 4-pass radix select over the fp32 bit patterns of the loss map (ce >= 0, so the
// unsigned order is the float order); every pass leaves at once when the threshold branch is taken (the common case),
// so the selection costs a few microseconds of launch latency there.
// state: u32 {use_thr, prefix, k_remaining, pad}; hist: u32 [256]
// ---------------------------------------------------------------------------------------------------------------
struct OhemState { unsigned use_thr, prefix, k, pad; };

__global__ void ohem_init(const float* sums3, long n, long n_sel, int force_topk, OhemState* st, unsigned* hist) {
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        st->use_thr = (!force_topk && (double)sums3[0] > (double)n_sel) ? 1u : 0u;
        st->prefix = 0;
        st->k = (unsigned)(n_sel < 1 ? 1 : (n_sel > n ? n : n_sel));
    }
}

// histogram of byte `pass` (3 = most significant) of the values whose higher bytes equal the prefix found so far
__global__ __launch_bounds__(TPB) void ohem_hist(const float* __restrict__ ce, long n, int pass, const OhemState* st, unsigned* hist) {
    if (st->use_thr) return;
    __shared__ unsigned h[256];
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned prefix = st->prefix;
    const int hs = 8 * (pass + 1);
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
        const unsigned b = __float_as_uint(fmaxf(ce[i], 0.f));   // (a rounding-negative loss must not sort above the positives)
        if (pass == 3 || (b >> hs) == prefix) atomicAdd(&h[(b >> (8 * pass)) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 256 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// walk the bins from the top: the bin where the cumulative count reaches k holds the k-th largest value
__global__ void ohem_pick(OhemState* st, unsigned* hist) {
    if (threadIdx.x == 0 && !st->use_thr) {
        unsigned k = st->k, b = 255;
        for (;; --b) {
            const unsigned c = hist[b];
            if (c >= k || b == 0) break;
            k -= c;
        }
        st->k = k;
        st->prefix = (st->prefix << 8) | b;
    }
    __syncthreads();
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
}

// per-block partials {count(ce > v_k), count(ce == v_k), sum(ce | ce > v_k)} (fixed-order reduction in ohem_final)
__global__ __launch_bounds__(TPB) void ohem_tail(const float* __restrict__ ce, long n, const OhemState* st, float* partials) {
    if (st->use_thr) return;
    __shared__ float red[3][TPB / 64];
    const float vk = __uint_as_float(st->prefix);
    float ngt = 0.f, neq = 0.f, sgt = 0.f;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
        const float v = fmaxf(ce[i], 0.f);
        if (v > vk) { ngt += 1.f; sgt += v; } else if (v == vk) neq += 1.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ngt += __shfl_xor(ngt, o); neq += __shfl_xor(neq, o); sgt += __shfl_xor(sgt, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ngt; red[1][threadIdx.x >> 6] = neq; red[2][threadIdx.x >> 6] = sgt; }
    __syncthreads();
    if (threadIdx.x < 3) partials[(size_t)blockIdx.x * 3 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ void ohem_final(const float* sums3, float thr, long n_sel, const OhemState* st, const float* partials, int nblk, float* sel3, float* loss) {
    __shared__ double sh[3][TPB];
    double a[3] = {0, 0, 0};
    if (!st->use_thr)
        for (int i = threadIdx.x; i < nblk; i += TPB)
            for (int k = 0; k < 3; ++k) a[k] += (double)partials[(size_t)i * 3 + k];
    for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] = a[k];
    __syncthreads();
    for (int o = TPB / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o)
            for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (st->use_thr) {
            sel3[0] = thr; sel3[1] = 0.f; sel3[2] = 1.f / sums3[0];
            loss[0] = sums3[1] / sums3[0];
        } else {
            const double vk = (double)__uint_as_float(st->prefix), ngt = sh[0][0], neq = sh[1][0], sgt = sh[2][0];
            const double take = (double)n_sel - ngt;                     // how many of the values equal to v_k are selected
            sel3[0] = (float)vk;
            sel3[1] = neq > 0 ? (float)(take / neq) : 0.f;
            sel3[2] = (float)(1.0 / (double)n_sel);
            loss[0] = (float)((sgt + take * vk) / (double)n_sel);
        }
    }
}

// Atomic form of the backward kernels (footprints == NULL): the low-resolution footprints of neighbouring pixel tiles are added
// with float atomics, so the sum of the <= 4 contributions to a border pixel depends on the arrival order in its last bit (the
// reference's F.interpolate backward does the same).  MGN_SERIAL_SCATTER=1 (debugging, slow) launches one tile per launch in a
// fixed order.  The default path of the product is the footprint table + adjoint_gather (bit-reproducible).
template <typename F>
inline void serial_scatter(dim3 grid, F&& launch) {   // launch(grid, bx0, by0)
    if (!getenv("MGN_SERIAL_SCATTER")) { launch(grid, 0, 0); return; }
    for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) launch(dim3(1, 1, grid.z), (int)bx, (int)by);
}

inline int geom_ok(int B, int h, int w, int H, int W) { return B >= 1 && h >= 2 && w >= 2 && H >= h && W >= w; }

// exact footprint bound of this geometry: the largest low-res extent of any TX x TY tile, with scatter_tile's own (float) expressions
inline void footprint_bound(int h, int w, int H, int W, int TY, int* fr, int* fc) {
    const float ry = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, rx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    int r = 1, c = 1;
    for (int Y0 = 0; Y0 < H; Y0 += TY) {
        const int Yl = (Y0 + TY < H ? Y0 + TY : H) - 1;
        int l0 = (int)(Y0 * ry), l1 = (int)(Yl * ry);
        l0 = l0 < h - 1 ? l0 : h - 1; l1 = l1 < h - 1 ? l1 : h - 1; l1 = l1 + 1 < h - 1 ? l1 + 1 : h - 1;
        r = l1 - l0 + 1 > r ? l1 - l0 + 1 : r;
    }
    for (int X0 = 0; X0 < W; X0 += TX) {
        const int Xl = (X0 + TX < W ? X0 + TX : W) - 1;
        int l0 = (int)(X0 * rx), l1 = (int)(Xl * rx);
        l0 = l0 < w - 1 ? l0 : w - 1; l1 = l1 < w - 1 ? l1 : w - 1; l1 = l1 + 1 < w - 1 ? l1 + 1 : w - 1;
        c = l1 - l0 + 1 > c ? l1 - l0 + 1 : c;
    }
    *fr = r < MAXR ? r : MAXR; *fc = c < MAXC ? c : MAXC;
}
inline size_t footprint_floats(int B, int h, int w, int H, int W, int KK, int TY) {
    int fr, fc;
    footprint_bound(h, w, H, W, TY, &fr, &fc);
    return (size_t)B * ((H + TY - 1) / TY) * ((W + TX - 1) / TX) * fr * fc * KK;
}

inline UpGeom make_geom(int B, int h, int w, int H, int W, long sb, long sh, long sw, int TY = TYS) {
    UpGeom g;
    g.B = B; g.h = h; g.w = w; g.H = H; g.W = W; g.sb = sb; g.sh = sh; g.sw = sw; g.bx0 = g.by0 = 0;
    g.foot = nullptr; g.ty = TY; g.kt = 1;
    footprint_bound(h, w, H, W, TY, &g.fr, &g.fc);
    g.ry = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    g.rx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    return g;
}

inline int table_channels(int K) { return K == 1 ? 1 : (K + 3) / 4 * 4; }

inline void launch_gather(const UpGeom& g, float* out, int out_stride, hipStream_t s) {
    const long n = (long)g.B * g.h * g.w;
    const dim3 grid((unsigned)((n + TPB - 1) / TPB)), blk(TPB);
    switch (g.kt / 4) {
        case 0: hipLaunchKernelGGL(adjoint_gather<0>, grid, blk, 0, s, g, out, out_stride); break;
        case 1: hipLaunchKernelGGL(adjoint_gather<1>, grid, blk, 0, s, g, out, out_stride); break;
        case 2: hipLaunchKernelGGL(adjoint_gather<2>, grid, blk, 0, s, g, out, out_stride); break;
        case 3: hipLaunchKernelGGL(adjoint_gather<3>, grid, blk, 0, s, g, out, out_stride); break;
        case 4: hipLaunchKernelGGL(adjoint_gather<4>, grid, blk, 0, s, g, out, out_stride); break;
        case 5: hipLaunchKernelGGL(adjoint_gather<5>, grid, blk, 0, s, g, out, out_stride); break;
        case 6: hipLaunchKernelGGL(adjoint_gather<6>, grid, blk, 0, s, g, out, out_stride); break;
        case 7: hipLaunchKernelGGL(adjoint_gather<7>, grid, blk, 0, s, g, out, out_stride); break;
        default: hipLaunchKernelGGL(adjoint_gather<8>, grid, blk, 0, s, g, out, out_stride); break;
    }
}

}  // namespace

extern "C" {

#ifndef MGN_F16
int mgn_upce_partials(int B, int H, int W) { return B * ((H + 3) / 4) * ((W + 63) / 64); }
#endif

int MGN_SYM(mgn_upce_fwd)(const void* logits_bf16, long sb, long sh, long sw, int B, int h, int w, int H, int W, int K, const long* labels,
                 const float* weights, int ignore, float thr, float* ce_map, float* partials, float* sums3, void* stream) {
    if (!logits_bf16 || !labels || !ce_map || !partials || !sums3 || !geom_ok(B, h, w, H, W)) return MGN_EINVAL;
    if (K < 1 || K > 32 || sw % 8 != 0 || sh % 8 != 0 || sb % 8 != 0 || sw < ((K + 7) / 8) * 8) return MGN_ENOTSUP;
    const UpGeom g = make_geom(B, h, w, H, W, sb, sh, sw);
    const dim3 grid((W + 63) / 64, (H + 3) / 4, B);
    hipStream_t s = (hipStream_t)stream;
    const uint16_t* lg = (const uint16_t*)logits_bf16;
    switch ((K + 7) / 8) {
        case 1: hipLaunchKernelGGL(upce_fwd<1>, grid, dim3(TPB), 0, s, lg, g, K, labels, weights, ignore, thr, ce_map, partials); break;
        case 2: hipLaunchKernelGGL(upce_fwd<2>, grid, dim3(TPB), 0, s, lg, g, K, labels, weights, ignore, thr, ce_map, partials); break;
        case 3: hipLaunchKernelGGL(upce_fwd<3>, grid, dim3(TPB), 0, s, lg, g, K, labels, weights, ignore, thr, ce_map, partials); break;
        default: hipLaunchKernelGGL(upce_fwd<4>, grid, dim3(TPB), 0, s, lg, g, K, labels, weights, ignore, thr, ce_map, partials); break;
    }
    hipLaunchKernelGGL(sum3_kernel, dim3(1), dim3(SUMT), 0, s, partials, (int)(grid.x * grid.y * grid.z), sums3);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int MGN_SYM(mgn_upce_bwd)(const void* logits_bf16, long sb, long sh, long sw, int B, int h, int w, int H, int W, int K, int Kp,
                 const long* labels, const float* weights, int ignore, const float* ce_map, const float* sel3, const float* gout,
                 float* dlogits, float* footprints, void* stream) {
    if (!logits_bf16 || !labels || !ce_map || !sel3 || !gout || !dlogits || !geom_ok(B, h, w, H, W)) return MGN_EINVAL;
    if (K < 1 || K > 32 || Kp < K || sw % 8 != 0 || sh % 8 != 0 || sb % 8 != 0) return MGN_ENOTSUP;
    if (!footprint_ok(h, w, H, W)) return MGN_ENOTSUP;
    UpGeom g = make_geom(B, h, w, H, W, sb, sh, sw, TYU);
    g.foot = footprints; g.kt = table_channels(K);
    const dim3 grid((W + TX - 1) / TX, (H + TYU - 1) / TYU, B);
    hipStream_t s = (hipStream_t)stream;
    const uint16_t* lg = (const uint16_t*)logits_bf16;
    const int k8 = (K + 7) / 8;
    const size_t lds = sizeof(float) * (k8 == 1 ? ScatterLds<8, TYU>::FLOATS : k8 == 2 ? ScatterLds<16, TYU>::FLOATS : k8 == 3 ? ScatterLds<24, TYU>::FLOATS : ScatterLds<32, TYU>::FLOATS);
    static bool attr = false;
    if (!attr) {   // more than 64 KB of dynamic LDS for 17..32 classes
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&upce_bwd<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * ScatterLds<24, TYU>::FLOATS));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&upce_bwd<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * ScatterLds<32, TYU>::FLOATS));
        attr = true;
    }
    serial_scatter(grid, [&](dim3 gr, int bx0, int by0) {
        UpGeom gg = g;
        gg.bx0 = bx0; gg.by0 = by0;
        switch (k8) {
            case 1: hipLaunchKernelGGL(upce_bwd<1>, gr, dim3(TPB), lds, s, lg, gg, K, Kp, labels, weights, ignore, ce_map, sel3, gout, dlogits); break;
            case 2: hipLaunchKernelGGL(upce_bwd<2>, gr, dim3(TPB), lds, s, lg, gg, K, Kp, labels, weights, ignore, ce_map, sel3, gout, dlogits); break;
            case 3: hipLaunchKernelGGL(upce_bwd<3>, gr, dim3(TPB), lds, s, lg, gg, K, Kp, labels, weights, ignore, ce_map, sel3, gout, dlogits); break;
            default: hipLaunchKernelGGL(upce_bwd<4>, gr, dim3(TPB), lds, s, lg, gg, K, Kp, labels, weights, ignore, ce_map, sel3, gout, dlogits); break;
        }
    });
    if (footprints) launch_gather(g, dlogits, Kp, s);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

#ifndef MGN_F16
int mgn_ohem_select_workspace_bytes(long n, size_t* bytes) {
    if (!bytes || n < 1) return MGN_EINVAL;
    *bytes = sizeof(OhemState) + 256 * sizeof(unsigned) + sizeof(float) * 3 * 1024;
    return MGN_OK;
}
#endif

#ifndef MGN_F16
int mgn_ohem_select(const float* ce_map, long n, const float* sums3, float thr, long n_sel, int force_topk, float* sel3, float* loss,
                    void* workspace, size_t workspace_bytes, void* stream) {
    if (!ce_map || !sums3 || !sel3 || !loss || !workspace || n < 1 || n_sel < 1) return MGN_EINVAL;
    if (workspace_bytes < sizeof(OhemState) + 256 * sizeof(unsigned) + sizeof(float) * 3 * 1024) return MGN_ENOSPC;
    hipStream_t s = (hipStream_t)stream;
    OhemState* st = (OhemState*)workspace;
    unsigned* hist = (unsigned*)(st + 1);
    float* partials = (float*)(hist + 256);
    const int nblk = (int)(n / (TPB * 16) < 1 ? 1 : (n / (TPB * 16) > 1024 ? 1024 : n / (TPB * 16)));
    hipLaunchKernelGGL(ohem_init, dim3(1), dim3(256), 0, s, sums3, n, n_sel, force_topk, st, hist);
    for (int pass = 3; pass >= 0; --pass) {
        hipLaunchKernelGGL(ohem_hist, dim3(nblk), dim3(TPB), 0, s, ce_map, n, pass, (const OhemState*)st, hist);
        hipLaunchKernelGGL(ohem_pick, dim3(1), dim3(256), 0, s, st, hist);
    }
    hipLaunchKernelGGL(ohem_tail, dim3(nblk), dim3(TPB), 0, s, ce_map, n, (const OhemState*)st, partials);
    hipLaunchKernelGGL(ohem_final, dim3(1), dim3(TPB), 0, s, sums3, thr, n_sel, (const OhemState*)st, (const float*)partials, nblk, sel3, loss);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
#endif

int MGN_SYM(mgn_ins_loss_fwd)(const float* center_lr, long csb, long csh, long csw, const void* offset_lr_bf16, long osb, long osh, long osw,
                     int B, int h, int w, int H, int W, const float* ct, const float* cw, const float* ot, const float* ow,
                     float oscale, float* partials, float* out4, void* stream) {
    if (!center_lr || !offset_lr_bf16 || !ct || !cw || !ot || !ow || !partials || !out4 || !geom_ok(B, h, w, H, W)) return MGN_EINVAL;
    InsMaps m;
    m.center = nullptr; m.center_f = center_lr; m.offset = (const uint16_t*)offset_lr_bf16;
    m.gc = make_geom(B, h, w, H, W, csb, csh, csw);
    m.go = make_geom(B, h, w, H, W, osb, osh, osw);
    m.ct = ct; m.cw = cw; m.ot = ot; m.ow = ow; m.oscale = oscale;
    const dim3 grid((W + 63) / 64, (H + 3) / 4, B);
    hipLaunchKernelGGL(ins_fwd, grid, dim3(TPB), 0, (hipStream_t)stream, m, partials);
    hipLaunchKernelGGL(sum4_kernel, dim3(1), dim3(SUMT), 0, (hipStream_t)stream, partials, (int)(grid.x * grid.y * grid.z), out4);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int MGN_SYM(mgn_ins_loss_bwd)(const float* center_lr, long csb, long csh, long csw, const void* offset_lr_bf16, long osb, long osh, long osw,
                     int B, int h, int w, int H, int W, const float* ct, const float* cw, const float* ot, const float* ow,
                     float oscale, const float* out4, const float* gout2, float* dco, float* footprints, void* stream) {
    if (!center_lr || !offset_lr_bf16 || !ct || !cw || !ot || !ow || !out4 || !gout2 || !dco || !geom_ok(B, h, w, H, W)) return MGN_EINVAL;
    if (!footprint_ok(h, w, H, W)) return MGN_ENOTSUP;
    InsMaps m;
    m.center = nullptr; m.center_f = center_lr; m.offset = (const uint16_t*)offset_lr_bf16;
    m.gc = make_geom(B, h, w, H, W, csb, csh, csw);
    m.gc.foot = footprints; m.gc.kt = table_channels(3);
    m.go = make_geom(B, h, w, H, W, osb, osh, osw);
    m.ct = ct; m.cw = cw; m.ot = ot; m.ow = ow; m.oscale = oscale;
    const dim3 grid((W + TX - 1) / TX, (H + TYS - 1) / TYS, B);
    serial_scatter(grid, [&](dim3 gr, int bx0, int by0) {
        InsMaps mm = m;
        mm.gc.bx0 = bx0; mm.gc.by0 = by0;
        hipLaunchKernelGGL(ins_bwd, gr, dim3(TPB), 0, (hipStream_t)stream, mm, out4, gout2, dco);
    });
    if (footprints) launch_gather(m.gc, dco, 4, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

#ifndef MGN_F16
int mgn_upsample1_fwd(const float* lr, int B, int h, int w, int H, int W, float* out, void* stream) {
    if (!lr || !out || !geom_ok(B, h, w, H, W)) return MGN_EINVAL;
    const UpGeom g = make_geom(B, h, w, H, W, (long)h * w, w, 1);
    hipLaunchKernelGGL(up1_fwd, dim3((W + 63) / 64, (H + 3) / 4, B), dim3(TPB), 0, (hipStream_t)stream, lr, g, out);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
#endif

#ifndef MGN_F16
int mgn_upsample1_bwd(const float* dfull, int B, int h, int w, int H, int W, float* dlr_zeroed, float* footprints, void* stream) {
    if (!dfull || !dlr_zeroed || !geom_ok(B, h, w, H, W)) return MGN_EINVAL;
    if (!footprint_ok(h, w, H, W)) return MGN_ENOTSUP;
    UpGeom g = make_geom(B, h, w, H, W, (long)h * w, w, 1);
    g.foot = footprints;
    serial_scatter(dim3((W + TX - 1) / TX, (H + TYS - 1) / TYS, B), [&](dim3 gr, int bx0, int by0) {
        UpGeom gg = g;
        gg.bx0 = bx0; gg.by0 = by0;
        hipLaunchKernelGGL(up1_bwd, gr, dim3(TPB), 0, (hipStream_t)stream, dfull, gg, dlr_zeroed);
    });
    if (footprints) launch_gather(g, dlr_zeroed, 1, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

/* floats of the footprint table of the reproducible backward of which = 0: mgn_upce_bwd (channels = K), 1: mgn_ins_loss_bwd (3),
 * 2: mgn_upsample1_bwd (1) for this geometry (the kernels use different tile heights) */
int mgn_adjoint_footprint_floats(int which, int B, int h, int w, int H, int W, int channels, size_t* floats) {
    if (!floats || !geom_ok(B, h, w, H, W) || channels < 1 || which < 0 || which > 2) return MGN_EINVAL;
    if (!footprint_ok(h, w, H, W)) return MGN_ENOTSUP;
    *floats = footprint_floats(B, h, w, H, W, table_channels(channels), which == 0 ? TYU : TYS);
    return MGN_OK;
}
#endif

}  // extern "C"

// conv.hip -- convolutions of the MGNet trunk as implicit GEMM on the bf16 matrix cores of gfx950.
//
// Replaces torch.nn.functional.conv2d -> cuDNN/MIOpen behind detectron2.layers.Conv2d / nn.Conv2d in
// mgnet/modeling/res_net.py:28-60,96-104 and layers.py:53-72,110-118,146-149,201-210,234-256,283-311 (84 convs; 1x1,
// 3x3 stride 1/2; the 7x7 stems with 3/9 input channels stay on the staging path).
//
// Layout: activations NHWC (torch channels_last) bf16, weights [Cout][KH][KW][Cin] bf16 (K = tap-major, channel-minor),
// fp32 accumulation in the MFMA accumulators.  A 1x1 conv is the plain GEMM [N*H*W, Cin] x [Cin, Cout]; a 3x3 conv is
// the same contraction with K = 9*Cin: the A operand is gathered tap by tap with zero fill for the padding.
//
//   forward / data-gradient : conv_igemm   C[m, co] = sum_{tap, ci} In[pix(m, tap), ci] * W[co, tap, ci]
//        (data gradient = the same kernel over dOut with flipped/transposed weights; stride-2 layers use `up`:
//         a tap contributes only where (o + k - pad) is divisible by the forward stride)
//   weight gradient          : conv_wgrad   dW[co, tap, ci] = sum_m dOut[m, co] * In[pix(m, tap), ci]
//        (K = pixels; split over blockIdx.z, fp32 atomics into the [Cout][KH][KW][Cin] fp32 gradient)
//
// Tiling (wave64, v_mfma_f32_32x32x16_bf16): block 128(M) x 128(N) x 32(K), 4 wavefronts as 2x2, each 64x64 =
// 2x2 MFMA tiles -> 8 MFMAs per wavefront per k-step; LDS tiles [row][32 k] with an 80-byte row pitch (conflict-free
// ds_read_b128 fragment reads); register-staged double buffering (global loads of step s+1 in flight during the MFMAs
// of step s), one barrier per k-step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mgnet_hip.h"

namespace {

#include "h16.h"
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BM = 128;

struct ConvParams {
    const uint16_t* in;   // [N, IH, IW, Cin] bf16
    const uint16_t* w;    // [Cout, KH, KW, Cin] bf16
    void* out;            // [N, OH, OW, Cout] bf16 or fp32
    const float* bias;    // [Cout] or null
    const uint16_t* residual;  // [N, OH, OW, Cout] bf16 added to the result before rounding, or null
    int N, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, up, relu, out_f32;   // relu: 0 none, 1 ReLU, 2 leaky ReLU with `slope`
    float slope;
    int xcd_bands;        // 1: remap blockIdx.x so that every XCD works on one contiguous band of pixel tiles (see xcd_tile)
    float* stat_part;     // [pixel tiles][Cout][2]: per-tile sums of r, r^2 over the ROUNDED outputs (statistics of the InPlaceABNSync that
                          // follows; only without bias / ReLU / residual / fp32 output), or null
};
MGN_PLAN_RO_CONV(ConvParams, MGN_RO(in) MGN_RO(w) MGN_RO(bias) MGN_RO(residual))   // pointers the kernels only read through (launch-plan dependency analysis, csrc/mgn_launch.h)

// Workgroups are dealt to the 8 XCDs round-robin (block b -> XCD b % 8, observed; MI355X_MICROARCH.md) and every XCD has its
// own 4 MB L2.  With tile = blockIdx.x each L2 ends up loading (nearly) the whole input: vertically adjacent tiles share
// their 3x3 halo rows but sit on different XCDs.  Remapped, XCD k owns the contiguous band of tiles
// [start_k, start_k + cnt_k): the rows a band needs are fetched into ONE L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_tile(int b, int nb) {
    const int k = b & 7, j = b >> 3, q = nb >> 3, r = nb & 7;
    return k * q + (k < r ? k : r) + j;
}

// Parity class of the output pixels of an `up`-strided gather (data gradient of a stride-`up` conv).  Class (a, b) =
// (oh % up, ow % up) only meets the taps kh = k0h + up*i, kw = k0w + up*i, k0 = (pad - a) mod up; up == 1: one class,
// all taps.  pixel(): class-local pixel index -> linear output pixel.
struct UpClass {
    int a, b, OHc, OWc, k0h, k0w, nkh, nkw;
    __device__ __forceinline__ UpClass(const ConvParams& p, int cz) {
        a = cz / p.up; b = cz - a * p.up;
        OHc = (p.OH - a + p.up - 1) / p.up; OWc = (p.OW - b + p.up - 1) / p.up;
        if (OHc < 0) OHc = 0;
        if (OWc < 0) OWc = 0;
        const int ra = ((p.pad - a * p.stride) % p.up + p.up) % p.up, rb = ((p.pad - b * p.stride) % p.up + p.up) % p.up;
        k0h = ra; k0w = rb;
        nkh = k0h < p.KH ? (p.KH - k0h + p.up - 1) / p.up : 0;
        nkw = k0w < p.KW ? (p.KW - k0w + p.up - 1) / p.up : 0;
    }
    __device__ __forceinline__ long pixel(const ConvParams& p, long mc) const {
        if (p.up == 1) return mc;
        const int n = (int)(mc / ((long)OHc * OWc));
        const int rem = (int)(mc - (long)n * OHc * OWc);
        const int ohc = rem / OWc, owc = rem - ohc * OWc;
        return ((long)n * p.OH + a + ohc * p.up) * p.OW + b + owc * p.up;
    }
};

__device__ __forceinline__ uint16_t f2bf(float f) { return (uint16_t)mgn_f2h(f); }   // 16-bit activation format of this TU (h16.h)

// epilogue of the implicit-GEMM kernels: 4 consecutive output channels of one pixel (+ bias, + residual, ReLU)
__device__ __forceinline__ void emit4(const ConvParams& p, long m, int co, float a0, float a1, float a2, float a3, bool vec_ok) {
    float v[4] = {a0, a1, a2, a3};
    if (p.residual && vec_ok) {  // fused accumulation of a second gradient branch (bf16, same layout as the output)
        const uint2 r = *reinterpret_cast<const uint2*>(p.residual + m * p.Cout + co);
        v[0] += mgn_lo2f(r.x); v[1] += mgn_hi2f(r.x);
        v[2] += mgn_lo2f(r.y); v[3] += mgn_hi2f(r.y);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] += (p.bias && co + e < p.Cout) ? p.bias[co + e] : 0.f;
        if (p.residual && !vec_ok && co + e < p.Cout) v[e] += mgn_h2f(p.residual[m * p.Cout + co + e]);
        if (p.relu == 1) v[e] = fmaxf(v[e], 0.f);
        else if (p.relu == 2) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
    }
    if (vec_ok) {
        if (p.out_f32)
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + m * p.Cout + co) = make_float4(v[0], v[1], v[2], v[3]);
        else
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + m * p.Cout + co) =
                make_uint2((uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16), (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16));
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (co + e >= p.Cout) break;
            if (p.out_f32) reinterpret_cast<float*>(p.out)[m * p.Cout + co + e] = v[e];
            else reinterpret_cast<uint16_t*>(p.out)[m * p.Cout + co + e] = f2bf(v[e]);
        }
    }
}


// Statistics epilogue of the implicit-GEMM kernels (D = W-rows x pixels: lane & 31 = pixel, registers = channels 8q + 4(lane>>5) + e of
// the wave's NJ 32-channel tiles).  A lane has summed r and r^2 of its channels over its pixels; 16-lane DPP butterflies (quad_perm
// [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror) give every lane its row's sum, lanes 0/16/32/48 park it in LDS, and after a block
// barrier one thread per (channel, moment) adds the WM*2 parts in a fixed order and stores the tile's partial row.
template <int NJ>
__device__ __forceinline__ void stats_flush(float (&s1)[NJ][4][4], float (&s2)[NJ][4][4], float* red, int wm, int WM, int ch_wave, int BNch,
                                            int lane, int tid, float* dst, int nvalid_ch) {
    auto row_sum = [](float v) {
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
        return v;
    };
    __syncthreads();   // every wave has left the k loop: the tile memory is free
    const int rw = (lane >> 4) & 1, hi = lane >> 5;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = row_sum(s1[j][q][e]), b = row_sum(s2[j][q][e]);
                if ((lane & 15) == 0)
                    *reinterpret_cast<float2*>(red + ((wm * 2 + rw) * BNch + ch_wave + j * 32 + 8 * q + 4 * hi + e) * 2) = make_float2(a, b);
            }
    __syncthreads();
    for (int t = tid; t < BNch * 2; t += blockDim.x) {
        if ((t >> 1) >= nvalid_ch) continue;
        float v = 0.f;
        for (int k = 0; k < WM * 2; ++k) v += red[k * BNch * 2 + t];
        dst[t] = v;
    }
}

// XOR swizzle of the 16-byte slot inside a 128-byte LDS row (found by exhaustive search: conflict-free for both the
// 8-lane ds_write_b128 groups of the transposing loader and the 16-lane ds_read_b128 groups of the fragment reads)
__device__ __forceinline__ int swz(int row) { return ((row >> 3) & 1) | (((row >> 4) & 1) << 1) | (((row ^ (row >> 1) ^ (row >> 5)) & 1) << 2); }

// one k-slab of KS*16 reduction elements: wave tile (32*MT) x (32*NT); LDS rows are LPITCH bytes apart
// SWAP: D = B-rows x A-rows (the accumulator's 4 consecutive registers then run along the B-row index)
template <int MT, int NT, int KS, int LPITCH, bool SWZ, bool SWAP = false>
__device__ __forceinline__ void mma_tile(const unsigned char* sA, const unsigned char* sB, int wm, int wn, int lane,
                                         f32x16 (&acc)[MT][NT]) {
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const int slot = kk * 2 + (lane >> 5);
        h16x8 a[MT], b[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = wm * 32 * MT + i * 32 + (lane & 31);
            a[i] = *reinterpret_cast<const h16x8*>(sA + row * LPITCH + ((SWZ ? (slot ^ (LPITCH == 64 ? ((row >> 2) & 3) : swz(row))) : slot) << 4));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int row = wn * 32 * NT + j * 32 + (lane & 31);
            b[j] = *reinterpret_cast<const h16x8*>(sB + row * LPITCH + ((SWZ ? (slot ^ (LPITCH == 64 ? ((row >> 2) & 3) : swz(row))) : slot) << 4));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[i][j] = SWAP ? MGN_MFMA_32x32x16(b[j], a[i], acc[i][j])
                                 : MGN_MFMA_32x32x16(a[i], b[j], acc[i][j]);
    }
}

// KS = k-slab in units of 16 channels (BK = 16*KS input channels of one tap per k-step); LDS rows are padded by 16 bytes
// PACK (small Cin = 8 or 16, the 7x7 stems): the reduction index is k = tap*Cin + c with several taps per 32-wide
// k-slab; every 16-byte loader segment then belongs to its own tap.  Weights are [Cout][Kpad], Kpad = ksteps*32.
template <int NT, int KS, bool PACK = false>
__global__ __launch_bounds__(256) void conv_igemm(ConvParams p) {
    constexpr int BN = 64 * NT, BK = 16 * KS, PITCH = BK * 2, TILE_BYTES = 128 * PITCH;  // XOR-swizzled rows, no padding
    constexpr int SEGS = BK / 8;          // 16-byte segments per row
    constexpr int RPP = 256 / SEGS;       // rows covered per loader pass
    constexpr int NR = 128 / RPP;         // loader passes (rows per thread)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int bm = blockIdx.x, bn = blockIdx.y;
    const long M = (long)p.N * p.OH * p.OW;
    const int lrow = tid / SEGS, seg = tid % SEGS;  // loader: rows lrow + r*RPP, 16-byte segment `seg` of the k-slab

    // Buffer descriptors: an out-of-range voffset makes the hardware return 0, which realises the zero padding and the
    // M / Cout tails without divergent branches around the loads.
    const uint32_t in_bytes = (uint32_t)((size_t)p.N * p.IH * p.IW * p.Cin * 2), w_bytes = (uint32_t)((size_t)p.Cout * (PACK ? ((p.KH * p.KW * p.Cin + BK - 1) / BK) * BK : p.KH * p.KW * p.Cin) * 2);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.in), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0, w_bytes, 0x00020000);
    constexpr int OOB = (int)0x80000000;

    int ihb[NR], iwb[NR], abase[NR], wbase[NR];
    bool vm[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const long m = (long)bm * BM + lrow + r * RPP;
        vm[r] = m < M;
        const long mm = vm[r] ? m : 0;
        const int n = (int)(mm / ((long)p.OH * p.OW));
        const int rem = (int)(mm - (long)n * p.OH * p.OW);
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        ihb[r] = oh * p.stride - p.pad;
        iwb[r] = ow * p.stride - p.pad;
        abase[r] = (n * p.IH * p.IW * p.Cin + (PACK ? 0 : seg * 8)) * 2;  // byte offset of image n (+ the 16-byte segment)
        const int co = bn * BN + lrow + r * RPP;
        const int wrow = PACK ? ((p.KH * p.KW * p.Cin + BK - 1) / BK) * BK : p.KH * p.KW * p.Cin;
        wbase[r] = (co < p.Cout && lrow + r * RPP < BN) ? (co * wrow + seg * 8) * 2 : OOB;
    }
    const int cpt = PACK ? 1 : p.Cin / BK;  // k-steps per tap
    const int ksteps = PACK ? (p.KH * p.KW * p.Cin + BK - 1) / BK : p.KH * p.KW * cpt;
    int ksl = 0;  // PACK: k-step being loaded

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra[NR], rb[NR];
    int kh = 0, kw = 0, cc = 0;  // state of the NEXT k-step to load
    int avoff[NR], wvoff[NR];    // per-tap byte offsets (OOB when the tap falls into the padding)
    auto set_tap = [&]() {
        int wtap = (kh * p.KW + kw) * p.Cin * 2;
        int tkh = kh, tkw = kw, coff = 0;
        bool tap_ok = true;
        if (PACK) {  // this thread's segment of the k-slab: k = ksl*BK + seg*8 -> (tap, channel offset)
            const int k = ksl * BK + seg * 8;
            const int tap = k / p.Cin;
            coff = (k - tap * p.Cin) * 2;
            tkh = tap / p.KW;
            tkw = tap - tkh * p.KW;
            tap_ok = tap < p.KH * p.KW;
            wtap = ksl * BK * 2;
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            int th = ihb[r] + tkh, tw = iwb[r] + tkw;
            bool ok = vm[r] && tap_ok;
            if (p.up > 1) {
                ok = ok && th >= 0 && tw >= 0 && (th % p.up == 0) && (tw % p.up == 0);
                th /= p.up;
                tw /= p.up;
            }
            ok = ok && th >= 0 && th < p.IH && tw >= 0 && tw < p.IW;
            avoff[r] = ok ? abase[r] + (th * p.IW + tw) * p.Cin * 2 + coff : OOB;
            wvoff[r] = wbase[r] == OOB ? OOB : wbase[r] + wtap;
        }
    };
    set_tap();
    auto load_next = [&]() {
        const int soff = cc * BK * 2;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            ra[r] = __builtin_amdgcn_raw_buffer_load_b128(rsA, avoff[r], soff, 0);
            rb[r] = __builtin_amdgcn_raw_buffer_load_b128(rsB, wvoff[r], soff, 0);
        }
        if (PACK) {
            ++ksl;
            set_tap();
        } else if (++cc == cpt) {
            cc = 0;
            if (++kw == p.KW) { kw = 0; ++kh; }
            set_tap();
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int row = lrow + r * RPP;
            const int sw = (KS == 2 ? (seg ^ ((row >> 2) & 3)) : (seg ^ swz(row))) * 16;
            *reinterpret_cast<u32x4*>(&smem[buf][0][row * PITCH + sw]) = ra[r];
            if (row < BN) *reinterpret_cast<u32x4*>(&smem[buf][1][row * PITCH + sw]) = rb[r];
        }
    };

    load_next();
    store_tile(0);
    __syncthreads();
    for (int ks = 0; ks < ksteps; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < ksteps) load_next();
        mma_tile<2, NT, KS, PITCH, true, true>(smem[buf][0], smem[buf][1], wm, wn, lane, acc);
        if (ks + 1 < ksteps) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue.  Operands are swapped in the MFMA (D = W-rows x pixel-rows), so in the C/D layout
    //   col = lane & 31 -> pixel,  row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) -> output channel:
    // the 4 registers e..e+3 are 4 CONSECUTIVE channels of one pixel = one 8-byte (bf16) / 16-byte (fp32) store.
    const bool vec_ok = (p.Cout % 4) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long m = (long)bm * BM + wm * 64 + i * 32 + (lane & 31);
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = bn * BN + wn * 32 * NT + j * 32 + 8 * q + 4 * (lane >> 5);
                if (co >= p.Cout) continue;
                emit4(p, m, co, acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3], vec_ok);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// conv_igemm_glds: same tiling, but the tiles go HBM/L2 -> LDS directly (buffer_load_dwordx4 ... lds, no VGPR staging)
// into a ring of three LDS buffers, so the loads of k-step s+2 are in flight while k-step s is on the matrix cores
// (counted s_waitcnt vmcnt + raw s_barrier; guide: "glds spanning the barrier").  LDS-DMA writes lane-linearly
// (wave base + lane*16), which is exactly the unpadded [row][64 B] tile; the XOR swizzle is therefore applied to the
// SOURCE segment each lane fetches.
// ---------------------------------------------------------------------------------------------------------------
// PACK (the 7x7 stems, Cin = 8 | 16 channel-padded inputs): k = tap * Cin + c, weights [Cout][Kpad]; every 16-byte DMA segment
// of a k-step belongs to its own tap (Cin = 8: four taps per 32-wide step; Cin = 16: two), so the per-lane gather address is
// recomputed per step from the segment's tap -- the same ring / MFMA code otherwise.
template <int NT, bool PACK = false>
__global__ __launch_bounds__(256) void conv_igemm_glds(ConvParams p) {
    constexpr int BN = 64 * NT, BK = 32, PITCH = 64, TILE = 128 * PITCH;  // 8 KB per operand tile
    constexpr int LPT = 2 + NT;  // LDS-DMA instructions per thread and tile (2 A passes + NT B passes)
    __shared__ __attribute__((aligned(16))) unsigned char smem[3][2][TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int bm = p.xcd_bands ? xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x, bn = blockIdx.y;
    // up > 1 (data gradient of a strided conv): blockIdx.z is the parity class (a, b) of the output pixels; only the
    // taps kh = k0h + up*i meet a non-zero of the zero-upsampled gradient, so each class is a dense conv over its taps
    const UpClass uc(p, blockIdx.z);
    const long M = (long)p.N * uc.OHc * uc.OWc;
    if ((long)bm * BM >= M) return;
    const int lrow = tid >> 2, seg = tid & 3;

    const int kpad = PACK ? (p.KH * p.KW * p.Cin + 31) / 32 * 32 : p.KH * p.KW * p.Cin;   // weight row length
    const uint32_t in_bytes = (uint32_t)((size_t)p.N * p.IH * p.IW * p.Cin * 2), w_bytes = (uint32_t)((size_t)p.Cout * kpad * 2);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.in), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0, w_bytes, 0x00020000);
    constexpr int OOB = (int)0x80000000;

    int ihb[2], iwb[2], abase[2], wbase[2], ptap[2], pcoff[2];
    bool vm[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = lrow + r * 64;
        const int sseg = seg ^ ((row >> 2) & 3);  // source segment that lands in LDS slot `seg` of this row
        // PACK: tap of this segment inside a k-step and its byte offset inside the tap's channels
        ptap[r] = p.Cin == 8 ? sseg : (sseg >> 1);
        pcoff[r] = p.Cin == 8 ? 0 : (sseg & 1) * 16;
        const long m = (long)bm * BM + row;
        vm[r] = m < M;
        const long mm = vm[r] ? m : 0;
        const int n = (int)(mm / ((long)uc.OHc * uc.OWc));
        const int rem = (int)(mm - (long)n * uc.OHc * uc.OWc);
        const int ohc = rem / uc.OWc, owc = rem - ohc * uc.OWc;
        ihb[r] = (uc.a + ohc * p.up) * p.stride - p.pad;
        iwb[r] = (uc.b + owc * p.up) * p.stride - p.pad;
        abase[r] = PACK ? n * p.IH * p.IW * p.Cin * 2 : (n * p.IH * p.IW * p.Cin + sseg * 8) * 2;
        const int co = bn * BN + row;
        wbase[r] = (co < p.Cout && row < BN) ? (co * kpad + sseg * 8) * 2 : OOB;
    }
    const int cpt = PACK ? 1 : p.Cin / BK;
    const int ksteps = PACK ? kpad / BK : uc.nkh * uc.nkw * cpt;
    const int tps = PACK ? BK / p.Cin : 1, ntaps = p.KH * p.KW;   // taps per k-step (PACK)

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    int khi = 0, kwi = 0, cc = 0;
    int avoff[2], wvoff[2];
    auto set_tap = [&]() {
        if (PACK) {   // cc = k-step: segment's tap = cc * tps + ptap
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int tap = cc * tps + ptap[r];
                const int kh = tap / p.KW, kw = tap - kh * p.KW;
                const int th = ihb[r] + kh, tw = iwb[r] + kw;
                const bool ok = vm[r] && tap < ntaps && th >= 0 && tw >= 0 && th < p.IH && tw < p.IW;
                avoff[r] = ok ? abase[r] + (th * p.IW + tw) * p.Cin * 2 + pcoff[r] : OOB;
                wvoff[r] = wbase[r];
            }
            return;
        }
        const int kh = uc.k0h + khi * p.up, kw = uc.k0w + kwi * p.up;
        const int wtap = (kh * p.KW + kw) * p.Cin * 2;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int th = ihb[r] + kh, tw = iwb[r] + kw;
            bool ok = vm[r] && th >= 0 && tw >= 0;
            if (p.up > 1) {  // divisible by construction of the class
                th /= p.up;
                tw /= p.up;
            }
            ok = ok && th < p.IH && tw < p.IW;
            avoff[r] = ok ? abase[r] + (th * p.IW + tw) * p.Cin * 2 : OOB;
            wvoff[r] = wbase[r] == OOB ? OOB : wbase[r] + wtap;
        }
    };
    set_tap();
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto issue = [&](int buf) {  // this wave's 1 KB pieces: rows [wave*16, wave*16+16) (+64) of the A and B tiles
        const int soff = cc * BK * 2;
        unsigned char* a0 = &smem[buf][0][wave * 1024];
        unsigned char* b0 = &smem[buf][1][wave * 1024];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)a0, 16, avoff[0], PACK ? 0 : soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a0 + 4096), 16, avoff[1], PACK ? 0 : soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)b0, 16, wvoff[0], soff, 0, 0);
        if (NT == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(b0 + 4096), 16, wvoff[1], soff, 0, 0);
        if (PACK) { ++cc; set_tap(); return; }
        if (++cc == cpt) {
            cc = 0;
            if (++kwi == uc.nkw) { kwi = 0; ++khi; }
            set_tap();
        }
    };

    if (ksteps > 0) issue(0);
    if (ksteps > 1) issue(1);
    int buf = 0;
    for (int ks = 0; ks < ksteps; ++ks) {
        if (ks + 1 < ksteps) {
            if (LPT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (ks + 2 < ksteps) issue(buf == 0 ? 2 : buf - 1);  // (ks+2) % 3
        mma_tile<2, NT, 2, PITCH, true, true>(smem[buf][0], smem[buf][1], wm, wn, lane, acc);
        buf = buf == 2 ? 0 : buf + 1;
    }

    const bool vec_ok = (p.Cout % 4) == 0;
    float st1[NT][4][4], st2[NT][4][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) st1[j][q][e] = st2[j][q][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long mc = (long)bm * BM + wm * 64 + i * 32 + (lane & 31);
        if (mc >= M) continue;
        const long m = uc.pixel(p, mc);   // linear output pixel (n*OH + oh)*OW + ow
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = bn * BN + wn * 32 * NT + j * 32 + 8 * q + 4 * (lane >> 5);
                if (co >= p.Cout) continue;
                emit4(p, m, co, acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3], vec_ok);
                if (p.stat_part) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float r = mgn_h2f(f2bf(acc[i][j][q * 4 + e]));
                        st1[j][q][e] += r;
                        st2[j][q][e] = fmaf(r, r, st2[j][q][e]);
                    }
                }
            }
    }
    if (p.stat_part)
        stats_flush<NT>(st1, st2, reinterpret_cast<float*>(&smem[0][0][0]), wm, 2, wn * 32 * NT, BN, lane, tid,
                        p.stat_part + ((size_t)bm * p.Cout + (size_t)bn * BN) * 2, p.Cout - bn * BN);
}


// ---------------------------------------------------------------------------------------------------------------
// conv1x1_stream: 1x1 convolutions (stride 1 or 2, no padding) as a STREAMING GEMM.  These layers are memory-bound (the
// 256 -> 256 one at 1/8 resolution moves 268 MB for 34 GFLOP) and the generic implicit-GEMM kernel runs them at ~2 TB/s: it
// fetches each 512-byte pixel row in eight 64-byte k-slices and re-reads the weight tile from L2 for every block tile
// (as many bytes again as the activations).  Here
//   * the block is persistent (one per CU) and keeps its slice of the weights in REGISTERS for the whole launch
//     (wave (wn) owns 32*NT output channels: NT * CIN/16 MFMA fragments, <= 128 VGPRs);
//   * pixel rows travel whole: BM x CIN tiles HBM/L2 -> LDS by LDS-DMA (16 B per lane, consecutive lanes = consecutive
//     bytes of a row, XOR-swizzled by the SOURCE chunk so that the fragment reads are conflict-free), double-buffered with
//     counted vmcnt waits so that the next tile's loads and the previous tile's stores stay in flight during the MFMAs;
//   * D = W-rows x pixels (lane = pixel) with the v_permlane32_swap epilogue: 16-byte stores of 8 consecutive channels.
// Grid: (min(tiles, 256), Cout / (32*NT*WN)).  No bias / ReLU / residual / fp32 output (those layers keep the generic kernel).
struct Conv1Params {
    const uint16_t* in;   // [N, IH, IW, CIN]   (in2 != null: channels [0, CIN/2) only, [N, IH, IW, CIN/2])
    const uint16_t* w;    // [Cout, CIN]
    uint16_t* out;        // [N, OH, OW, Cout]  (out2 != null: channels [0, Cout/2) only, [N, OH, OW, Cout/2])
    int N, IH, IW, OH, OW, Cout, stride;
    long M;               // N * OH * OW
    int ntiles;
    const uint16_t* in2;  // the input is the channel concatenation (in | in2) of two maps, never materialised (FeatureFusionModule)
    uint16_t* out2;       // the output's upper half of the channels goes to a map of its own (the concatenation's data gradient)
};
MGN_PLAN_RO_CONV(Conv1Params, MGN_RO(in) MGN_RO(w) MGN_RO(in2))   // pointers the kernels only read through (launch-plan dependency analysis, csrc/mgn_launch.h)

template <int CIN, int NT, int WN>
struct C1 {
    static constexpr int WM = 4 / WN;                               // waves along the pixels
    static constexpr int BM = CIN >= 512 ? 64 : (CIN >= 256 ? 128 : 256);
    static constexpr int PITCH = CIN * 2;                           // bytes per pixel row in LDS
    static constexpr int TILE = BM * PITCH;
    static constexpr int LDS = 2 * TILE;
    static constexpr int NP = TILE / 1024;                          // DMA instructions (1 KB each) per tile
    static constexpr int NPW = NP / 4;                              // per wave
    static constexpr int PG = BM / WM / 32;                         // 32-pixel groups per wave
    static constexpr int ST = PG * NT * 2;                          // 16-byte stores per wave and tile
    static constexpr int KK = CIN / 16;
    static constexpr int CH = PITCH / 16;                           // 16-byte chunks per row
    static_assert(NP % 4 == 0 && NT * KK * 4 <= 128 && ST + NPW <= 63, "conv1x1_stream configuration");
    // rows whose pitch is a multiple of 256 B alias in the LDS banks: XOR the chunk index with the row; shorter rows need a
    // coarser row index (two 128-byte rows / four 64-byte rows share one 256-byte bank cycle)
    static __device__ __forceinline__ int swz(int r) { return PITCH >= 256 ? (r & 7) : (PITCH == 128 ? ((r >> 1) & 7) : ((r >> 2) & 3)); }
};

// TWO: the input rows come from two maps of CIN / 2 channels each (p.in, p.in2).  The LDS tile is then [2 halves][BM rows][CIN bytes]
// (a 1-KB DMA piece stays inside one half, so the source is fixed per piece at compile time); k-step kk reads half kk / (KK / 2).
template <int CIN, int NT, int WN, bool TWO = false>
__device__ __forceinline__ void conv1x1_body(const Conv1Params& p) {
    using C = C1<CIN, NT, WN>;
    constexpr int HP = C::PITCH / 2, HTILE = C::BM * HP;   // TWO: bytes per half row / per half tile
    static_assert(!TWO || (HP >= 256 && C::NP % 8 == 0), "two-source rows: 256-byte halves, whole pieces per half");
    extern __shared__ __attribute__((aligned(16))) unsigned char c1sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5;
    const int wn = wave % WN, wm = wave / WN;
    const int co_w = (blockIdx.y * WN + wn) * 32 * NT;   // this wave's first output channel

    // weights -> registers: fragment (t, kk) = W[co_w + 32 t + (lane & 31)][16 kk + 8 hi .. +7]
    h16x8 wr[NT][C::KK];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint16_t* wp = p.w + (size_t)(co_w + 32 * t + (lane & 31)) * CIN + 8 * hi;
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) wr[t][kk] = *reinterpret_cast<const h16x8*>(wp + kk * 16);
    }

    const uint32_t in_bytes = (uint32_t)((size_t)p.N * p.IH * p.IW * (TWO ? CIN / 2 : CIN) * 2);
    const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.in), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsI2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(TWO ? p.in2 : p.in), 0, in_bytes, 0x00020000);
    constexpr int OOB = (int)0x80000000;
    typedef __attribute__((address_space(3))) void* lds_ptr;

    // DMA piece j (1 KB) of a tile: LDS bytes [1024 j, 1024 j + 1024); lane l writes 16 bytes at 1024 j + 16 l = row r, physical
    // chunk pc; it fetches the logical chunk pc ^ swz(r) of pixel (tile base + r)
    auto issue = [&](int tile, int buf) {
        unsigned char* base = c1sm + buf * C::TILE;
#pragma unroll
        for (int i = 0; i < C::NPW; ++i) {
            const int j = wave + 4 * i;
            constexpr int PITCH_L = TWO ? HP : C::PITCH;                 // row pitch of the LDS (half) tile this piece lands in
            const bool second = TWO && i >= C::NPW / 2;                  // (j >= NP / 2 <=> i >= NPW / 2: wave < 4)
            const int o = (TWO ? (j - (second ? C::NP / 2 : 0)) : j) * 1024 + lane * 16;
            const int r = o / PITCH_L, pc = (o % PITCH_L) >> 4;
            const long m = (long)tile * C::BM + r;
            int voff = OOB;
            if (m < p.M) {
                long pix = m;
                if (p.stride != 1) {   // output pixel -> input pixel (n, stride * oh, stride * ow)
                    const int ow = (int)(m % p.OW);
                    const long t2 = m / p.OW;
                    const int oh = (int)(t2 % p.OH), n = (int)(t2 / p.OH);
                    pix = ((long)n * p.IH + (long)oh * p.stride) * p.IW + (long)ow * p.stride;
                }
                voff = (int)(pix * PITCH_L) + ((pc ^ C::swz(r)) << 4);
            }
            if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsI2, (lds_ptr)(base + j * 1024), 16, voff, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsI, (lds_ptr)(base + j * 1024), 16, voff, 0, 0, 0);
        }
    };

    int tile = blockIdx.x, buf = 0;
    if (tile >= p.ntiles) return;
    issue(tile, 0);
    bool first = true;
    for (; tile < p.ntiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        const bool has_next = next < p.ntiles;
        if (has_next) issue(next, buf ^ 1);
        // this tile's loads must have landed; allowed in flight: the next tile's loads and the previous tile's stores
        if (has_next) {
            if (first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NPW + C::ST) : "memory");
        } else {
            if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::ST) : "memory");
        }
        first = false;
        __builtin_amdgcn_s_barrier();
        const unsigned char* a = c1sm + buf * C::TILE;
#pragma unroll
        for (int g = 0; g < C::PG; ++g) {
            const int r = (wm * C::PG + g) * 32 + (lane & 31);
            const unsigned char* row = a + r * (TWO ? HP : C::PITCH);
            const int sw = C::swz(r);
            f32x16 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) {
                const h16x8 x = TWO ? *reinterpret_cast<const h16x8*>(row + (kk >= C::KK / 2 ? HTILE : 0) + ((((2 * kk + hi) & (HP / 16 - 1)) ^ sw) << 4))
                                    : *reinterpret_cast<const h16x8*>(row + (((2 * kk + hi) ^ sw) << 4));
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = MGN_MFMA_32x32x16(wr[t][kk], x, acc[t]);
            }
            // D = W-rows x pixels: col = lane & 31 -> pixel, row = (e & 3) + 8 (e >> 2) + 4 hi -> channel; lane pairs swap 4-channel
            // groups so that each lane stores 8 consecutive channels (16 bytes)
            const long m = (long)tile * C::BM + r;
            const bool ok = m < p.M;
            // (out2: this wave's channels lie in one half of the output channels -- 32 NT divides Cout / 2)
            const int ohalf = p.Cout >> 1;
            uint16_t* opix = !p.out2 ? p.out + (size_t)(ok ? m : 0) * p.Cout + co_w
                                     : (co_w >= ohalf ? p.out2 + (size_t)(ok ? m : 0) * ohalf + (co_w - ohalf) : p.out + (size_t)(ok ? m : 0) * ohalf + co_w);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) {
                    uint32_t pk[2][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int qd = 2 * qp + u;
                        pk[u][0] = (uint32_t)f2bf(acc[t][qd * 4 + 0]) | ((uint32_t)f2bf(acc[t][qd * 4 + 1]) << 16);
                        pk[u][1] = (uint32_t)f2bf(acc[t][qd * 4 + 2]) | ((uint32_t)f2bf(acc[t][qd * 4 + 3]) << 16);
                    }
                    const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                    // (every lane issues the store so that the vmcnt bookkeeping is uniform; rows past M write nothing: exec mask)
                    if (ok) *reinterpret_cast<uint4*>(opix + 32 * t + 16 * qp + 8 * hi) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
        }
        __builtin_amdgcn_s_barrier();   // all fragment reads of this buffer are done before the tile after next lands in it
        buf ^= 1;
    }
}
__global__ __launch_bounds__(256, 1) void conv1x1_s_256_2_4(Conv1Params p) { conv1x1_body<256, 2, 4>(p); }
__global__ __launch_bounds__(256, 1) void conv1x1_s_256_2_4_cat(Conv1Params p) { conv1x1_body<256, 2, 4, true>(p); }
__global__ __launch_bounds__(256, 1) void conv1x1_s_256_1_1(Conv1Params p) { conv1x1_body<256, 1, 1>(p); }
__global__ __launch_bounds__(256, 1) void conv1x1_s_32_2_4(Conv1Params p) { conv1x1_body<32, 2, 4>(p); }
__global__ __launch_bounds__(256, 1) void conv1x1_s_64_1_4(Conv1Params p) { conv1x1_body<64, 1, 4>(p); }
__global__ __launch_bounds__(256, 1) void conv1x1_s_128_2_4(Conv1Params p) { conv1x1_body<128, 2, 4>(p); }
__global__ __launch_bounds__(256, 1) void conv1x1_s_512_1_4(Conv1Params p) { conv1x1_body<512, 1, 4>(p); }

template <int CIN, int NT, int WN, typename K>
static int launch_conv1x1(K kernel, Conv1Params& q, hipStream_t st, bool plan_only = false) {
    using C = C1<CIN, NT, WN>;
    // per KERNEL, not per instantiation of this function: conv1x1_s_256_2_4 and its two-source twin share <256, 2, 4, K>
    static const void* attr_done[4] = {nullptr, nullptr, nullptr, nullptr};
    const void* kp = reinterpret_cast<const void*>(kernel);
    bool seen = false;
    for (const void* d : attr_done) seen = seen || d == kp;
    if (!seen) {
        (void)hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        for (const void*& d : attr_done)
            if (!d) { d = kp; break; }
    }
    const long tiles = (q.M + C::BM - 1) / C::BM;
    if (tiles > 0x7fffffffL) return MGN_EINVAL;
    q.ntiles = (int)tiles;
    const int gy = q.Cout / (32 * NT * WN);
    // the register-resident weights pay off over several tiles per block; small layers (one tile per CU or less) are
    // faster on the generic kernel (measured 19.7 vs 17.9 us for 512 -> 256 at 32 x 64)
    if (tiles * gy < 512 && !getenv("MGN_CONV_FORCE1X1")) return 1;
    if (plan_only) return 2;
    const int gx = (int)(tiles < 256 ? tiles : 256);   // persistent: one block per CU
    hipLaunchKernelGGL(kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), C::LDS, st, q);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------
// conv_igemm_big: the same LDS-DMA pipeline with a 256(M) x BN(N) block tile, BN = 64*NWN (128 or 256), 2 x NWN waves of
// 128 x 64 each (4 x 2 MFMA tiles: 6 fragment reads per 8 MFMAs instead of 8, half the L2->LDS bytes per flop of the
// 128 x 128 tile).  BK = 32, ring of three 16 KB + BN*64 B stages, two k-steps in flight across the raw barrier.
// Used when the layer has enough 256-pixel tiles to fill the chip (host dispatch).
// ---------------------------------------------------------------------------------------------------------------
template <int NWN, int NWM = 2>
struct IgemmBig {
    static constexpr int BMB = 128 * NWM, BN = 64 * NWN, NW = NWM * NWN, THREADS = 64 * NW;
    static constexpr int ATILE = BMB * 64, BTILE = BN * 64, STAGE = ATILE + BTILE, LDS = 3 * STAGE;
    static constexpr int PA = (BMB / 16) / NW, PB = (BN / 16) / NW;   // 1-KB DMA pieces (16 rows x 64 B) per wave and stage
};

template <int NWN, int NWM = 2>
__device__ __forceinline__ void igemm_big_body(const ConvParams& p) {
    using C = IgemmBig<NWN, NWM>;
    constexpr int PA = C::PA, PB = C::PB;
    extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / NWN, wn = wave % NWN;
    const int bm = p.xcd_bands ? xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x, bn = blockIdx.y;
    const UpClass uc(p, blockIdx.z);
    const long M = (long)p.N * uc.OHc * uc.OWc;
    if ((long)bm * C::BMB >= M) return;
    const int lrow = lane >> 2, seg = lane & 3;   // row inside a 16-row piece, 16-byte slot

    const uint32_t in_bytes = (uint32_t)((size_t)p.N * p.IH * p.IW * p.Cin * 2), w_bytes = (uint32_t)((size_t)p.Cout * p.KH * p.KW * p.Cin * 2);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.in), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0, w_bytes, 0x00020000);
    constexpr int OOB = (int)0x80000000;

    // piece q of wave w covers tile rows (w + NW*q)*16 .. +16
    int ihb[PA], iwb[PA], abase[PA], wbase[PB];
    bool vm[PA];
#pragma unroll
    for (int q = 0; q < PA; ++q) {
        const int row = (wave + C::NW * q) * 16 + lrow;
        const int sseg = seg ^ ((row >> 2) & 3);
        const long m = (long)bm * C::BMB + row;
        vm[q] = m < M;
        const long mm = vm[q] ? m : 0;
        const int n = (int)(mm / ((long)uc.OHc * uc.OWc));
        const int rem = (int)(mm - (long)n * uc.OHc * uc.OWc);
        const int ohc = rem / uc.OWc, owc = rem - ohc * uc.OWc;
        ihb[q] = (uc.a + ohc * p.up) * p.stride - p.pad;
        iwb[q] = (uc.b + owc * p.up) * p.stride - p.pad;
        abase[q] = (n * p.IH * p.IW * p.Cin + sseg * 8) * 2;
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int row = (wave + C::NW * q) * 16 + lrow;
        const int sseg = seg ^ ((row >> 2) & 3);
        const int co = bn * C::BN + row;
        wbase[q] = co < p.Cout ? (co * p.KH * p.KW * p.Cin + sseg * 8) * 2 : OOB;
    }
    const int cpt = p.Cin / 32;
    const int ksteps = uc.nkh * uc.nkw * cpt;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    int khi = 0, kwi = 0, cc = 0;
    int avoff[PA], wtap = 0;
    auto set_tap = [&]() {
        const int kh = uc.k0h + khi * p.up, kw = uc.k0w + kwi * p.up;
        wtap = (kh * p.KW + kw) * p.Cin * 2;
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            int th = ihb[q] + kh, tw = iwb[q] + kw;
            bool ok = vm[q] && th >= 0 && tw >= 0;
            if (p.up > 1) {
                th /= p.up;
                tw /= p.up;
            }
            ok = ok && th < p.IH && tw < p.IW;
            avoff[q] = ok ? abase[q] + (th * p.IW + tw) * p.Cin * 2 : OOB;
        }
    };
    set_tap();
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto issue = [&](int buf) {
        const int soff = cc * 64;
        unsigned char* a0 = bsm + buf * C::STAGE;
        unsigned char* b0 = a0 + C::ATILE;
#pragma unroll
        for (int q = 0; q < PA; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a0 + (wave + C::NW * q) * 1024), 16, avoff[q], soff, 0, 0);
#pragma unroll
        for (int q = 0; q < PB; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(b0 + (wave + C::NW * q) * 1024), 16, wbase[q] == OOB ? OOB : wbase[q] + wtap, soff, 0, 0);
        if (++cc == cpt) {
            cc = 0;
            if (++kwi == uc.nkw) { kwi = 0; ++khi; }
            set_tap();
        }
    };

    if (ksteps > 0) issue(0);
    if (ksteps > 1) issue(1);
    int buf = 0;
    for (int ks = 0; ks < ksteps; ++ks) {
        if (ks + 1 < ksteps) {
            if (PA + PB == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (PA + PB == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (ks + 2 < ksteps) issue(buf == 0 ? 2 : buf - 1);  // (ks+2) % 3
        const unsigned char* sA = bsm + buf * C::STAGE;
        mma_tile<4, 2, 2, 64, true, true>(sA, sA + C::ATILE, wm, wn, lane, acc);
        buf = buf == 2 ? 0 : buf + 1;
    }

    const bool vec_ok = (p.Cout % 4) == 0;
    float st1[2][4][4], st2[2][4][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) st1[j][q][e] = st2[j][q][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long mc = (long)bm * C::BMB + wm * 128 + i * 32 + (lane & 31);
        if (mc >= M) continue;
        const long m = uc.pixel(p, mc);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = bn * C::BN + wn * 64 + j * 32 + 8 * q + 4 * (lane >> 5);
                if (co >= p.Cout) continue;
                emit4(p, m, co, acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3], vec_ok);
                if (p.stat_part) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float r = mgn_h2f(f2bf(acc[i][j][q * 4 + e]));
                        st1[j][q][e] += r;
                        st2[j][q][e] = fmaf(r, r, st2[j][q][e]);
                    }
                }
            }
    }
    if (p.stat_part)
        stats_flush<2>(st1, st2, reinterpret_cast<float*>(bsm), wm, NWM, wn * 64, C::BN, lane, tid,
                       p.stat_part + ((size_t)bm * p.Cout + (size_t)bn * C::BN) * 2, p.Cout - bn * C::BN);
}
__global__ __launch_bounds__(256, 2) void conv_igemm_big128(ConvParams p) { igemm_big_body<2>(p); }
__global__ __launch_bounds__(512, 1) void conv_igemm_big256(ConvParams p) { igemm_big_body<4>(p); }
__global__ __launch_bounds__(512, 1) void conv_igemm_big512x128(ConvParams p) { igemm_big_body<2, 4>(p); }

// ---------------------------------------------------------------------------------------------------------------
// conv3x3_c64: 3x3 / stride 1 / pad 1 convolution with 64 input channels (ResNet layer1 forward and data gradient, 16
// launches per step), row march with the WEIGHTS IN REGISTERS.
//
// With only 64 output channels per tile the implicit-GEMM kernels are bound by staging the gathered A operand (every input
// element is fetched 9 times, once per tap, and used 64 times).  Here a block owns a 128-pixel strip of one image and
// marches down its rows: input rows live in an LDS ring ([pixel][64 ch], filled by LDS-DMA, zero padding from the
// buffer bounds), a tap is an address offset into the ring (each input row is fetched from HBM once), and every wave
// keeps its 32 output channels x 576 reduction elements of the weights in 144 VGPRs for the whole march, so the only
// LDS traffic is one ds_read_b128 per MFMA.  8 waves = 4 pixel groups x 2 channel halves; 36 MFMAs per wave and row;
// rows are prefetched two steps ahead (ring of 5), output rows are stored straight from the accumulators.
// ---------------------------------------------------------------------------------------------------------------
struct Conv64Params {
    const uint16_t* in;    // [N, H, W, 64] bf16
    const uint16_t* w;     // [Cout, 3, 3, 64] bf16
    uint16_t* out;         // [N, H, W, Cout] bf16
    const uint16_t* residual;  // like ConvParams::residual
    float* stat_part;          // [nslices][Cout][2] per-slice sums of r, r^2 over the ROUNDED outputs (the statistics pass of the
                               // InPlaceABNSync that follows, see conv_win.hip / mgn_iabn_coeffs_from_partials), or null
    int N, H, W, Cout;
    int strips, chunks, rows_per_chunk, nslices, co_tiles;
    const float* bias;     // ACT kernels: out = act(conv + bias + residual), act 0 none | 1 ReLU | 2 leaky ReLU with `slope` (the eval-mode
    int act;               // fold of the InPlaceABNSync that follows, mgn_conv_igemm_act)
    float slope;
};
MGN_PLAN_RO_CONV(Conv64Params, MGN_RO(in) MGN_RO(w) MGN_RO(residual) MGN_RO(bias))   // pointers the kernels only read through (launch-plan dependency analysis, csrc/mgn_launch.h)
constexpr int C64_NR = 8;   // ring of input rows: r-1 .. r+2 in use by the two output rows of a step, r+3 .. r+6 in flight
constexpr int C64_INROW = 136 * 128, C64_LDS = C64_NR * C64_INROW;

template <bool STATS, bool RES, bool ACT = false>
__device__ __forceinline__ void conv3x3_c64_body(const Conv64Params& p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char c64sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wpx = wave >> 1, wco = wave & 1;
    const int L = blockIdx.x, q = L >> 3;
    const int slice = (q / p.co_tiles) * 8 + (L & 7), tile = q % p.co_tiles;
    if (slice >= p.nslices) return;
    int s = slice;
    // strips fastest: blocks launched together walk the strips of the SAME image rows, so the rows stream from DRAM whole
    const int strip = s % p.strips; s /= p.strips;
    const int chunk = s % p.chunks, n = s / p.chunks;
    const int ow0 = strip * 128;
    const int r0 = chunk * p.rows_per_chunk, r1 = r0 + p.rows_per_chunk < p.H ? r0 + p.rows_per_chunk : p.H;
    const int co_w = tile * 64 + wco * 32;   // this wave's 32 output channels

    // weights -> registers: fragment (tap, kk) = W[co_w + (lane & 31)][tap][kk*16 + 8*(lane>>5) .. +7]
    h16x8 wr[36];
    {
        const uint16_t* wp = p.w + ((size_t)(co_w + (lane & 31)) * 9) * 64 + 8 * (lane >> 5);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) wr[t * 4 + kk] = *reinterpret_cast<const h16x8*>(wp + t * 64 + kk * 16);
    }

    const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.in), 0, (uint32_t)((size_t)p.N * p.H * p.W * 64 * 2), 0x00020000);
    constexpr int OOB = (int)0x80000000;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // LDS-DMA: 17 one-KB pieces (8 pixels x 128 B) per input row, all issued by the loader waves (below)
    // Roles (round 4): the two waves of a SIMD (w and w + 4) never issue memory instructions at the same time.  Waves 0-3 COMPUTE
    // FIRST: the step's 72 MFMAs, then the stores of its two output rows.  Waves 4-7 LOAD FIRST: the stores of their PREVIOUS step's
    // rows, then ALL 34 LDS-DMA pieces of the two input rows fetched ahead, then their MFMAs -- which run while the partner stores and
    // waits at the next barrier.  Before, every wave issued its share of the loads behind the barrier and its stores in front of the
    // next one: ~2200 cycles of vector-memory issue per step (a 1-KB piece or store occupies the CU's memory path ~34 cycles) during
    // which no wave fed the matrix pipes (68 % longer than the same loop without its stores).
    const int grp = wave >> 2;
    int voffL[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int j = (wave & 3) + 4 * i;
        const int px = 8 * j + (lane >> 3);
        const int seg = (lane & 7) ^ ((px >> 1) & 7);
        const int iw = ow0 - 1 + px;
        voffL[i] = (j < 17 && iw >= 0 && iw < p.W) ? (iw * 64 + seg * 8) * 2 : OOB;
    }
    auto issue_in_loader = [&](int ih, int slot) {
        const bool ok = ih >= 0 && ih < p.H;
        const int soff = ok ? ((n * p.H + ih) * p.W) * 128 : 0;
        unsigned char* base = c64sm + slot * C64_INROW;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int j = (wave & 3) + 4 * i;
            if (j < 17) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsI, (lds_ptr)(base + j * 1024), 16, ok ? voffL[i] : OOB, soff, 0, 0);
        }
    };

    // A-fragment byte offset inside an input row for (kw, kk): patch pixel px = 32*wpx + (lane & 31) + kw, 16-byte slot (2*kk + lane>>5)
    // XOR-swizzled by the pixel = aoffk[kw] ^ (kk << 5)  (one register per kw: px * 128 has no bits below 128)
    const int hi = lane >> 5;
    int aoffk[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int px = 32 * wpx + (lane & 31) + kw;
        aoffk[kw] = px * 128 + ((hi ^ ((px >> 1) & 7)) << 4);
    }
    float st1[4][4], st2[4][4];   // statistics of this lane's 16 channels over its pixels (only with p.stat_part)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) st1[q][e] = st2[q][e] = 0.f;
    // TWO output rows per step (72 MFMAs per wave between barriers; the weight registers serve both rows)
    if (grp == 1) {   // rows r0-1 .. r0+4: the first step's four rows + one step ahead
#pragma unroll
        for (int k = 0; k < 6; ++k) issue_in_loader(r0 - 1 + k, k);
    }
    int si = 0;  // slot of input row r-1
    const int opx = ow0 + 32 * wpx + (lane & 31);
    auto slot = [](int x) { return x >= C64_NR ? x - C64_NR : x; };
    f32x16 acc0, acc1;   // output rows r and r+1
    // the residual (fused skip gradient) of a step's two rows is fetched BEFORE the step's MFMAs, in the layout of the 16-byte stores:
    // its HBM latency then passes under the matrix work instead of in front of the stores
    uint4 resv[2][2];
    auto load_residual = [&](int r) {
        if (opx < p.W) {
#pragma unroll
            for (int rr2 = 0; rr2 < 2; ++rr2) {
                const int rr = r + rr2 < r1 ? r + rr2 : r;   // (odd number of rows: the last step's second row is not stored)
                const uint16_t* rpix = p.residual + ((size_t)(n * p.H + rr) * p.W + opx) * p.Cout + co_w;
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) resv[rr2][qp] = *reinterpret_cast<const uint4*>(rpix + 16 * qp + 8 * hi);
            }
        }
    };
    float bv[4][4];   // ACT: this lane's 16 bias values (channel = co_w + 8 qd + 4 hi + e)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ACT && p.bias) t = *reinterpret_cast<const float4*>(p.bias + co_w + 8 * qd + 4 * hi);
        bv[qd][0] = t.x; bv[qd][1] = t.y; bv[qd][2] = t.z; bv[qd][3] = t.w;
    }
    auto store_rows = [&](int r) {
        // D = W-rows x pixels: col = lane & 31 -> pixel, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) -> output channel.
        // v_permlane32_swap exchanges the 4-channel groups between lane l and lane l+32 so that every lane owns 8 CONSECUTIVE
        // channels of its pixel: 16-byte stores, half as many store instructions (pack8x2 below).
        if (opx < p.W) {
#pragma unroll
            for (int rr2 = 0; rr2 < 2; ++rr2) {
                if (r + rr2 >= r1) break;   // wave-uniform (odd number of rows: the last step has one row)
                uint16_t* opix = p.out + ((size_t)(n * p.H + r + rr2) * p.W + opx) * p.Cout + co_w;
                {
#pragma unroll
                    for (int qp = 0; qp < 2; ++qp) {   // channel groups (qd = 2qp, 2qp+1) -> channels 16*qp + 8*hi .. +7
                        // optional residual (the fused skip gradient): read with ONE 16-byte load in the layout of the store below and
                        // brought back to the accumulator layout by the same two v_permlane32_swap (the exchange is its own inverse);
                        // added in fp32 before the rounding, as the 8-byte form it replaces (which cost 4x the memory instructions)
                        uint32_t rp[2][2] = {{0u, 0u}, {0u, 0u}};
                        if (RES) {
                            const uint4 R = resv[rr2][qp];
                            const auto u0 = __builtin_amdgcn_permlane32_swap(R.x, R.z, false, false);
                            const auto u1 = __builtin_amdgcn_permlane32_swap(R.y, R.w, false, false);
                            rp[0][0] = u0[0]; rp[1][0] = u0[1]; rp[0][1] = u1[0]; rp[1][1] = u1[1];
                        }
                        uint32_t pk[2][2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int qd = 2 * qp + u;
                            float v0 = rr2 ? acc1[qd * 4 + 0] : acc0[qd * 4 + 0], v1 = rr2 ? acc1[qd * 4 + 1] : acc0[qd * 4 + 1];
                            float v2 = rr2 ? acc1[qd * 4 + 2] : acc0[qd * 4 + 2], v3 = rr2 ? acc1[qd * 4 + 3] : acc0[qd * 4 + 3];
                            if (ACT) { v0 += bv[qd][0]; v1 += bv[qd][1]; v2 += bv[qd][2]; v3 += bv[qd][3]; }
                            if (RES) {
                                v0 += mgn_lo2f(rp[u][0]); v1 += mgn_hi2f(rp[u][0]);
                                v2 += mgn_lo2f(rp[u][1]); v3 += mgn_hi2f(rp[u][1]);
                            }
                            if (ACT && p.act) {
                                const float sl = p.act == 1 ? 0.f : p.slope;
                                v0 = v0 > 0.f ? v0 : v0 * sl; v1 = v1 > 0.f ? v1 : v1 * sl; v2 = v2 > 0.f ? v2 : v2 * sl; v3 = v3 > 0.f ? v3 : v3 * sl;
                            }
                            pk[u][0] = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
                            pk[u][1] = (uint32_t)f2bf(v2) | ((uint32_t)f2bf(v3) << 16);
                            if (STATS) {   // (every lane in here owns a valid pixel)
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    const float d0 = mgn_lo2f(pk[u][h]), d1 = mgn_hi2f(pk[u][h]);
                                    st1[qd][2 * h] += d0; st2[qd][2 * h] = fmaf(d0, d0, st2[qd][2 * h]);
                                    st1[qd][2 * h + 1] += d1; st2[qd][2 * h + 1] = fmaf(d1, d1, st2[qd][2 * h + 1]);
                                }
                            }
                        }
                        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                        *reinterpret_cast<uint4*>(opix + 16 * qp + 8 * hi) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                    }
                }
            }
        }
    };
    for (int r = r0; r < r1; r += 2) {
        // Rows up to r+2 must have landed: the loader waves issued them two steps ago (or in the prologue), and the only LOADS they have
        // issued since are the two rows of the previous step (>= 8 pieces per wave) and, on the residual path, its 4 residual loads (stores
        // in between can only make the wait longer).  The other waves issue no LDS-DMA.
        if (grp == 1) {
            if (RES) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // (+ the 4 residual loads of the previous step)
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (grp == 1) {
            if (r > r0) store_rows(r - 2);
            issue_in_loader(r + 5, slot(si + 6));   // (always issued: rows past the chunk land in unused slots)
            issue_in_loader(r + 6, slot(si + 7));
        }
        if (RES) load_residual(r);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const unsigned char* rowa = c64sm + slot(si + kh) * C64_INROW;
            const unsigned char* rowb = c64sm + slot(si + kh + 1) * C64_INROW;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const h16x8 a = *reinterpret_cast<const h16x8*>(rowa + (aoffk[kw] ^ (kk << 5)));
                    const h16x8 b = *reinterpret_cast<const h16x8*>(rowb + (aoffk[kw] ^ (kk << 5)));
                    const int idx = (kh * 3 + kw) * 4 + kk;
                    acc0 = MGN_MFMA_32x32x16(wr[idx], a, acc0);
                    acc1 = MGN_MFMA_32x32x16(wr[idx], b, acc1);
                }
            }
        }
        if (grp == 0) store_rows(r);
        si = slot(si + 2);
    }
    if (grp == 1) store_rows(r0 + ((r1 - r0 - 1) & ~1));   // the loader waves' last step
    if (STATS) {
        // 16-lane DPP butterflies, the 8 parts of a channel (4 pixel groups x 2 lane rows) through LDS, one partial row per slice
        auto row_sum = [](float v) {
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
            return v;
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // rows prefetched past the chunk still land in the ring
        __syncthreads();
        float* red = reinterpret_cast<float*>(c64sm);       // [wpx 4][lane row 2][64 channels][2]
        const int rw = (lane >> 4) & 1;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = row_sum(st1[q][e]), b = row_sum(st2[q][e]);
                if ((lane & 15) == 0)
                    *reinterpret_cast<float2*>(red + ((wpx * 2 + rw) * 64 + wco * 32 + 8 * q + 4 * hi + e) * 2) = make_float2(a, b);
            }
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k * 128 + tid];
            p.stat_part[((size_t)slice * p.Cout + tile * 64) * 2 + tid] = t;
        }
    }
}

__global__ __launch_bounds__(512, 1) void conv3x3_c64(Conv64Params p) { conv3x3_c64_body<false, false>(p); }
__global__ __launch_bounds__(512, 1) void conv3x3_c64_res(Conv64Params p) { conv3x3_c64_body<false, true>(p); }
__global__ __launch_bounds__(512, 1) void conv3x3_c64_stats(Conv64Params p) { conv3x3_c64_body<true, false>(p); }
__global__ __launch_bounds__(512, 1) void conv3x3_c64_act(Conv64Params p) { conv3x3_c64_body<false, false, true>(p); }
__global__ __launch_bounds__(512, 1) void conv3x3_c64_act_res(Conv64Params p) { conv3x3_c64_body<false, true, true>(p); }


// ---------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------
struct WgradParams {
    const uint16_t* dout;  // [N, OH, OW, Cout] bf16
    const uint16_t* in;    // [N, IH, IW, Cin] bf16
    float* dw;             // [Cout, KH, KW, Cin] fp32, accumulated with atomics
    int N, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad;
    int ci_tiles;          // ceil(Cin / (64*NT))
    long m_per_split;      // pixels per blockIdx.z (multiple of 64)
    int oihw, cin_real;    // final layout of dw: [Cout][cin_real][KH][KW] (torch parameter layout) or [Cout][KH][KW][Cin]
    float* partial;        // [gridDim.z][Cout][taps*Cin] per-split partial results (plain stores, no atomics)
    int remap_tiles, co_tiles;   // remap_tiles > 0: 1-D grid, see wgrad_block
    const uint16_t* in2;   // != null (conv_wgrad only): the input is the channel concatenation (in | in2) of two maps of Cin / 2 channels
};
MGN_PLAN_RO_CONV(WgradParams, MGN_RO(dout) MGN_RO(in) MGN_RO(in2))   // pointers the kernels only read through (launch-plan dependency analysis, csrc/mgn_launch.h)

// Which (output-channel tile, tap x input-channel tile, pixel split) a block of the split-K tile kernels works on.  With the plain 3-D
// grid the tile blocks of ONE pixel split -- which read the same dOut / input pixels -- have consecutive linear ids and therefore land
// on different XCDs (block id % 8), each with its own L2: every tile re-reads its operands from HBM / the memory-side cache.  Remapped
// (1-D grid of T x 8 x ceil(splits / 8) blocks), the T tile blocks of a split share an XCD and are dispatched together.
struct WgradBlock { int bco, by, bz; };
__device__ __forceinline__ WgradBlock wgrad_block(const WgradParams& p) {
    if (!p.remap_tiles) return {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    const int h = blockIdx.x, xcd = h & 7, slot = h >> 3, tile = slot % p.remap_tiles;
    return {tile % p.co_tiles, tile / p.co_tiles, (slot / p.remap_tiles) * 8 + xcd};
}

constexpr int WBK = 64;            // pixels per k-step
constexpr int WPITCH = 128;        // 64 pixels x 2 bytes per channel row, XOR-swizzled (no padding)
constexpr int WTILE = 128 * WPITCH;

// 8 pixels x 8 channels (one uint4 per pixel) -> 8 channels x 8 pixels (one uint4 per channel), in registers
__device__ __forceinline__ void transpose8x8(const uint4 (&in)[8], uint4 (&out)[8]) {
    const uint32_t* I = reinterpret_cast<const uint32_t*>(in);   // I[p*4 + d]
    uint32_t* O = reinterpret_cast<uint32_t*>(out);              // O[c*4 + q]
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t x = I[(2 * q) * 4 + d], y = I[(2 * q + 1) * 4 + d];
            O[(2 * d) * 4 + q] = (x & 0xffffu) | (y << 16);
            O[(2 * d + 1) * 4 + q] = (x >> 16) | (y & 0xffff0000u);
        }
}

// Block: (64*MT) output channels x (64*NT) input channels of one tap, a slice of the pixels.  Waves 0,1 stream dOut,
// waves 2,3 stream (gathered) In: each thread loads 8 pixels x 8 channels with 16-byte loads, transposes in registers
// and writes 8 x 16 bytes into the [channel][pixel] LDS tile that the MFMA fragments read.  Pixel coordinates advance
// incrementally (no integer division in the loop).
// PACK (Cin = 8 or 16): the column index of dW is n' = tap*Cin + c, a 64*NT-column tile spans several taps and every
// loader thread owns the tap of its 8-channel segment.
template <int MT, int NT, bool PACK = false>
__global__ __launch_bounds__(256) void conv_wgrad(WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];  // [2 buffers][A | B][WTILE]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const WgradBlock wb = wgrad_block(p);
    const int bco = wb.bco;
    const int tap_blk = PACK ? 0 : wb.by / p.ci_tiles, bci = PACK ? wb.by : wb.by % p.ci_tiles;
    int kh = tap_blk / p.KW, kw = tap_blk % p.KW;
    const long M = (long)p.N * p.OH * p.OW;
    const long m_begin = (long)wb.bz * p.m_per_split;
    const long m_end = m_begin + p.m_per_split < M ? m_begin + p.m_per_split : M;
    if (m_begin >= M) return;
    const bool isB = tid >= 128;
    const int t = tid & 127;
    const int cseg = t & 15, pg = t >> 4;  // 8-channel segment, 8-pixel group
    int c0 = (isB ? bci * 64 * NT : bco * 64 * MT) + cseg * 8;
    bool vc = c0 < (isB ? p.Cin : p.Cout) && cseg * 8 < 64 * (isB ? NT : MT);
    // two-source input (64 NT divides Cin / 2: a block's column tile lies in one of the maps)
    const int cpitch = p.in2 ? p.Cin >> 1 : p.Cin;
    const bool second = p.in2 && bci * 64 * NT >= cpitch;
    if (isB && second) c0 -= cpitch;
    if (PACK && isB) {  // c0 is a packed column: split into (tap, channel)
        const int tap = c0 / p.Cin;
        vc = tap < p.KH * p.KW && cseg * 8 < 64 * NT;
        c0 -= tap * p.Cin;
        kh = tap / p.KW;
        kw = tap - kh * p.KW;
    }
    // coordinates of this thread's first pixel of the current k-step
    long mcur = m_begin + pg * 8;
    int pn = (int)(mcur / ((long)p.OH * p.OW));
    int prem = (int)(mcur - (long)pn * p.OH * p.OW);
    int poh = prem / p.OW, pow_ = prem - poh * p.OW;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.dout), 0, (uint32_t)((size_t)M * p.Cout * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(second ? p.in2 : p.in), 0, (uint32_t)((size_t)p.N * p.IH * p.IW * cpitch * 2), 0x00020000);
    constexpr int OOB = (int)0x80000000;
    uint4 rg[8];
    auto load = [&]() {  // loads the k-step starting at pixel mcur (out-of-range -> 0 via the buffer bounds), advances by WBK
        if (!isB) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int off = (vc && mcur + q < m_end) ? (int)(((mcur + q) * p.Cout + c0) * 2) : OOB;
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
                rg[q] = make_uint4(v.x, v.y, v.z, v.w);
            }
        } else if (pow_ + 8 <= p.OW) {
            // fast path: the 8 pixels lie in one output row -> one row test, offsets differ by a constant step
            const int ih = poh * p.stride - p.pad + kh, iw0 = pow_ * p.stride - p.pad + kw;
            const bool rowok = vc && ih >= 0 && ih < p.IH;
            const int base = (((pn * p.IH + ih) * p.IW + iw0) * cpitch + c0) * 2, step = p.stride * cpitch * 2;
            const int rem = (int)(m_end - mcur < 8 ? m_end - mcur : 8);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int iw = iw0 + q * p.stride;
                const bool ok = rowok && q < rem && iw >= 0 && iw < p.IW;
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsB, ok ? base + q * step : OOB, 0, 0);
                rg[q] = make_uint4(v.x, v.y, v.z, v.w);
            }
        } else {
            int n = pn, oh = poh, ow = pow_;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int ih = oh * p.stride - p.pad + kh, iw = ow * p.stride - p.pad + kw;
                const bool ok = vc && mcur + q < m_end && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW;
                const int off = ok ? (((n * p.IH + ih) * p.IW + iw) * cpitch + c0) * 2 : OOB;
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsB, off, 0, 0);
                rg[q] = make_uint4(v.x, v.y, v.z, v.w);
                const bool wrap = (ow + 1 == p.OW);
                ow = wrap ? 0 : ow + 1;
                const bool wrap2 = wrap && (oh + 1 == p.OH);
                oh = wrap ? (wrap2 ? 0 : oh + 1) : oh;
                n += wrap2 ? 1 : 0;
            }
        }
        if (isB) {
            pow_ += WBK;
            while (pow_ >= p.OW) { pow_ -= p.OW; if (++poh == p.OH) { poh = 0; ++pn; } }
        }
        mcur += WBK;
    };
    auto store = [&](int buf) {
        uint4 tr[8];
        transpose8x8(rg, tr);
        unsigned char* base = wsm + (size_t)(buf * 2 + (isB ? 1 : 0)) * WTILE;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int row = cseg * 8 + e;
            *reinterpret_cast<uint4*>(base + row * WPITCH + ((pg ^ swz(row)) << 4)) = tr[e];
        }
    };

    load();
    store(0);
    __syncthreads();
    int buf = 0;
    for (long m0 = m_begin; m0 < m_end; m0 += WBK) {
        const bool more = m0 + WBK < m_end;
        if (more) load();
        mma_tile<MT, NT, 4, WPITCH, true>(wsm + (size_t)(buf * 2) * WTILE, wsm + (size_t)(buf * 2 + 1) * WTILE, wm, wn, lane, acc);
        if (more) store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    const int ncols = PACK ? p.KH * p.KW * p.Cin : p.Cin;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int ci = bci * 64 * NT + wn * 32 * NT + j * 32 + (lane & 31);
        if (ci >= ncols) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = bco * 64 * MT + wm * 32 * MT + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (co >= p.Cout) continue;
                // per-split partial tile in [Cout][tap][Cin] order: coalesced plain stores; conv_wgrad_reduce sums the splits
                const size_t wsize = (size_t)p.Cout * p.KH * p.KW * p.Cin;
                const size_t idx = PACK ? (size_t)co * ncols + ci : (((size_t)co * p.KH * p.KW + tap_blk) * p.Cin + ci);
                p.partial[(size_t)wb.bz * wsize + idx] = acc[i][j][e];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// conv_wgrad3x3: weight gradient of the 3x3 / stride 1 / pad 1 layers (the bulk of the trunk), all 9 taps in one block.
//
// Block = one 64(co) x 64(ci) tile of dW for ALL 9 taps over a slice of pixels (image n, a 64-column strip, a run of
// rows).  The block marches down its rows; LDS holds a ring of 4 input rows (72 pixels: the strip + halo) and 2 dOut
// rows in the NATURAL NHWC order [pixel][64 channels], filled by LDS-DMA (no VGPR staging, no register transpose);
// the zero padding and every tail come from the buffer bounds check.  The MFMA fragments (K = pixels) are gathered
// with ds_read_b64_tr_b16, the transposing LDS read of gfx950: a 16-lane group reads 4 pixels x 16 channels and every
// lane receives the 4 pixels of its channel.  A tap (kh, kw) is only an address offset of the In fragment (row slot,
// +kw pixels), so dOut and In are read from HBM ONCE for the 9 taps (the per-tap kernel read them 9 times), and the
// dOut fragment is reused by 9 MFMAs.  Each wave owns a 32x32 corner of the tile for the 9 taps (9 accumulators).
// 16-byte slot swizzle: slot ^= 4 * bit1(pixel): any 4 consecutive pixels x 4 slots then cover all 64 banks once.
// Blocks that share a pixel slice (the other co/ci tiles) are placed on the same XCD back to back (shared L2).
// ---------------------------------------------------------------------------------------------------------------
struct Wgrad3Params {
    const uint16_t* dout;  // [N, H, W, Cout] bf16
    const uint16_t* in;    // [N, H, W, Cin] bf16
    float* partial;        // [nslices][Cout][9][Cin]
    int N, H, W, Cin, Cout;
    int co_tiles, ci_tiles, strips, chunks, rows_per_chunk, nslices, ng;
};
MGN_PLAN_RO_CONV(Wgrad3Params, MGN_RO(dout) MGN_RO(in))   // pointers the kernels only read through (launch-plan dependency analysis, csrc/mgn_launch.h)
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ h16x8 tr_frag(const unsigned char* p) {  // 8 k-values (pixels p, p+4 rows apart by 128 B) of this lane's channel
    typedef __attribute__((address_space(3))) s16x4* lp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p + 512));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(h16x8, v);
}

// NG = number of 4-wave groups: the strip is 64*NG pixels wide, group q owns pixels [64q, 64q+64) of every row and the
// groups' accumulators are summed through LDS at the end (half the split partials for NG = 2).
// Rows are prefetched W3_D = 2 steps ahead (ring of 5 In rows and 3 dOut rows) with a counted s_waitcnt.
#ifndef W3_PD
#define W3_PD 4   // fragment prefetch distance of conv_wgrad3x3, in MFMAs
#endif
template <int NG>
struct W3 {
    static constexpr int SW = 64 * NG, INROW = (SW + 8) * 128, OUTROW = SW * 128;
    static constexpr int NIN = 5, NOUT = 3, LDS = NIN * INROW + NOUT * OUTROW;
    static constexpr int PIN = SW / 8 + 1, POUT = SW / 8, NW = 4 * NG;      // 1-KB DMA pieces per row, waves
    static constexpr int IMAX = (PIN + POUT + NW - 1) / NW;                 // pieces per wave and row step (upper bound)
};

template <int NG>
__device__ __forceinline__ void wgrad3x3_body(const Wgrad3Params& p) {
    using C = W3<NG>;
    extern __shared__ __attribute__((aligned(16))) unsigned char w3sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int tiles = p.co_tiles * p.ci_tiles;
    const int L = blockIdx.x, q = L >> 3;
    const int slice = (q / tiles) * 8 + (L & 7), tile = q % tiles;
    if (slice >= p.nslices) return;
    const int co0 = (tile / p.ci_tiles) * 64, ci0 = (tile % p.ci_tiles) * 64;
    int s = slice;
    // strips fastest: blocks launched together walk the strips of the SAME image rows, so the rows stream from DRAM whole
    const int strip = s % p.strips; s /= p.strips;
    const int chunk = s % p.chunks, n = s / p.chunks;
    const int ow0 = strip * C::SW;
    const int r0 = chunk * p.rows_per_chunk, r1 = r0 + p.rows_per_chunk < p.H ? r0 + p.rows_per_chunk : p.H;

    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.dout), 0, (uint32_t)((size_t)p.N * p.H * p.W * p.Cout * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.in), 0, (uint32_t)((size_t)p.N * p.H * p.W * p.Cin * 2), 0x00020000);
    constexpr int OOB = (int)0x80000000;
    typedef __attribute__((address_space(3))) void* lds_ptr;

    // LDS-DMA plan: PIN + POUT one-KB pieces per row step; wave w issues pieces w, w + NW, ...
    int voff[C::IMAX];
#pragma unroll
    for (int i = 0; i < C::IMAX; ++i) {
        const int j = wave + C::NW * i;
        const int px = 8 * (j < C::PIN ? j : j - C::PIN) + (lane >> 3);
        const int seg = (lane & 7) ^ (((px >> 1) & 1) << 2);
        if (j < C::PIN) {
            const int iw = ow0 - 1 + px;
            voff[i] = (iw >= 0 && iw < p.W) ? (iw * p.Cin + ci0 + seg * 8) * 2 : OOB;
        } else {
            const int ow = ow0 + px;
            voff[i] = (j < C::PIN + C::POUT && ow < p.W) ? (ow * p.Cout + co0 + seg * 8) * 2 : OOB;
        }
    }
    auto issue_in = [&](int ih, int slot) {
        const bool ok = ih >= 0 && ih < p.H;
        const int soff = ok ? ((n * p.H + ih) * p.W) * p.Cin * 2 : 0;
        unsigned char* base = w3sm + slot * C::INROW;
#pragma unroll
        for (int i = 0; i < C::IMAX; ++i) {
            const int j = wave + C::NW * i;
            if (j < C::PIN) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsI, (lds_ptr)(base + j * 1024), 16, ok ? voff[i] : OOB, soff, 0, 0);
        }
    };
    auto issue_out = [&](int oh, int slot) {
        const bool ok = oh < p.H;
        const int soff = ok ? ((n * p.H + oh) * p.W) * p.Cout * 2 : 0;
        unsigned char* base = w3sm + C::NIN * C::INROW + slot * C::OUTROW;
#pragma unroll
        for (int i = 0; i < C::IMAX; ++i) {
            const int j = wave + C::NW * i;
            if (j >= C::PIN && j < C::PIN + C::POUT)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, (lds_ptr)(base + (j - C::PIN) * 1024), 16, ok ? voff[i] : OOB, soff, 0, 0);
        }
    };
    // NG = 2 (two waves per SIMD): inside the row loop the waves of a SIMD ALTERNATE between loading and computing.  A 1-KB LDS-DMA piece
    // occupies the CU's vector-memory path for ~34 cycles and the issuing wave with it; with every wave issuing its share behind the
    // barrier (33 pieces per row step) the matrix pipes idled ~1100 of ~3500 cycles per row (measured: the same loop without its DMA
    // runs at 92 % of the MFMA rate, profiles/r04_wgrad3x3_roles.txt).  Now the 4 waves of group (r - r0) & 1 -- one per SIMD -- issue the
    // WHOLE row step (pieces (wave & 3) + 4 i) while their SIMD partners start the step's 36 MFMAs at once, then compute themselves while
    // the partners wait at the next barrier; the roles swap every row.
    constexpr int LMAX = NG == 2 ? (C::PIN + C::POUT + 3) / 4 : 1;
    int voffL[LMAX];
    if (NG == 2) {
#pragma unroll
        for (int i = 0; i < LMAX; ++i) {
            const int j = (wave & 3) + 4 * i;
            const int px = 8 * (j < C::PIN ? j : j - C::PIN) + (lane >> 3);
            const int seg = (lane & 7) ^ (((px >> 1) & 1) << 2);
            if (j < C::PIN) {
                const int iw = ow0 - 1 + px;
                voffL[i] = (iw >= 0 && iw < p.W) ? (iw * p.Cin + ci0 + seg * 8) * 2 : OOB;
            } else {
                const int ow = ow0 + px;
                voffL[i] = (j < C::PIN + C::POUT && ow < p.W) ? (ow * p.Cout + co0 + seg * 8) * 2 : OOB;
            }
        }
    }
    auto issue_step_loader = [&](int ih, int islot, int oh, int oslot) {
        const bool oki = ih >= 0 && ih < p.H, oko = oh < p.H;
        const int soffi = oki ? ((n * p.H + ih) * p.W) * p.Cin * 2 : 0, soffo = oko ? ((n * p.H + oh) * p.W) * p.Cout * 2 : 0;
#pragma unroll
        for (int i = 0; i < LMAX; ++i) {
            const int j = (wave & 3) + 4 * i;
            if (j < C::PIN)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsI, (lds_ptr)(w3sm + islot * C::INROW + j * 1024), 16, oki ? voffL[i] : OOB, soffi, 0, 0);
            else if (j < C::PIN + C::POUT)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, (lds_ptr)(w3sm + C::NIN * C::INROW + oslot * C::OUTROW + (j - C::PIN) * 1024), 16,
                                                         oko ? voffL[i] : OOB, soffo, 0, 0);
        }
    };
    // pieces this wave issues per row step (for the counted wait)
    const int my_cnt = (C::PIN + C::POUT - wave + C::NW - 1) / C::NW;

    // fragment gather addresses (ds_read_b64_tr_b16): lane group g = lane>>4 reads pixels 8*(g>>1) + (i>>2) [+4], channels
    // 16*(g&1) + 4*(i&3) .. +3 of the wave's 32-channel half
    const int g = lane >> 4, i4 = lane & 15;
    const int prow = grp * 64 + 8 * (g >> 1) + (i4 >> 2);
    auto lds_off = [](int px, int ch) { return px * 128 + ((((ch >> 3) ^ (((px >> 1) & 1) << 2))) << 4) + (ch & 7) * 2; };
    const int aA = C::NIN * C::INROW + lds_off(prow, wm * 32 + (g & 1) * 16 + 4 * (i4 & 3));
    int aB[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) aB[kw] = lds_off(prow + kw, wn * 32 + (g & 1) * 16 + 4 * (i4 & 3));

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // ring slots: In row ih lives in slot (ih - r0 + 1) % 5, dOut row oh in slot (oh - r0) % 3
    issue_in(r0 - 1, 0);
    issue_in(r0, 1);
    issue_in(r0 + 1, 2);
    issue_out(r0, 0);
    if (r0 + 1 < r1) {
        issue_in(r0 + 2, 3);
        issue_out(r0 + 1, 1);
    }
    int si = 0, so = 0;  // slots of In row r-1 and dOut row r
    for (int r = r0; r < r1; ++r) {
        const bool pf = r + 2 < r1;   // the rows two steps ahead are loaded during this step
        const int pf_is = si + 4 >= 5 ? si - 1 : si + 4, pf_os = so + 2 >= 3 ? so - 1 : so + 2;
        const bool loader = NG == 2 && grp == ((r - r0) & 1);
        if (NG == 1 || r == r0) {
            // every wave issued the prologue / (NG = 1) issues its share of every step: the step group of row r+1 may stay in flight
            if (r + 1 < r1) {
                if (my_cnt == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else if (my_cnt == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (my_cnt == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else if (loader || r == r0 + 1) {
            // the rows of step r were issued at step r-2 by THIS group (nothing since), or in the prologue by everyone; the other group's
            // step-(r-1) pieces (<= 9 per wave, at least 8) are the only ones that may stay in flight
            if (!loader && r0 + 2 < r1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (NG == 1) {
            if (pf) {
                issue_in(r + 3, pf_is);
                issue_out(r + 2, pf_os);
            }
        } else if (loader && pf) {
            issue_step_loader(r + 3, pf_is, r + 2, pf_os);
        }
        const int s1 = si + 1 >= 5 ? si - 4 : si + 1, s2 = si + 2 >= 5 ? si - 3 : si + 2;
        const unsigned char* sa = w3sm + aA + so * C::OUTROW;
        const unsigned char* sb[3] = {w3sm + si * C::INROW, w3sm + s1 * C::INROW, w3sm + s2 * C::INROW};
        // The 36 MFMAs of a row step (4 k-slabs of 16 pixels x 9 taps) as ONE software-pipelined stream: the In fragment of
        // MFMA m is read W3_PD MFMAs ahead into a register ring, the dOut fragment of a slab W3_PD MFMAs before the slab's
        // first MFMA -- left to itself the scheduler issues each tap's transposing reads right behind the previous tap's MFMA
        // and waits out the LDS latency in front of 18 of the 36 MFMAs.  sched_barrier pins the order; the s_waitcnt lgkmcnt
        // values are the compiler's (LDS reads return in order).
        constexpr int PD = W3_PD, NB = W3_PD + 1;
        h16x8 fa[2], fb[NB];
        auto rd_b = [&](int m) {   // m = 9 * ks + 3 * kh + kw
            const int ks = m / 9, kh = (m % 9) / 3, kw = m % 3;
            fb[m % NB] = tr_frag(sb[kh] + aB[kw] + ks * 2048);
        };
        fa[0] = tr_frag(sa);
#pragma unroll
        for (int m = 0; m < PD; ++m) rd_b(m);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 36; ++m) {
            if (m + PD < 36) {
                if ((m + PD) % 9 == 0) fa[((m + PD) / 9) & 1] = tr_frag(sa + ((m + PD) / 9) * 2048);
                rd_b(m + PD);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[m % 9] = MGN_MFMA_32x32x16(fa[(m / 9) & 1], fb[m % NB], acc[m % 9]);
            __builtin_amdgcn_sched_barrier(0);
        }
        si = s1;
        so = so + 1 >= 3 ? 0 : so + 1;
    }
    if (NG == 2) {  // sum the two pixel groups: 3 taps at a time through the (now idle) ring memory
        float* red = reinterpret_cast<float*>(w3sm);
        const int t256 = tid & 255;
#pragma unroll
        for (int tb = 0; tb < 9; tb += 3) {
            __syncthreads();
            if (grp == 1) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[(t * 16 + e) * 256 + t256] = acc[tb + t][e];
            }
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[tb + t][e] += red[(t * 16 + e) * 256 + t256];
            }
        }
        if (grp == 1) return;
    }
    const size_t wsize = (size_t)p.Cout * 9 * p.Cin;
    float* dst = p.partial + (size_t)slice * wsize;
    const int ci = ci0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            dst[((size_t)co * 9 + t) * p.Cin + ci] = acc[t][e];
        }
}

__global__ __launch_bounds__(256, 2) void conv_wgrad3x3_s64(Wgrad3Params p) { wgrad3x3_body<1>(p); }
__global__ __launch_bounds__(512, 1) void conv_wgrad3x3_s128(Wgrad3Params p) { wgrad3x3_body<2>(p); }

// ---------------------------------------------------------------------------------------------------------------
// conv_wgrad_stem: weight gradient of the 7x7 / stride 2 / pad 3 stems on the channel-padded input (CP = 8 or 16), same
// construction as conv_wgrad3x3: a block marches down a 128-output-pixel strip; the input rows (natural [pixel][CP]
// order, 7 + 4 rows in a ring of 11, two new rows per step) and the dOut rows ([pixel][64], ring of 3) arrive by LDS-DMA,
// fragments are gathered with ds_read_b64_tr_b16 (K = output pixels; the stride-2 gather is only a per-lane address).
// dW columns of one kernel row kh are laid out as [8 kw slots][CP] (slot 7 is padding): a 16-column fragment block is
// then 32 CONTIGUOUS input bytes for either CP, and a kernel row is 4 (CP=16) or 2 (CP=8) 32-column MFMA tiles.
// 8 waves x 7 accumulator tiles: CP=16: 2 co halves x 4 groups of 7 of the 28 column tiles; CP=8: 2 co halves x 2 groups
// of 7 of the 14 column tiles x 2 pixel halves (summed through LDS at the end).
// ---------------------------------------------------------------------------------------------------------------
struct WgradStemParams {
    const uint16_t* dout;  // [N, OH, OW, Cout] bf16
    const uint16_t* in;    // [N, IH, IW, CP] bf16
    float* partial;        // [nslices][Cout][7][8*CP]
    int N, IH, IW, OH, OW, Cout;
    int strips, chunks, rows_per_chunk, nslices, co_tiles;
};
MGN_PLAN_RO_CONV(WgradStemParams, MGN_RO(dout) MGN_RO(in))   // pointers the kernels only read through (launch-plan dependency analysis, csrc/mgn_launch.h)
template <int CP>
struct WS {
    static constexpr int PIN = CP == 16 ? 9 : (CP == 8 ? 5 : 3);      // 1-KB pieces per input row of the strip (262 pixels)
    static constexpr int INROW = PIN * 1024, OUTROW = 128 * 128, POUT = 16;
    static constexpr int NIN = 11, NOUT = 3, LDS = NIN * INROW + NOUT * OUTROW;
    static constexpr int PXG = 16 / CP, TPK = CP / 4;   // pixel groups, 32-column tiles per kernel row
    static constexpr int COL0 = CP == 4 ? 4 : 3;      // image column of slot 0 relative to 2*ow (CP = 4: slot 0 is the padding slot, kw = slot - 1)
    static constexpr int KSW = 8 / PXG;               // 16-pixel k-steps per wave and row
    static constexpr int PXB = CP * 2;                // bytes per input pixel
};

template <int CP>
__device__ __forceinline__ void wgrad_stem_body(const WgradStemParams& p) {
    using C = WS<CP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char wssm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1;
    const int ngrp = CP == 16 ? wave >> 1 : (CP == 8 ? (wave >> 1) & 1 : 0), pxg = CP == 16 ? 0 : (CP == 8 ? wave >> 2 : wave >> 1);
    const int L = blockIdx.x, q8 = L >> 3;
    const int slice = (q8 / p.co_tiles) * 8 + (L & 7), tile = q8 % p.co_tiles;
    if (slice >= p.nslices) return;
    const int co0 = tile * 64;
    int s = slice;
    // strips fastest: blocks launched together walk the strips of the SAME image rows, so the rows stream from DRAM whole
    const int strip = s % p.strips; s /= p.strips;
    const int chunk = s % p.chunks, n = s / p.chunks;
    const int ow0 = strip * 128;
    const int r0 = chunk * p.rows_per_chunk, r1 = r0 + p.rows_per_chunk < p.OH ? r0 + p.rows_per_chunk : p.OH;

    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.dout), 0, (uint32_t)((size_t)p.N * p.OH * p.OW * p.Cout * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.in), 0, (uint32_t)((size_t)p.N * p.IH * p.IW * C::PXB), 0x00020000);
    constexpr int OOB = (int)0x80000000;
    typedef __attribute__((address_space(3))) void* lds_ptr;

    // LDS-DMA plan.  Input row pieces j (1 KB = a contiguous run of the image row, starting at column 2*ow0 - 3) go to
    // wave (j + rot) % 8 (rot = 0 / 4 for the two rows of a step); dOut pieces jj to wave jj % 8.
    int vin[2], vout[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // input piece index for this wave is resolved at issue time (depends on rot); the lane part is fixed:
        vin[i] = 0;
        const int jj = wave + 8 * i;
        const int px = 8 * jj + (lane >> 3);
        const int seg = (lane & 7) ^ (((px >> 1) & 1) << 2);
        const int ow = ow0 + px;
        vout[i] = ow < p.OW ? (ow * p.Cout + co0 + seg * 8) * 2 : OOB;
    }
    (void)vin;
    const int iw_base = 2 * ow0 - C::COL0;
    auto issue_in = [&](int ih, int slot, int rot) {
        const bool ok = ih >= 0 && ih < p.IH;
        const int soff = ok ? ((n * p.IH + ih) * p.IW) * C::PXB : 0;
        unsigned char* base = wssm + slot * C::INROW;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = ((wave - rot) & 7) + 8 * i;
            if (j < C::PIN) {
                const int boff = j * 1024 + lane * 16;
                const int iw = iw_base + boff / C::PXB;
                const int vo = (ok && iw >= 0 && iw < p.IW) ? iw_base * C::PXB + boff : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsI, (lds_ptr)(base + j * 1024), 16, vo, soff, 0, 0);
            }
        }
    };
    auto issue_out = [&](int oh, int slot) {
        const bool ok = oh < p.OH;
        const int soff = ok ? ((n * p.OH + oh) * p.OW) * p.Cout * 2 : 0;
        unsigned char* base = wssm + C::NIN * C::INROW + slot * C::OUTROW;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, (lds_ptr)(base + (wave + 8 * i) * 1024), 16, ok ? vout[i] : OOB, soff, 0, 0);
    };
    int my_cnt = 2;  // pieces per step of this wave: 2 dOut + the input pieces of both rows
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (((wave - 0) & 7) + 8 * i < C::PIN) ++my_cnt;
        if (((wave - 4) & 7) + 8 * i < C::PIN) ++my_cnt;
    }

    // fragment addresses.  A (dOut): pixel rows, co columns.  B (input): rows = output pixels -> input pixel 2*px (+kw
    // through the column block), columns = 32-byte blocks of the kernel row.
    const int g = lane >> 4, i4 = lane & 15;
    const int prow = pxg * (128 / C::PXG) + 8 * (g >> 1) + (i4 >> 2);
    auto lds_off = [](int px, int ch) { return px * 128 + ((((ch >> 3) ^ (((px >> 1) & 1) << 2))) << 4) + (ch & 7) * 2; };
    const int aA = C::NIN * C::INROW + lds_off(prow, wm * 32 + (g & 1) * 16 + 4 * (i4 & 3));
    const int bB = 2 * prow * C::PXB + 32 * (g & 1) + 8 * (i4 & 3);
    typedef __attribute__((address_space(3))) s16x4* lp;

    f32x16 acc[7];
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // Role-alternating waves (round 4, as in wgrad3x3_body): in the row loop the four waves of group (r - r0) & 1 -- one per SIMD -- issue
    // the WHOLE step (two input rows + one dOut row, pieces (wave & 3) + 4 i) while their SIMD partners start the step's MFMAs; a 1-KB
    // LDS-DMA piece occupies the CU's vector-memory path ~34 cycles, and with every wave issuing its share behind the barrier that was
    // 1150 (CP = 16) / 880 (CP = 8) cycles per step without an MFMA in flight, against 3580 / 1790 cycles of MFMA issue.
    const int grp = wave >> 2;
    constexpr int TOT = 2 * C::PIN + C::POUT, LMAX = (TOT + 3) / 4, LMIN = TOT / 4;
    auto issue_step_loader = [&](int iha, int sa, int ihb, int sb, int oh, int oslot) {
#pragma unroll
        for (int i = 0; i < LMAX; ++i) {
            const int jj = (wave & 3) + 4 * i;
            if (jj < 2 * C::PIN) {
                const bool second = jj >= C::PIN;
                const int j = second ? jj - C::PIN : jj, ih = second ? ihb : iha, slot = second ? sb : sa;
                const bool ok = ih >= 0 && ih < p.IH;
                const int soff = ok ? ((n * p.IH + ih) * p.IW) * C::PXB : 0;
                const int boff = j * 1024 + lane * 16;
                const int iw = iw_base + boff / C::PXB;
                const int vo = (ok && iw >= 0 && iw < p.IW) ? iw_base * C::PXB + boff : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsI, (lds_ptr)(wssm + slot * C::INROW + j * 1024), 16, vo, soff, 0, 0);
            } else if (jj < TOT) {
                const int jo = jj - 2 * C::PIN;
                const bool ok = oh < p.OH;
                const int soff = ok ? ((n * p.OH + oh) * p.OW) * p.Cout * 2 : 0;
                const int px = 8 * jo + (lane >> 3);
                const int seg = (lane & 7) ^ (((px >> 1) & 1) << 2);
                const int ow = ow0 + px;
                const int vo = (ok && ow < p.OW) ? (ow * p.Cout + co0 + seg * 8) * 2 : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, (lds_ptr)(wssm + C::NIN * C::INROW + oslot * C::OUTROW + jo * 1024), 16, vo, soff, 0, 0);
            }
        }
    };

    // ring slots: input row ih -> (ih - (2*r0 - 3)) % 11, dOut row oh -> (oh - r0) % 3
    for (int k = 0; k < 7; ++k) issue_in(2 * r0 - 3 + k, k, (k & 1) * 4);
    issue_out(r0, 0);
    issue_in(2 * r0 + 4, 7, 0);
    issue_in(2 * r0 + 5, 8, 4);
    issue_out(r0 + 1, 1);
    int s0 = 0, so = 0;  // slots of input row 2r-3 and dOut row r
    for (int r = r0; r < r1; ++r) {
        const bool loader = grp == ((r - r0) & 1);
        if (r == r0) {
            // the prologue (every wave its share): the group of step r0+1 may stay in flight
            if (my_cnt == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (my_cnt == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if (my_cnt == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (my_cnt == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else if (loader) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this group issued step r's rows two steps ago (or in the prologue), nothing since
        } else if (r == r0 + 1) {
            // the prologue's rows of step r0+1 must have landed; this group's pieces of step r0 (>= LMIN per wave) may stay in flight
            if (LMIN == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (LMIN == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            static_assert(LMIN == 8 || LMIN == 6 || LMIN == 5, "counted wait of the second step");
        }
        __builtin_amdgcn_s_barrier();
        if (loader) {   // rows of step r+2 (always issued: out-of-range rows read as zeros into slots nobody uses)
            const int sa = s0 + 9 >= 11 ? s0 - 2 : s0 + 9, sb = s0 + 10 >= 11 ? s0 - 1 : s0 + 10;
            issue_step_loader(2 * r + 6, sa, 2 * r + 7, sb, r + 2, so + 2 >= 3 ? so - 1 : so + 2);
        }
        const unsigned char* sa_ = wssm + aA + so * C::OUTROW;
        const unsigned char* bp[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int t = ngrp * 7 + i, kh = t / C::TPK, qq = t % C::TPK;
            const int sl = s0 + kh >= 11 ? s0 + kh - 11 : s0 + kh;
            bp[i] = wssm + sl * C::INROW + bB + qq * 64;
        }
        // the KSW x 7 MFMAs of a step as one software-pipelined stream (input fragment of MFMA m read 4 MFMAs ahead)
        constexpr int NM = C::KSW * 7, PD = 4, NB = PD + 1;
        h16x8 fa[2], fb[NB];
        auto rd_b = [&](int m) {
            const int ks = m / 7, i = m % 7;
            const unsigned char* q = bp[i] + ks * (32 * C::PXB);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(q));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(q + 8 * C::PXB));
            fb[m % NB] = __builtin_bit_cast(h16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        fa[0] = tr_frag(sa_);
#pragma unroll
        for (int m = 0; m < PD; ++m) rd_b(m);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (m + PD < NM) {
                if ((m + PD) % 7 == 0) fa[((m + PD) / 7) & 1] = tr_frag(sa_ + ((m + PD) / 7) * 2048);
                rd_b(m + PD);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[m % 7] = MGN_MFMA_32x32x16(fa[(m / 7) & 1], fb[m % NB], acc[m % 7]);
            __builtin_amdgcn_sched_barrier(0);
        }
        s0 = s0 + 2 >= 11 ? s0 - 9 : s0 + 2;
        so = so + 1 >= 3 ? 0 : so + 1;
    }
    if (C::PXG > 1) {  // sum the pixel groups (fixed order 1, 2, ..): 3 tiles at a time through the (now idle) ring memory
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float* red = reinterpret_cast<float*>(wssm);
        constexpr int GT = 512 / C::PXG;          // threads of a pixel group
        const int tg = tid & (GT - 1);
#pragma unroll
        for (int src = 1; src < C::PXG; ++src)
#pragma unroll
            for (int tb = 0; tb < 7; tb += 3) {
                __syncthreads();
                if (pxg == src) {
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        if (tb + t < 7)
#pragma unroll
                            for (int e = 0; e < 16; ++e) red[(t * 16 + e) * GT + tg] = acc[tb + t][e];
                }
                __syncthreads();
                if (pxg == 0) {
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        if (tb + t < 7)
#pragma unroll
                            for (int e = 0; e < 16; ++e) acc[tb + t][e] += red[(t * 16 + e) * GT + tg];
                }
            }
        if (pxg != 0) return;
    }
    const size_t rowf = (size_t)7 * 8 * CP;                      // floats per output channel
    float* dst = p.partial + (size_t)slice * p.Cout * rowf;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int t = ngrp * 7 + i, kh = t / C::TPK, qq = t % C::TPK;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            dst[(size_t)co * rowf + kh * 8 * CP + qq * 32 + (lane & 31)] = acc[i][e];
        }
    }
}
__global__ __launch_bounds__(512, 1) void conv_wgrad_stem16(WgradStemParams p) { wgrad_stem_body<16>(p); }
__global__ __launch_bounds__(512, 1) void conv_wgrad_stem8(WgradStemParams p) { wgrad_stem_body<8>(p); }
__global__ __launch_bounds__(512, 1) void conv_wgrad_stem4(WgradStemParams p) { wgrad_stem_body<4>(p); }

// dw[co][c][kh][kw] (torch layout, c < cin_real) = sum over the slices of partial[s][co][kh][kw*CP + c]
// block = one (co, kh) row of 8*CP floats: 32 lanes x 16 bytes along the row times 8 groups along the slices, combined in a
// fixed order through LDS (16-byte coalesced loads; the first version read 4 bytes per thread with a CP*4-byte stride)
__global__ __launch_bounds__(256) void conv_wgrad_stem_reduce(const float* __restrict__ partial, int splits, int Cout, int CP, int cin_real,
                                                              float* __restrict__ dw) {
    __shared__ float4 sc[8][32];
    const int co = blockIdx.x / 7, kh = blockIdx.x % 7, el = threadIdx.x & 31, g = threadIdx.x >> 5;
    const size_t rowf = (size_t)56 * CP;
    const int nv = 8 * CP / 4;   // float4 per row (32 or 16)
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (el < nv) {
        const float* q = partial + (size_t)co * rowf + (size_t)kh * 8 * CP + el * 4;
        const size_t zs = (size_t)Cout * rowf;
        int z = g;
        for (; z + 8 < splits; z += 16) {
            const float4 a = *reinterpret_cast<const float4*>(q + z * zs), c = *reinterpret_cast<const float4*>(q + (z + 8) * zs);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += c.x; s1.y += c.y; s1.z += c.z; s1.w += c.w;
        }
        if (z < splits) {
            const float4 a = *reinterpret_cast<const float4*>(q + z * zs);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
        s0.x += s1.x; s0.y += s1.y; s0.z += s1.z; s0.w += s1.w;
    }
    sc[g][el] = s0;
    __syncthreads();
    if (g == 0 && el < nv) {
        float4 r = sc[0][el];
#pragma unroll
        for (int k = 1; k < 8; ++k) { r.x += sc[k][el].x; r.y += sc[k][el].y; r.z += sc[k][el].z; r.w += sc[k][el].w; }
        const float v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = el * 4 + k, slot = e / CP, c = e - slot * CP;   // row position = slot * CP + c
            const int kw = CP == 4 ? slot - 1 : slot;                      // (CP = 4: slot 0 is the padding slot)
            if (kw >= 0 && kw < 7 && c < cin_real) dw[(((size_t)co * cin_real + c) * 7 + kh) * 7 + kw] = v[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// conv_wgrad_tr: the per-tap weight-gradient tile kernel (1x1 convs, strided 3x3) without register staging: 32-pixel
// slabs of dOut [pixel][64*MT co] and of the gathered input [pixel][64*NT ci] go HBM/L2 -> LDS by LDS-DMA into a ring of
// three stages (counted vmcnt + raw barrier, like conv_igemm_glds); the K = pixel fragments are read with
// ds_read_b64_tr_b16.  Replaces conv_wgrad's 8x8 register transposes (64 VALU + 8 ds_write_b128 per thread and slab).
// 16-byte slot swizzle by pixel so that the 4 pixel rows of a transposing read fall into different bank quarters.
// ---------------------------------------------------------------------------------------------------------------
template <int MT, int NT>
__global__ __launch_bounds__(256) void conv_wgrad_tr(WgradParams p) {
    constexpr int RA = 128 * MT, RB = 128 * NT;                 // bytes per pixel row of the A (dOut) / B (input) slab
    constexpr int ATILE = 32 * RA, BTILE = 32 * RB, STAGE = ATILE + BTILE;
    constexpr int PA = MT, PB = NT;                             // 1-KB pieces per wave and stage (4*MT resp. 4*NT per block)
    __shared__ __attribute__((aligned(16))) unsigned char sm[3 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const WgradBlock wb = wgrad_block(p);
    const int bco = wb.bco;
    const int tap = wb.by / p.ci_tiles, bci = wb.by % p.ci_tiles;
    const int kh = tap / p.KW, kw = tap % p.KW;
    const long M = (long)p.N * p.OH * p.OW;
    const long m_begin = (long)wb.bz * p.m_per_split;
    const long m_end = m_begin + p.m_per_split < M ? m_begin + p.m_per_split : M;
    if (m_begin >= M) return;
    const int ksteps = (int)((m_end - m_begin + 31) / 32);

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.dout), 0, (uint32_t)((size_t)M * p.Cout * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.in), 0, (uint32_t)((size_t)p.N * p.IH * p.IW * p.Cin * 2), 0x00020000);
    constexpr int OOB = (int)0x80000000;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto swzA = [](int px) { return MT == 1 ? (((px >> 1) & 1) << 2) : ((px & 3) << 2); };
    auto swzB = [](int px) { return NT == 1 ? (((px >> 1) & 1) << 2) : ((px & 3) << 2); };

    // loader state: piece q of this wave covers slab pixels (wave + 4q) * (8/MT) .. for A, (8/NT) for B
    constexpr int CA = 8 * MT, CB = 8 * NT;          // 16-byte chunks per pixel row
    int a_px[PA], a_col[PA], b_px[PB], b_col[PB];
    bool a_vc[PA], b_vc[PB];
    int bn[PB], boh[PB], bow[PB];                     // (image, oh, ow) of the B loader's pixel of the CURRENT slab
#pragma unroll
    for (int q = 0; q < PA; ++q) {
        a_px[q] = (wave + 4 * q) * (64 / CA) + lane / CA;
        const int ch = (lane % CA) ^ swzA(a_px[q]);
        a_col[q] = (bco * 64 * MT + ch * 8) * 2;
        a_vc[q] = bco * 64 * MT + ch * 8 < p.Cout;
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        b_px[q] = (wave + 4 * q) * (64 / CB) + lane / CB;
        const int ch = (lane % CB) ^ swzB(b_px[q]);
        b_col[q] = (bci * 64 * NT + ch * 8) * 2;
        b_vc[q] = bci * 64 * NT + ch * 8 < p.Cin;
        const long m = m_begin + b_px[q];
        bn[q] = (int)(m / ((long)p.OH * p.OW));
        const int rem = (int)(m - (long)bn[q] * p.OH * p.OW);
        boh[q] = rem / p.OW;
        bow[q] = rem - boh[q] * p.OW;
    }
    long mcur = m_begin;
    auto issue = [&](int buf) {
        unsigned char* a0 = sm + buf * STAGE;
        unsigned char* b0 = a0 + ATILE;
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const long m = mcur + a_px[q];
            const int vo = (a_vc[q] && m < m_end) ? (int)(m * p.Cout * 2) + a_col[q] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a0 + (wave + 4 * q) * 1024), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int ih = boh[q] * p.stride - p.pad + kh, iw = bow[q] * p.stride - p.pad + kw;
            const bool ok = b_vc[q] && mcur + b_px[q] < m_end && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW;
            const int vo = ok ? (((bn[q] * p.IH + ih) * p.IW + iw) * p.Cin) * 2 + b_col[q] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(b0 + (wave + 4 * q) * 1024), 16, vo, 0, 0, 0);
            bow[q] += 32;   // next slab
            while (bow[q] >= p.OW) { bow[q] -= p.OW; if (++boh[q] == p.OH) { boh[q] = 0; ++bn[q]; } }
        }
        mcur += 32;
    };

    // fragment addresses: lane group g = lane>>4 reads pixels 8*(g>>1) + (i>>2) [+4], columns 16*(g&1) + 4*(i&3) of a 32-wide tile
    const int g = lane >> 4, i4 = lane & 15;
    const int prow = 8 * (g >> 1) + (i4 >> 2);
    int fa[MT], fb[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int col = wm * 32 * MT + i * 32 + (g & 1) * 16 + 4 * (i4 & 3);
        fa[i] = prow * RA + (((col >> 3) ^ swzA(prow)) << 4) + (col & 7) * 2;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = wn * 32 * NT + j * 32 + (g & 1) * 16 + 4 * (i4 & 3);
        fb[j] = ATILE + prow * RB + (((col >> 3) ^ swzB(prow)) << 4) + (col & 7) * 2;
    }
    typedef __attribute__((address_space(3))) s16x4* lp;
    auto frag = [&](const unsigned char* q, int row_bytes) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(q));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(q + 4 * row_bytes));
        return __builtin_bit_cast(h16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    issue(0);
    if (ksteps > 1) issue(1);
    int buf = 0;
    for (int ks = 0; ks < ksteps; ++ks) {
        if (ks + 1 < ksteps) {
            if (PA + PB == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (PA + PB == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (ks + 2 < ksteps) issue(buf == 0 ? 2 : buf - 1);
        const unsigned char* st = sm + buf * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {   // the swizzle depends on pixel bits 0..1 (and 1): unchanged by +16 pixels
            h16x8 a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = frag(st + fa[i] + kk * 16 * RA, RA);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = frag(st + fb[j] + kk * 16 * RB, RB);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = MGN_MFMA_32x32x16(a[i], b[j], acc[i][j]);
        }
        buf = buf == 2 ? 0 : buf + 1;
    }
    const size_t wsize = (size_t)p.Cout * p.KH * p.KW * p.Cin;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int ci = bci * 64 * NT + wn * 32 * NT + j * 32 + (lane & 31);
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = bco * 64 * MT + wm * 32 * MT + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (co >= p.Cout) continue;
                p.partial[(size_t)wb.bz * wsize + (((size_t)co * p.KH * p.KW + tap) * p.Cin + ci)] = acc[i][j][e];
            }
    }
}

// dw = sum over the pixel splits (fixed order: deterministic), written in the requested layout.
// Block = 16 waves.  A wave owns 64 sixteen-byte columns of ONE output channel's (tap, ci) axis and every G-th split, G = 1 .. 16
// chosen from the split count so that a lane has about four splits to add (four independent accumulators: four 16-byte loads in
// flight per lane); the G waves of a column group are combined in wave order through LDS, and the 16 / G column groups of a block lie
// side by side.  Layers with few, large splits (512 -> 512: 8 x 9.4 MB) thus get blocks of 1024 columns x all splits, layers with
// hundreds of small ones (64 -> 64: 256 x 147 KB) get 16 waves per 64 columns instead of 4 -- the first form of this kernel (4 waves
// per 64 columns whatever the split count) left those layers with 192 blocks of 64 dependent loads: 0.7 TB/s inside the batched launch.
// (Round 1 had a thread per element: 4-byte loads in a dependent chain, 0.6 TB/s, 1.6 ms per training step.)
constexpr int RW = 16;   // waves per block
__host__ __device__ inline int reduce_groups(int splits) {
    int g = 1;
    while (g < RW && g * 4 < splits) g <<= 1;
    return g;
}
// blocks along the (tap, ci) axis of one output channel
__host__ __device__ inline int reduce_blocks_y(int splits, int taps, int Cin) {
    const int cols = 64 * (RW / reduce_groups(splits));
    return (taps * Cin / 4 + cols - 1) / cols;
}
__device__ __forceinline__ void f4add(float4& s, const float4& a) { s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w; }
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ partial, int splits, int Cout, int taps, int Cin, int oihw,
                                                  int cin_real, float* __restrict__ dw, int bx, int by) {
    __shared__ float4 scratch[RW][64];
    const int G = reduce_groups(splits);
    const int per = taps * Cin, co = bx, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = w % G, sub = w / G;
    const long wsize = (long)Cout * per;
    const int e4 = (by * (RW / G) + sub) * 64 + lane;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    if (e4 < per / 4) {
        const float* q = partial + (long)co * per + (long)e4 * 4;
        int z = g;
        for (; z + 3 * G < splits; z += 4 * G) {
            const float4 a = *reinterpret_cast<const float4*>(q + (long)z * wsize);
            const float4 b = *reinterpret_cast<const float4*>(q + (long)(z + G) * wsize);
            const float4 c = *reinterpret_cast<const float4*>(q + (long)(z + 2 * G) * wsize);
            const float4 d = *reinterpret_cast<const float4*>(q + (long)(z + 3 * G) * wsize);
            f4add(s0, a); f4add(s1, b); f4add(s2, c); f4add(s3, d);
        }
        if (z < splits) { f4add(s0, *reinterpret_cast<const float4*>(q + (long)z * wsize)); z += G; }
        if (z < splits) { f4add(s1, *reinterpret_cast<const float4*>(q + (long)z * wsize)); z += G; }
        if (z < splits) { f4add(s2, *reinterpret_cast<const float4*>(q + (long)z * wsize)); }
        f4add(s0, s1); f4add(s2, s3); f4add(s0, s2);
    }
    if (G > 1) {
        scratch[w][lane] = s0;
        __syncthreads();
        if (g != 0 || e4 >= per / 4) return;
        for (int k = 1; k < G; ++k) f4add(s0, scratch[w + k][lane]);   // wave order: fixed
    } else if (e4 >= per / 4) {
        return;
    }
    const float4 r = s0;
    if (!oihw) { *reinterpret_cast<float4*>(dw + (long)co * per + (long)e4 * 4) = r; return; }
    const int e = e4 * 4, tap = e / Cin, ci = e - tap * Cin;   // 4 consecutive input channels of one tap (Cin % 4 == 0)
    const float v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (ci + k < cin_real) dw[((long)co * cin_real + ci + k) * taps + tap] = v[k];
}
__global__ __launch_bounds__(64 * RW) void conv_wgrad_reduce(const float* __restrict__ partial, int splits, int Cout, int taps, int Cin, int oihw,
                                                             int cin_real, float* __restrict__ dw) {
    wgrad_reduce_body(partial, splits, Cout, taps, Cin, oihw, cin_real, dw, blockIdx.x, blockIdx.y);
}
// The reductions of MANY weight gradients in one launch (the gradient reducer batches the split-K sums of a bucket's convolutions:
// 70 launches of ~10 us per training step become one per bucket).  table: 10 x int64 per entry = {partial, dst, splits, Cout, taps,
// Cin, oihw, cin_real, first block, blocks along the (tap, ci) axis = mgn_conv_wgrad_reduce_blocks}; a block finds its entry by binary
// search on `first block`.
__global__ __launch_bounds__(64 * RW) void conv_wgrad_reduce_batch(const long long* __restrict__ table, int n_entries) {
    int lo = 0, hi = n_entries - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[(size_t)mid * 10 + 8] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long* e = table + (size_t)lo * 10;
    const int local = (int)((long long)blockIdx.x - e[8]), gy = (int)e[9];
    wgrad_reduce_body(reinterpret_cast<const float*>(e[0]), (int)e[2], (int)e[3], (int)e[4], (int)e[5], (int)e[6], (int)e[7],
                      reinterpret_cast<float*>(e[1]), local / gy, local % gy);
}
static inline dim3 wgrad_reduce_grid(int splits, int Cout, int taps, int Cin) { return dim3((unsigned)Cout, (unsigned)reduce_blocks_y(splits, taps, Cin)); }

// fp32 OIHW master weights -> bf16 kernel layouts, one launch per conv
//   mode 0: [Cout][KH][KW][Cin]                      (forward)
//   mode 1: [Cin][KH][KW][Cout], taps flipped        (data gradient: w'[ci][kh][kw][co] = w[co][ci][KH-1-kh][KW-1-kw])
//   mode 2: [Cout][Kpad], k = tap*Cp + c, zero padded (packed-tap stems; Cp = padded input channels);
//           Cp = 4 (7x7 only): k = kh*32 + (kw+1)*4 + c, Kpad = 224 -- the dense-row stem kernel
// element i of the bf16 layout (CoutP >= Cout: output channels zero-padded to CoutP, e.g. the few-class predictors to 32)
__device__ __forceinline__ float layout_value(const float* __restrict__ w, long i, int Cout, int CoutP, int Cin, int KH, int KW, int mode, int Cp) {
    if (mode == 0) {
        const int ci = (int)(i % Cin); long r = i / Cin;
        const int kw = (int)(r % KW); r /= KW;
        const int kh = (int)(r % KH); const int co = (int)(r / KH);
        return co < Cout ? w[(((long)co * Cin + ci) * KH + kh) * KW + kw] : 0.f;
    }
    if (mode == 1) {
        const int co = (int)(i % CoutP); long r = i / CoutP;
        const int kw = (int)(r % KW); r /= KW;
        const int kh = (int)(r % KH); const int ci = (int)(r / KH);
        return co < Cout ? w[(((long)co * Cin + ci) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)] : 0.f;
    }
    const int Kpad = (KH * KW * Cp + 31) / 32 * 32;
    const int k = (int)(i % Kpad), co = (int)(i / Kpad);
    if (Cp == 4) {   // dense kernel rows (7x7 stems, csrc/conv_stem.hip CP = 4): k = kh*32 + slot*4 + c, slot = kw + 1 (slot 0 is zero)
        const int kh = k >> 5, slot = (k >> 2) & 7, c = k & 3;
        return (co < Cout && kh < KH && slot >= 1 && slot <= KW && c < Cin) ? w[(((long)co * Cin + c) * KH + kh) * KW + slot - 1] : 0.f;
    }
    const int tap = k / Cp, c = k - tap * Cp;
    return (co < Cout && tap < KH * KW && c < Cin) ? w[(((long)co * Cin + c) * KH + tap / KW) * KW + tap % KW] : 0.f;
}

__global__ void weight_layout_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int Cout, int CoutP, int Cin, int KH, int KW,
                                     int mode, int Cp, long n_out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    out[i] = f2bf(layout_value(w, i, Cout, CoutP, Cin, KH, KW, mode, Cp));
}

// all conv weights of a model in ONE launch: table rows of 8 x int64 = {src fp32 OIHW, dst bf16, n_items, first block,
// Cout | CoutPad << 32, Cin, KH << 32 | KW, mode << 32 | Cp}; a block finds its row by binary search over the first-block column.
// Modes 0 / 1: an item is a (co, ci) PAIR -- the thread reads the pair's KH*KW taps (contiguous in OIHW) once and writes one
// element per tap, with the pair index ordered so that consecutive threads write consecutive elements (mode 0: ci fastest,
// mode 1: co fastest).  A thread per OUTPUT element re-read every source line once per tap (9x for the 3x3 layers).
// Mode 2 (packed stems): an item is an output element.
__global__ void weight_layout_batch_kernel(const long long* __restrict__ table, int n_entries) {
    int lo = 0, hi = n_entries - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[(size_t)mid * 8 + 3] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long* e = table + (size_t)lo * 8;
    const float* w = reinterpret_cast<const float*>(e[0]);
    uint16_t* out = reinterpret_cast<uint16_t*>(e[1]);
    const long n_out = e[2];
    const long i = ((long)blockIdx.x - e[3]) * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const int Cout = (int)(e[4] & 0xffffffff), CoutPr = (int)(e[4] >> 32), Cin = (int)e[5], KH = (int)(e[6] >> 32), KW = (int)(e[6] & 0xffffffff);
    const int mode = (int)(e[7] >> 32), Cp = (int)(e[7] & 0xffffffff);
    const int CoutP = CoutPr > Cout ? CoutPr : Cout;
    if (mode == 2) { out[i] = f2bf(layout_value(w, i, Cout, CoutP, Cin, KH, KW, mode, Cp)); return; }
    const int taps = KH * KW;
    if (mode == 1 && CoutP % 64 == 0 && Cin % 4 == 0 && taps <= 9) {
        // transposing layout through LDS: the block's 256 items are the tile (64 output channels) x (4 input channels), read along
        // the source rows (4 x taps contiguous floats per output channel) and written along the destination rows (64 consecutive
        // output channels = 128 B per wave and tap); the item-per-thread form below reads with a stride of Cin*taps floats
        __shared__ uint16_t tile[9][4][64 + 2];
        const int bl = (int)((long)blockIdx.x - e[3]), tiles_ci = Cin / 4;
        const int co0 = (bl / tiles_ci) * 64, ci0 = (bl % tiles_ci) * 4;
        {
            const int co_l = threadIdx.x >> 2, ci_l = threadIdx.x & 3, co = co0 + co_l;
            const float* src = w + ((long)co * Cin + ci0 + ci_l) * taps;
            for (int t = 0; t < taps; ++t) tile[t][ci_l][co_l] = co < Cout ? f2bf(src[t]) : (uint16_t)0;
        }
        __syncthreads();
        const int co_l = threadIdx.x & 63, ci_l = threadIdx.x >> 6;
        for (int t = 0; t < taps; ++t) out[((long)(ci0 + ci_l) * taps + (taps - 1 - t)) * CoutP + co0 + co_l] = tile[t][ci_l][co_l];
        return;
    }
    int co, ci;
    if (mode == 0) { co = (int)(i / Cin); ci = (int)(i - (long)co * Cin); }
    else { ci = (int)(i / CoutP); co = (int)(i - (long)ci * CoutP); }
    const float* src = w + ((long)co * Cin + ci) * taps;
    const bool real = co < Cout;
    for (int t = 0; t < taps; ++t) {
        const uint16_t v = real ? f2bf(src[t]) : (uint16_t)0;
        if (mode == 0) out[((long)co * taps + t) * Cin + ci] = v;                 // [Cout][KH][KW][Cin]
        else out[((long)ci * taps + (taps - 1 - t)) * CoutP + co] = v;            // [Cin][KH][KW][Cout], taps flipped
    }
}

}  // namespace

extern "C" {

int MGN_SYM(mgn_weight_layout)(const float* w_oihw, void* out_bf16, int Cout, int Cin, int KH, int KW, int mode, int Cp, int cout_pad, void* stream) {
    if (!w_oihw || !out_bf16 || Cout < 1 || Cin < 1 || KH < 1 || KW < 1 || mode < 0 || mode > 2) return MGN_EINVAL;
    if (mode == 2 && (Cp < Cin || (Cp != 4 && Cp != 8 && Cp != 16) || (Cp == 4 && (KH != 7 || KW != 7)))) return MGN_EINVAL;
    const int CoutP = cout_pad > Cout ? cout_pad : Cout;
    const int Kpad = mode == 2 ? (KH * KW * Cp + 31) / 32 * 32 : 0;
    const long n = mode == 2 ? (long)CoutP * Kpad : (long)CoutP * Cin * KH * KW;
    hipLaunchKernelGGL(weight_layout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oihw,
                       (uint16_t*)out_bf16, Cout, CoutP, Cin, KH, KW, mode, Cp, n);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int MGN_SYM(mgn_weight_layout_batch)(const void* table_dev, int n_entries, long total_blocks, void* stream) {
    if (!table_dev || n_entries < 1 || total_blocks < 1 || total_blocks > 0x7fffffffL) return MGN_EINVAL;
    hipLaunchKernelGGL(weight_layout_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const long long*)table_dev, n_entries);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

// the slice plan of conv3x3_c64 (one 8-wave block per CU): also the number of partial statistics rows it writes
static void c64_plan(int N, int OH, int OW, int Cout, Conv64Params* q) {
    q->N = N; q->H = OH; q->W = OW; q->Cout = Cout;
    q->strips = (OW + 127) / 128; q->co_tiles = Cout / 64;
    int chunks = (256 / q->co_tiles) / (N * q->strips);
    if (chunks > OH / 4) chunks = OH / 4;
    if (chunks < 1) chunks = 1;
    q->rows_per_chunk = (OH + chunks - 1) / chunks;
    q->chunks = (OH + q->rows_per_chunk - 1) / q->rows_per_chunk;
    q->nslices = N * q->strips * q->chunks;
}
static bool c64_eligible(int N, int IH, int IW, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int up) {
    return KH == 3 && KW == 3 && stride == 1 && pad == 1 && up == 1 && Cin == 64 && Cout % 64 == 0 && IH == OH && IW == OW &&
           (size_t)N * OH * OW * (Cout > 64 ? Cout : 64) * 2 < 0x7fffffffu && !getenv("MGN_CONV_NOC64");
}

static int conv_igemm_impl(const void* in, const void* w, void* out, const float* bias, int N, int IH, int IW, int Cin, int OH, int OW,
                           int Cout, int KH, int KW, int stride, int pad, int up, int relu, int out_f32, const void* residual,
                           float* stat_part, const float* stat_shift, int* plan_rows, void* stream, float slope = 0.f) {
    // relu: the activation of the epilogue -- 0 none, 1 ReLU, 2 leaky ReLU with `slope` (act(conv + bias + residual))
    // plan_rows != null: nothing is launched; *plan_rows = number of statistics rows the kernel chosen for this layer would write
    // (0 = that kernel has no statistics epilogue).  One decision path for launching and planning.
    if (plan_rows) *plan_rows = 0;
    if ((!plan_rows && (!in || !w || !out)) || N < 1 || IH < 1 || IW < 1 || OH < 1 || OW < 1 || Cout < 1 || KH < 1 || KW < 1 || stride < 1 || up < 1)
        return MGN_EINVAL;
    if (Cin == 4) {   // the dense-row stem layout: only the persistent stem kernel (csrc/conv_stem.hip) takes it
        if (KH != 7 || KW != 7 || stride != 2 || pad != 3 || up != 1 || bias || relu || out_f32 || residual) return MGN_ENOTSUP;
        const int sb = mgn_conv_stem7_blocks(N, IH, IW, Cin, OH, OW, Cout);
        if (sb <= 0) return MGN_ENOTSUP;
        if (plan_rows) { *plan_rows = sb; return MGN_OK; }
        return MGN_SYM(mgn_conv_stem7)(in, w, out, N, IH, IW, Cin, OH, OW, Cout, stat_part, stream);
    }
    const bool pack = (Cin == 8 || Cin == 16);            // small-Cin stems: taps packed into the k-slab
    if (!pack && (Cin < 32 || Cin % 32 != 0)) return MGN_ENOTSUP;  // k-slab = 32 or 64 input channels of one tap
    if (pack && up != 1) return MGN_ENOTSUP;
    ConvParams p;
    p.in = (const uint16_t*)in; p.w = (const uint16_t*)w; p.out = out; p.bias = bias; p.residual = (const uint16_t*)residual;
    if (residual && out_f32) return MGN_ENOTSUP;
    p.N = N; p.IH = IH; p.IW = IW; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout; p.KH = KH; p.KW = KW;
    p.stride = stride; p.pad = pad; p.up = up; p.relu = relu; p.slope = slope; p.out_f32 = out_f32; p.xcd_bands = 0; p.stat_part = stat_part;
    const bool stats_ok = !bias && !relu && !out_f32 && !residual && up == 1;   // what a statistics epilogue may be asked for
    // block -> XCD is (linear block id) % 8: the x-only remap is a per-XCD banding when the x extent is a multiple of 8 or the
    // grid is one-dimensional
    const bool no_xcd = getenv("MGN_CONV_NOXCD") != nullptr;
    auto xcd_ok = [&](long gxx, long gyz) { return (!no_xcd && gxx >= 16 && (gxx % 8 == 0 || gyz == 1)) ? 1 : 0; };
    const long M = (long)N * OH * OW;
    const long gx = (M + BM - 1) / BM;
    if (gx > 0x7fffffffL) return MGN_EINVAL;
    const bool k64 = (Cin % 64 == 0) && getenv("MGN_CONV_BK64");  // measured: no gain over BK=32 (LDS halves the residency)
    hipStream_t st = (hipStream_t)stream;
    if (!pack && KH == 1 && KW == 1 && pad == 0 && up == 1 && !bias && !relu && !out_f32 && !residual && (stride == 1 || stride == 2) &&
        OH == (IH - 1) / stride + 1 && OW == (IW - 1) / stride + 1 && (size_t)N * IH * IW * Cin * 2 < 0x7fffffffu && M < 0x7fffffffL &&
        !getenv("MGN_CONV_NO1X1") && !stat_part) {
        Conv1Params q;
        q.in = p.in; q.w = p.w; q.out = (uint16_t*)out; q.N = N; q.IH = IH; q.IW = IW; q.OH = OH; q.OW = OW; q.Cout = Cout;
        q.stride = stride; q.M = M; q.ntiles = 0; q.in2 = nullptr; q.out2 = nullptr;
        int rc1 = 1;   // 1 = not taken (shape without an instantiation, or too small): fall through to the generic kernels
        const bool pl = plan_rows != nullptr;   // (planning: 2 = would be taken -- the streaming kernel has no statistics epilogue)
        if (Cin == 256 && Cout % 256 == 0) rc1 = launch_conv1x1<256, 2, 4>(conv1x1_s_256_2_4, q, st, pl);
        else if (Cin == 256 && Cout == 32) rc1 = launch_conv1x1<256, 1, 1>(conv1x1_s_256_1_1, q, st, pl);
        else if (Cin == 32 && Cout % 256 == 0) rc1 = launch_conv1x1<32, 2, 4>(conv1x1_s_32_2_4, q, st, pl);
        else if (Cin == 64 && Cout % 128 == 0) rc1 = launch_conv1x1<64, 1, 4>(conv1x1_s_64_1_4, q, st, pl);
        else if (Cin == 128 && Cout % 256 == 0) rc1 = launch_conv1x1<128, 2, 4>(conv1x1_s_128_2_4, q, st, pl);
        else if (Cin == 512 && Cout % 128 == 0) rc1 = launch_conv1x1<512, 1, 4>(conv1x1_s_512_1_4, q, st, pl);
        if (rc1 == 2) return MGN_OK;
        if (rc1 <= 0) return rc1;
    }
    if (!pack && KH == 3 && KW == 3 && stride == 1 && pad == 1 && up == 1 && IH == OH && IW == OW && !out_f32 && !((bias || relu) && (stat_part || plan_rows))) {
        // windowed kernel (csrc/conv_win.hip): the input window of a 2-D pixel patch stays in LDS for all nine taps
        const int pr = mgn_conv_win_patch_rows(N, OH, OW, Cin, Cout);
        if (pr > 0 && plan_rows) {
            *plan_rows = stats_ok ? N * ((OH + pr - 1) / pr) * ((OW + 31) / 32) : 0;
            return MGN_OK;
        }
        if (pr > 0) {
            const int rcw = MGN_SYM(mgn_conv3x3_win_act)(in, w, out, N, OH, OW, Cin, Cout, residual, pr, stat_part, stat_shift, bias, relu, slope, stream);
            if (rcw != MGN_ENOTSUP || stat_part) return rcw;
        }
    }
    if (!pack && ((KH == 3 && KW == 3 && pad == 1) || (KH == 1 && KW == 1 && pad == 0)) && stride == 1 && up == 2 && !bias && !relu &&
        !out_f32 && !stat_part && !plan_rows && Cout % 64 == 0 && !getenv("MGN_CONV_NOUP2WIN")) {
        // data gradient of a 3x3 (or the shortcut's 1x1) stride-2 conv: all four parity classes from one low-resolution window
        // (csrc/conv_up2.hip)
        const int rcu = MGN_SYM(mgn_conv3x3_up2_win)(in, w, out, N, IH, IW, Cin, Cout, OH, OW, KH, residual, 0, stream);
        if (rcu != MGN_ENOTSUP) return rcu;
    }
    const bool c64 = !pack && !out_f32 && !((bias || relu) && (stat_part || plan_rows)) && c64_eligible(N, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, up);
    if (stat_part && !stats_ok) return MGN_ENOTSUP;   // (mgn_conv_stat_rows says which layers leave statistics behind)
    if (pack && KH == 7 && KW == 7 && stride == 2 && pad == 3 && stats_ok) {
        // the 64-channel stems: persistent windowed kernel with the weights in registers (csrc/conv_stem.hip), one statistics row per block
        const int sb = mgn_conv_stem7_blocks(N, IH, IW, Cin, OH, OW, Cout);
        if (sb > 0) {
            if (plan_rows) { *plan_rows = sb; return MGN_OK; }
            return MGN_SYM(mgn_conv_stem7)(in, w, out, N, IH, IW, Cin, OH, OW, Cout, stat_part, stream);
        }
    }
    if (pack && !getenv("MGN_CONV_NOPACKDMA") && stride >= 1 && up == 1 && (size_t)N * IH * IW * Cin * 2 < 0x7fffffffu) {
        // the stems on the LDS-DMA kernel (per-lane tap gather); grid like the generic LDS-DMA launch
        p.xcd_bands = xcd_ok(gx, (long)(Cout <= 64 ? (Cout + 63) / 64 : (Cout + 127) / 128));
        if (plan_rows) { *plan_rows = stats_ok ? (int)gx : 0; return MGN_OK; }
        if (Cout <= 64) hipLaunchKernelGGL((conv_igemm_glds<1, true>), dim3((unsigned)gx, (Cout + 63) / 64, 1), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_igemm_glds<2, true>), dim3((unsigned)gx, (Cout + 127) / 128, 1), dim3(256), 0, st, p);
        return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
    }
    if (stat_part && (pack || (!c64 && getenv("MGN_CONV_NOGLDS")))) return MGN_ENOTSUP;   // (the register-staged kernels have no epilogue)
    if (plan_rows && (pack || (!c64 && getenv("MGN_CONV_NOGLDS")))) return MGN_OK;
    if (pack) {
        if (Cout <= 64) hipLaunchKernelGGL((conv_igemm<1, 2, true>), dim3((unsigned)gx, (Cout + 63) / 64), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_igemm<2, 2, true>), dim3((unsigned)gx, (Cout + 127) / 128), dim3(256), 0, st, p);
    } else if (c64) {
        Conv64Params q;
        q.in = p.in; q.w = p.w; q.out = (uint16_t*)out; q.residual = p.residual; q.stat_part = stat_part;
        q.bias = bias; q.act = relu; q.slope = slope;
        c64_plan(N, OH, OW, Cout, &q);
        if (plan_rows) { *plan_rows = stats_ok ? q.nslices : 0; return MGN_OK; }
        static bool cattr = false;
        if (!cattr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64), hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_stats), hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_res), hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_act), hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_act_res), hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS);
            cattr = true;
        }
        if (q.stat_part && q.residual) return MGN_ENOTSUP;
        if (bias || relu) {   // out = act(conv + bias + residual): inference with the following norm folded in (mgn_conv_igemm_act)
            const dim3 g64((unsigned)((q.nslices + 7) / 8) * 8 * q.co_tiles);
            if (q.residual) hipLaunchKernelGGL(conv3x3_c64_act_res, g64, dim3(512), C64_LDS, st, q);
            else hipLaunchKernelGGL(conv3x3_c64_act, g64, dim3(512), C64_LDS, st, q);
            return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
        }
        if (q.residual) hipLaunchKernelGGL(conv3x3_c64_res, dim3((unsigned)((q.nslices + 7) / 8) * 8 * q.co_tiles), dim3(512), C64_LDS, st, q);
        else if (q.stat_part) hipLaunchKernelGGL(conv3x3_c64_stats, dim3((unsigned)((q.nslices + 7) / 8) * 8 * q.co_tiles), dim3(512), C64_LDS, st, q);
        else hipLaunchKernelGGL(conv3x3_c64, dim3((unsigned)((q.nslices + 7) / 8) * 8 * q.co_tiles), dim3(512), C64_LDS, st, q);
        return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
    } else if (!getenv("MGN_CONV_NOGLDS")) {
        // up > 1: one grid slice per parity class, sized for the largest class
        const long Mc = up > 1 ? (long)N * ((OH + up - 1) / up) * ((OW + up - 1) / up) : M;
        const unsigned gz = (unsigned)(up * up);
        const long gxc = (Mc + BM - 1) / BM;
        // big tiles when there are enough of them to fill 256 CUs (BN = 256: one 8-wave block per CU; BN = 128: two)
        const long gxb = (Mc + 255) / 256;
        static bool battr = false;
        if (!battr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_big128), hipFuncAttributeMaxDynamicSharedMemorySize, IgemmBig<2>::LDS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_big256), hipFuncAttributeMaxDynamicSharedMemorySize, IgemmBig<4>::LDS);
            battr = true;
        }
        const char* ebig = getenv("MGN_CONV_BIG");   // "0" disables, "128"/"256" force a tile
        const int fbig = ebig ? atoi(ebig) : -1;
        if (fbig != 0 && Cout % 128 == 0) {
            const bool can256 = Cout % 256 == 0;
            const long blocks256 = gxb * gz * (Cout / 256 > 0 ? Cout / 256 : 1);
            int pick = 0;
            if (fbig == 256 && can256) pick = 256;
            else if (fbig == 128) pick = 128;
            else if (fbig < 0) {
                if (can256 && blocks256 >= 200) pick = 256;
                // (BN = 128 measured slower than the 128 x 128 kernel on the 128-channel layers: only on request)
            }
            if (pick == 256) {
                if (plan_rows) { *plan_rows = stats_ok ? (int)gxb : 0; return MGN_OK; }
                p.xcd_bands = xcd_ok(gxb, (long)(Cout / 256) * gz);
                hipLaunchKernelGGL(conv_igemm_big256, dim3((unsigned)gxb, Cout / 256, gz), dim3(512), IgemmBig<4>::LDS, st, p);
                return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
            }
            // 512 x 128 tile: measured slower than the 128 x 128 kernel on the 128-channel layers (130 vs 122 us) -- the small-N
            // layers are bound by the 9x re-gathered A operand, not by the weight tile; only on request
            if (pick == 0 && Cout == 128 && fbig == 512) {
                if (plan_rows) { *plan_rows = stats_ok ? (int)((Mc + 511) / 512) : 0; return MGN_OK; }
                static bool a512 = false;
                constexpr int L512 = IgemmBig<2, 4>::LDS;
                if (!a512) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_big512x128), hipFuncAttributeMaxDynamicSharedMemorySize, L512);
                    a512 = true;
                }
                hipLaunchKernelGGL(conv_igemm_big512x128, dim3((unsigned)((Mc + 511) / 512), 1, gz), dim3(512), L512, st, p);
                return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
            }
            if (pick == 128) {
                if (plan_rows) { *plan_rows = stats_ok ? (int)gxb : 0; return MGN_OK; }
                p.xcd_bands = xcd_ok(gxb, (long)(Cout / 128) * gz);
                hipLaunchKernelGGL(conv_igemm_big128, dim3((unsigned)gxb, Cout / 128, gz), dim3(256), IgemmBig<2>::LDS, st, p);
                return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
            }
        }
        p.xcd_bands = xcd_ok(gxc, (long)(Cout <= 64 ? (Cout + 63) / 64 : (Cout + 127) / 128) * gz);
        if (plan_rows) { *plan_rows = stats_ok ? (int)gxc : 0; return MGN_OK; }
        if (Cout <= 64) hipLaunchKernelGGL((conv_igemm_glds<1>), dim3((unsigned)gxc, (Cout + 63) / 64, gz), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_igemm_glds<2>), dim3((unsigned)gxc, (Cout + 127) / 128, gz), dim3(256), 0, st, p);
    } else if (Cout <= 64) {
        const dim3 grid((unsigned)gx, (Cout + 63) / 64);
        if (k64) hipLaunchKernelGGL((conv_igemm<1, 4>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_igemm<1, 2>), grid, dim3(256), 0, st, p);
    } else {
        const dim3 grid((unsigned)gx, (Cout + 127) / 128);
        if (k64) hipLaunchKernelGGL((conv_igemm<2, 4>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_igemm<2, 2>), grid, dim3(256), 0, st, p);
    }
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int MGN_SYM(mgn_conv_igemm)(const void* in, const void* w, void* out, const float* bias, int N, int IH, int IW, int Cin, int OH, int OW,
                   int Cout, int KH, int KW, int stride, int pad, int up, int relu, int out_f32, const void* residual, void* stream) {
    return conv_igemm_impl(in, w, out, bias, N, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, up, relu, out_f32, residual, nullptr, nullptr, nullptr, stream);
}

/* out = act(conv(in, w) + bias + residual), act: 0 none | 1 ReLU | 2 leaky ReLU with `slope` -- mgn_conv_igemm with the activation as a
 * parameter and the bias / activation epilogue also in the windowed 3x3 and the 64-channel kernels: eval-mode `conv -> InPlaceABNSync`
 * (and the residual block's `-> + shortcut -> ReLU`) as ONE launch, the norm's fixed affine folded into the weights (scale) and this
 * bias (shift) by the caller.  16-bit output only. */
int MGN_SYM(mgn_conv_igemm_act)(const void* in, const void* w, void* out, const float* bias, int N, int IH, int IW, int Cin, int OH, int OW,
                                int Cout, int KH, int KW, int stride, int pad, int act, float slope, const void* residual, void* stream) {
    if (act < 0 || act > 2) return MGN_EINVAL;
    return conv_igemm_impl(in, w, out, bias, N, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, 1, act, 0, residual, nullptr, nullptr, nullptr, stream, slope);
}

int MGN_SYM(mgn_conv_igemm_stats)(const void* in, const void* w, void* out, int N, int IH, int IW, int Cin, int OH, int OW, int Cout, int KH,
                                  int KW, int stride, int pad, float* stat_partials, const float* stat_shift, void* stream) {
    if (!stat_partials) return MGN_EINVAL;
    return conv_igemm_impl(in, w, out, nullptr, N, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, 1, 0, 0, nullptr, stat_partials, stat_shift, nullptr, stream);
}

#ifndef MGN_F16
int mgn_conv_stat_rows(int N, int IH, int IW, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int* shifted) {
    if (shifted) *shifted = 0;
    if (N < 1 || OH < 1 || OW < 1 || Cout < 1) return 0;
    // the sums are taken around stat_shift only by the windowed kernel
    if (shifted && KH == 3 && KW == 3 && stride == 1 && pad == 1 && IH == OH && IW == OW && mgn_conv_win_patch_rows(N, OH, OW, Cin, Cout) > 0)
        *shifted = 1;
    int rows = 0;
    // (the same decision path as the launch: conv_igemm_impl in planning mode; the 16-bit format does not enter the decision)
    if (conv_igemm_impl(nullptr, nullptr, nullptr, nullptr, N, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, 1, 0, 0, nullptr, nullptr, nullptr,
                        &rows, nullptr) != MGN_OK)
        return 0;
    return rows;
}
#endif

static void wgrad_plan(int N, int OH, int OW, int Cin, int Cout, int KH, int KW, bool* pack, int* NT, int* MT, int* ci_tiles,
                       int* co_tiles, long* m_per_split, long* gz) {
    *pack = (Cin == 8 || Cin == 16);
    const int ncols = *pack ? KH * KW * Cin : Cin;
    *NT = ncols <= 64 ? 1 : 2;
    *MT = Cout <= 64 ? 1 : 2;
    *ci_tiles = (ncols + 64 * *NT - 1) / (64 * *NT);
    *co_tiles = (Cout + 64 * *MT - 1) / (64 * *MT);
    const long M = (long)N * OH * OW;
    const int tiles = *co_tiles * (*pack ? 1 : KH * KW) * *ci_tiles;
    long splits = (768 + tiles - 1) / tiles;              // ~768 blocks: 3 resident per CU on 256 CUs
    // >= 2048 pixels (32 k-steps) per block; the skinny 1x1 layers with one or two tiles (the predictors' 256 <-> 32 channels at
    // 128 x 256: 151 MB of inputs for 4 GFLOP) take 1024, i.e. two blocks per CU instead of one: 56 -> 43 us (round 5; the 256 -> 256
    // layer with its four tiles loses with it, 88 -> 112 us).  MGN_WGRAD_PXSPLIT overrides both (A/B)
    static const int px_env = getenv("MGN_WGRAD_PXSPLIT") ? atoi(getenv("MGN_WGRAD_PXSPLIT")) : 0;
    const int px_split = px_env > 0 ? px_env : (tiles <= 2 ? 1024 : 2048);
    long max_splits = (M + px_split - 1) / px_split;
    // ... unless that leaves most of the chip idle (the 1x1 layers of the 32x64 / 64x128 maps: 16-64 blocks, 44-56 us for 0.3-4 GFLOP):
    // then down to 256 pixels per block, up to one block per CU
    static const int min_px = getenv("MGN_WGRAD_MINPX") ? atoi(getenv("MGN_WGRAD_MINPX")) : 256;
    if (tiles * max_splits < 256 && min_px < 2048) {
        const long want = (256 + tiles - 1) / tiles, cap = (M + min_px - 1) / min_px;
        max_splits = want < cap ? want : cap;
    }
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    *m_per_split = ((M + splits - 1) / splits + WBK - 1) / WBK * WBK;
    *gz = (M + *m_per_split - 1) / *m_per_split;
}

// 7x7 / stride 2 / pad 3 on a channel-padded (8 | 16) input -> the stem row-march kernel
static bool wgrad_stem_plan(int N, int OH, int OW, int Cin, int Cout, int KH, int KW, int stride, int pad, int IH, int IW, WgradStemParams* p) {
    if (getenv("MGN_WGRAD_NOSTEM")) return false;
    if (KH != 7 || KW != 7 || stride != 2 || pad != 3 || (Cin != 4 && Cin != 8 && Cin != 16) || Cout % 64) return false;
    if (Cin == 4 && (IW & 1)) return false;   // (aligned pixel pairs)
    if (OH != (IH - 1) / 2 + 1 || OW != (IW - 1) / 2 + 1) return false;
    if ((size_t)N * OH * OW * Cout * 2 >= 0x7fffffffu || (size_t)N * IH * IW * Cin * 2 >= 0x7fffffffu) return false;
    p->N = N; p->IH = IH; p->IW = IW; p->OH = OH; p->OW = OW; p->Cout = Cout;
    p->co_tiles = Cout / 64;
    p->strips = (OW + 127) / 128;
    int chunks = (256 / p->co_tiles) / (N * p->strips);   // one 8-wave block per CU
    if (chunks > OH / 8) chunks = OH / 8;                 // >= 8 rows per block (prologue of 9 input rows)
    if (chunks < 1) chunks = 1;
    p->rows_per_chunk = (OH + chunks - 1) / chunks;
    p->chunks = (OH + p->rows_per_chunk - 1) / p->rows_per_chunk;
    p->nslices = N * p->strips * p->chunks;
    return true;
}

// 3x3 / stride 1 / pad 1 with 64-multiples of channels -> the all-taps row-march kernel
static bool wgrad3_plan(int N, int OH, int OW, int Cin, int Cout, int KH, int KW, int stride, int pad, int IH, int IW, Wgrad3Params* p) {
    if (getenv("MGN_WGRAD_NO3X3")) return false;
    if (KH != 3 || KW != 3 || stride != 1 || pad != 1 || IH != OH || IW != OW || Cin % 64 || Cout % 64) return false;
    if ((size_t)N * OH * OW * (Cin > Cout ? Cin : Cout) * 2 >= 0x7fffffffu) return false;  // 32-bit buffer offsets
    p->N = N; p->H = OH; p->W = OW; p->Cin = Cin; p->Cout = Cout;
    p->co_tiles = Cout / 64; p->ci_tiles = Cin / 64;
    const char* eng = getenv("MGN_WGRAD3_NG");
    p->ng = eng ? atoi(eng) : (OW > 64 ? 2 : 1);        // 128-pixel strips (8 waves) unless the image is narrower
    if (p->ng != 1 && p->ng != 2) p->ng = 1;
    p->strips = (OW + 64 * p->ng - 1) / (64 * p->ng);
    const int tiles = p->co_tiles * p->ci_tiles;
    const int resident = 256 * (2 / p->ng);               // blocks resident at once: one round of equal-work blocks
    int want = (resident + tiles - 1) / tiles;
    int chunks = want / (N * p->strips);
    if (chunks > OH / 4) chunks = OH / 4;                 // >= 4 rows per block (2 rows of prologue)
    if (chunks < 1) chunks = 1;
    p->rows_per_chunk = (OH + chunks - 1) / chunks;
    p->chunks = (OH + p->rows_per_chunk - 1) / p->rows_per_chunk;
    p->nslices = N * p->strips * p->chunks;
    return true;
}

#ifndef MGN_F16
int mgn_conv_wgrad_workspace_bytes(int N, int OH, int OW, int Cin, int Cout, int KH, int KW, size_t* bytes) {
    if (!bytes || N < 1 || OH < 1 || OW < 1 || Cin < 1 || Cout < 1 || KH < 1 || KW < 1) return MGN_EINVAL;
    bool pack; int NT, MT, cit, cot; long mps, gz;
    wgrad_plan(N, OH, OW, Cin, Cout, KH, KW, &pack, &NT, &MT, &cit, &cot, &mps, &gz);
    *bytes = sizeof(float) * (size_t)gz * Cout * KH * KW * Cin;
    WgradStemParams ps;
    if (wgrad_stem_plan(N, OH, OW, Cin, Cout, KH, KW, 2, 3, 2 * OH, 2 * OW, &ps)) {
        const size_t b7 = sizeof(float) * (size_t)ps.nslices * Cout * 56 * Cin;
        if (b7 > *bytes) *bytes = b7;
    }
    Wgrad3Params p3;  // stride / pad are not known here: cover the 3x3 stride-1 plan too
    if (wgrad3_plan(N, OH, OW, Cin, Cout, KH, KW, 1, 1, OH, OW, &p3)) {
        const size_t b3 = sizeof(float) * (size_t)p3.nslices * Cout * 9 * Cin;
        if (b3 > *bytes) *bytes = b3;
    }
    return MGN_OK;
}

#endif
// desc != null: "partial only" -- the split-K partials stay in the workspace, no reduction is launched, desc[0..7] = {partial,
// 0, splits, Cout, taps, Cin, oihw, cin_real} describes the reduction for mgn_conv_wgrad_reduce_batch (dw is not written)
static int wgrad_impl(const void* dout, const void* in, float* dw, int N, int IH, int IW, int Cin, int OH, int OW, int Cout, int KH,
                      int KW, int stride, int pad, int oihw_cin, void* workspace, size_t workspace_bytes, long long* desc, void* stream,
                      const void* in2 = nullptr) {
    if (!dout || !in || (!dw && !desc) || !workspace || N < 1 || IH < 1 || IW < 1 || OH < 1 || OW < 1 || KH < 1 || KW < 1 || stride < 1) return MGN_EINVAL;
    if ((Cin % 8 != 0 && Cin != 4) || Cout % 8 != 0) return MGN_ENOTSUP;
    WgradParams p;
    p.dout = (const uint16_t*)dout; p.in = (const uint16_t*)in; p.dw = dw;
    p.N = N; p.IH = IH; p.IW = IW; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.oihw = oihw_cin > 0; p.cin_real = oihw_cin > 0 ? oihw_cin : Cin;
    p.in2 = (const uint16_t*)in2;
    if (oihw_cin > Cin) return MGN_EINVAL;
    // two-source input (in | in2): the 1x1 layers of the split-K tile kernel only, column tiles of 128 inside one map
    if (in2 && (KH != 1 || KW != 1 || stride != 1 || pad != 0 || Cin % 256 != 0 || (size_t)N * IH * IW * Cin >= 0x7fffffffu)) return MGN_ENOTSUP;
    hipStream_t st = (hipStream_t)stream;
    const size_t wsize = (size_t)Cout * KH * KW * Cin;
    WgradStemParams ps;
    if (oihw_cin > 0 && wgrad_stem_plan(N, OH, OW, Cin, Cout, KH, KW, stride, pad, IH, IW, &ps)) {
        if (desc) return MGN_ENOTSUP;   // (the stems keep their own reduction kernel)
        if (workspace_bytes < sizeof(float) * (size_t)ps.nslices * Cout * 56 * Cin) return MGN_ENOSPC;
        ps.dout = p.dout; ps.in = p.in; ps.partial = (float*)workspace;
        static bool attrs = false;
        if (!attrs) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_stem16), hipFuncAttributeMaxDynamicSharedMemorySize, WS<16>::LDS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_stem8), hipFuncAttributeMaxDynamicSharedMemorySize, WS<8>::LDS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_stem4), hipFuncAttributeMaxDynamicSharedMemorySize, WS<4>::LDS);
            attrs = true;
        }
        const unsigned nblk = (unsigned)((ps.nslices + 7) / 8) * 8 * ps.co_tiles;
        if (Cin == 16) hipLaunchKernelGGL(conv_wgrad_stem16, dim3(nblk), dim3(512), WS<16>::LDS, st, ps);
        else if (Cin == 8) hipLaunchKernelGGL(conv_wgrad_stem8, dim3(nblk), dim3(512), WS<8>::LDS, st, ps);
        else hipLaunchKernelGGL(conv_wgrad_stem4, dim3(nblk), dim3(512), WS<4>::LDS, st, ps);
        hipLaunchKernelGGL(conv_wgrad_stem_reduce, dim3((unsigned)Cout * 7), dim3(256), 0, st, (const float*)workspace, ps.nslices, Cout, Cin,
                           oihw_cin, dw);
        return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
    }
    // 8-byte pixels (Cin == 4) exist for the dense-row stem kernel only: every kernel below reads 16-byte segments of 8 channels and
    // would take two pixels for one
    if (Cin == 4) return MGN_ENOTSUP;
    Wgrad3Params p3;
    if (wgrad3_plan(N, OH, OW, Cin, Cout, KH, KW, stride, pad, IH, IW, &p3)) {
        if (workspace_bytes < sizeof(float) * p3.nslices * wsize) return MGN_ENOSPC;
        p3.dout = p.dout; p3.in = p.in; p3.partial = (float*)workspace;
        static bool attr3 = false;
        if (!attr3) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad3x3_s64), hipFuncAttributeMaxDynamicSharedMemorySize, W3<1>::LDS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad3x3_s128), hipFuncAttributeMaxDynamicSharedMemorySize, W3<2>::LDS);
            attr3 = true;
        }
        const unsigned nblk = (unsigned)((p3.nslices + 7) / 8) * 8 * p3.co_tiles * p3.ci_tiles;
        if (p3.ng == 2) hipLaunchKernelGGL(conv_wgrad3x3_s128, dim3(nblk), dim3(512), W3<2>::LDS, st, p3);
        else hipLaunchKernelGGL(conv_wgrad3x3_s64, dim3(nblk), dim3(256), W3<1>::LDS, st, p3);
        if (desc) {
            desc[0] = (long long)(uintptr_t)workspace; desc[1] = 0; desc[2] = p3.nslices; desc[3] = Cout; desc[4] = 9; desc[5] = Cin;
            desc[6] = p.oihw; desc[7] = p.cin_real;
        } else {
            hipLaunchKernelGGL(conv_wgrad_reduce, wgrad_reduce_grid(p3.nslices, Cout, 9, Cin), dim3(64 * RW), 0, st, (const float*)workspace, p3.nslices, Cout, 9,
                               Cin, p.oihw, p.cin_real, dw);
        }
        return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
    }
    bool pack; int NT, MT, co_tiles; long gz;
    wgrad_plan(N, OH, OW, Cin, Cout, KH, KW, &pack, &NT, &MT, &p.ci_tiles, &co_tiles, &p.m_per_split, &gz);
    if (workspace_bytes < sizeof(float) * gz * wsize) return MGN_ENOSPC;
    p.partial = (float*)workspace;
    dim3 grid(co_tiles, (pack ? 1 : KH * KW) * p.ci_tiles, (unsigned)gz);
    p.remap_tiles = 0; p.co_tiles = co_tiles;
    const long tiles_per_split = (long)grid.x * grid.y;
    if (tiles_per_split > 1 && gz >= 8 && tiles_per_split * ((gz + 7) / 8 * 8) < 0x7fffffffL && !getenv("MGN_WGRAD_NOREMAP")) {
        p.remap_tiles = (int)tiles_per_split;      // the tile blocks of a pixel split on one XCD (wgrad_block)
        grid = dim3((unsigned)(tiles_per_split * ((gz + 7) / 8 * 8)), 1, 1);
    }
    const size_t lds = 4 * (size_t)WTILE;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad<1, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad<2, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const size_t M_total = (size_t)N * OH * OW;
    // (measured: the transposing-read kernel wins on the strided 3x3 layers, 168->144 / 124->90 / 120->90 us; the 1x1 layers
    //  are bound by their inputs and split partials either way and stay on conv_wgrad)
    const bool use_tr = !pack && KH * KW > 1 && !getenv("MGN_WGRAD_NOTR") && M_total * (Cout > Cin ? Cout : Cin) * 2 < 0x7fffffffu &&
                        (size_t)N * IH * IW * Cin * 2 < 0x7fffffffu;
    if (pack && MT == 1) hipLaunchKernelGGL((conv_wgrad<1, 2, true>), grid, dim3(256), lds, st, p);
    else if (pack) hipLaunchKernelGGL((conv_wgrad<2, 2, true>), grid, dim3(256), lds, st, p);
    else if (use_tr && MT == 1 && NT == 1) hipLaunchKernelGGL((conv_wgrad_tr<1, 1>), grid, dim3(256), 0, st, p);
    else if (use_tr && MT == 1) hipLaunchKernelGGL((conv_wgrad_tr<1, 2>), grid, dim3(256), 0, st, p);
    else if (use_tr && NT == 1) hipLaunchKernelGGL((conv_wgrad_tr<2, 1>), grid, dim3(256), 0, st, p);
    else if (use_tr) hipLaunchKernelGGL((conv_wgrad_tr<2, 2>), grid, dim3(256), 0, st, p);
    else if (MT == 1 && NT == 1) hipLaunchKernelGGL((conv_wgrad<1, 1>), grid, dim3(256), lds, st, p);
    else if (MT == 1) hipLaunchKernelGGL((conv_wgrad<1, 2>), grid, dim3(256), lds, st, p);
    else if (NT == 1) hipLaunchKernelGGL((conv_wgrad<2, 1>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((conv_wgrad<2, 2>), grid, dim3(256), lds, st, p);
    if (desc) {
        desc[0] = (long long)(uintptr_t)workspace; desc[1] = 0; desc[2] = gz; desc[3] = Cout; desc[4] = KH * KW; desc[5] = Cin;
        desc[6] = p.oihw; desc[7] = p.cin_real;
    } else {
        hipLaunchKernelGGL(conv_wgrad_reduce, wgrad_reduce_grid((int)gz, Cout, KH * KW, Cin), dim3(64 * RW), 0, st, (const float*)workspace, (int)gz, Cout,
                           KH * KW, Cin, p.oihw, p.cin_real, dw);
    }
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int MGN_SYM(mgn_conv_wgrad)(const void* dout, const void* in, float* dw, int N, int IH, int IW, int Cin, int OH, int OW, int Cout, int KH,
                   int KW, int stride, int pad, int oihw_cin, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dw) return MGN_EINVAL;
    return wgrad_impl(dout, in, dw, N, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, oihw_cin, workspace, workspace_bytes, nullptr, stream);
}

int MGN_SYM(mgn_conv_wgrad_partial)(const void* dout, const void* in, int N, int IH, int IW, int Cin, int OH, int OW, int Cout, int KH, int KW,
                                    int stride, int pad, int oihw_cin, void* workspace, size_t workspace_bytes, long long* desc8, void* stream) {
    if (!desc8) return MGN_EINVAL;
    return wgrad_impl(dout, in, nullptr, N, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, oihw_cin, workspace, workspace_bytes, desc8, stream);
}

/* weight gradient of a 1x1 convolution whose input is the channel concatenation (in0 | in1) of two [N, H, W, Cin / 2] maps (never
 * materialised); dw / desc8 as in mgn_conv_wgrad / mgn_conv_wgrad_partial (exactly one of them non-null).  Cin % 256 == 0. */
int MGN_SYM(mgn_conv_wgrad_cat)(const void* dout, const void* in0, const void* in1, float* dw, int N, int H, int W, int Cin, int Cout,
                                void* workspace, size_t workspace_bytes, long long* desc8, void* stream) {
    if (!in1 || (!dw) == (!desc8)) return MGN_EINVAL;
    return wgrad_impl(dout, in0, dw, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, Cin, workspace, workspace_bytes, desc8, stream, in1);
}

/* out = conv1x1(in0 | in1, w): the FeatureFusionModule's convolution over the concatenation of its two inputs (layers.py:316-317)
 * without the concatenated map.  in0, in1: [N, H, W, 128]; w: [Cout][256] (layout mode 0); out: [N, H, W, Cout], Cout % 256 == 0.
 * MGN_ENOTSUP: other channel counts, or a map too small for the streaming kernel (the caller concatenates). */
int MGN_SYM(mgn_conv1x1_cat)(const void* in0, const void* in1, const void* w, void* out, int N, int H, int W, int Cin, int Cout, void* stream) {
    if (!in0 || !in1 || !w || !out || N < 1 || H < 1 || W < 1) return MGN_EINVAL;
    const long M = (long)N * H * W;
    if (Cin != 256 || Cout % 256 != 0 || (size_t)M * 128 * 2 >= 0x7fffffffu || M >= 0x7fffffffL || getenv("MGN_CONV_NO1X1")) return MGN_ENOTSUP;
    Conv1Params q;
    q.in = (const uint16_t*)in0; q.in2 = (const uint16_t*)in1; q.w = (const uint16_t*)w; q.out = (uint16_t*)out; q.out2 = nullptr;
    q.N = N; q.IH = H; q.IW = W; q.OH = H; q.OW = W; q.Cout = Cout; q.stride = 1; q.M = M; q.ntiles = 0;
    const int rc = launch_conv1x1<256, 2, 4>(conv1x1_s_256_2_4_cat, q, (hipStream_t)stream);
    return rc == 1 ? MGN_ENOTSUP : rc;
}

/* (out0 | out1) = conv1x1(in, w): the data gradient of the convolution above -- in: [N, H, W, 256] (the output gradient), w: [Cout = 256]
 * [256] (layout mode 1), out0 / out1: [N, H, W, 128] each = the gradients of the two concatenated maps, written directly. */
int MGN_SYM(mgn_conv1x1_split)(const void* in, const void* w, void* out0, void* out1, int N, int H, int W, int Cin, int Cout, void* stream) {
    if (!in || !w || !out0 || !out1 || N < 1 || H < 1 || W < 1) return MGN_EINVAL;
    const long M = (long)N * H * W;
    if (Cin != 256 || Cout != 256 || (size_t)M * 256 * 2 >= 0x7fffffffu || M >= 0x7fffffffL || getenv("MGN_CONV_NO1X1")) return MGN_ENOTSUP;
    Conv1Params q;
    q.in = (const uint16_t*)in; q.in2 = nullptr; q.w = (const uint16_t*)w; q.out = (uint16_t*)out0; q.out2 = (uint16_t*)out1;
    q.N = N; q.IH = H; q.IW = W; q.OH = H; q.OW = W; q.Cout = Cout; q.stride = 1; q.M = M; q.ntiles = 0;
    const int rc = launch_conv1x1<256, 2, 4>(conv1x1_s_256_2_4, q, (hipStream_t)stream);
    return rc == 1 ? MGN_ENOTSUP : rc;
}

#ifndef MGN_F16
int mgn_conv_wgrad_reduce_batch(const void* table_dev, int n_entries, long total_blocks, void* stream) {
    if (!table_dev || n_entries < 1 || total_blocks < 1 || total_blocks > 0x7fffffffL) return MGN_EINVAL;
    hipLaunchKernelGGL(conv_wgrad_reduce_batch, dim3((unsigned)total_blocks), dim3(64 * RW), 0, (hipStream_t)stream, (const long long*)table_dev, n_entries);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}
/* blocks along the (tap, ci) axis of ONE output channel of a reduction with `splits` partial tiles: column 9 of the batch table; an entry
 * takes Cout x this many blocks */
int mgn_conv_wgrad_reduce_blocks(int splits, int taps, int Cin) {
    if (splits < 1 || taps < 1 || Cin < 4 || Cin % 4) return MGN_EINVAL;
    return reduce_blocks_y(splits, taps, Cin);
}
#endif

}  // extern "C"

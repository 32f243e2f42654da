// conv_stem.hip -- the 7x7 / stride 2 / pad 3 stems (64 output channels) on the channel-padded input of csrc/prep.hip as a
// PERSISTENT windowed implicit GEMM with the weights in registers (gfx950).
//
// Replaces, for mgnet/modeling/res_net.py:96-104 (BasicStem conv1 of the backbone, 3 -> 64) and the 9-channel pose-net stem
// (res_net.py:169-181 / mg_net.py:268-274), the packed-tap path of conv.hip (`conv_igemm_glds<1, PACK>`), which gathers every
// 16-byte (pixel, channel-half) segment once per tap and k-step, recomputes the gather address with integer divisions every step
// and runs 4 MFMAs per barrier: 360-450 TFLOP/s on the padded reduction.  Same contract: in [N, IH, IW, CP] 16-bit with CP = 8 | 16
// (zero padded channels), weights in layout mode 2 of mgn_weight_layout ([64][Kpad], k = tap*CP + c, zero padded),
// out [N, OH, OW, 64]; optional statistics rows of the rounded outputs for the InPlaceABNSync that follows.
//
// Design: the whole weight matrix is tiny (64 x 392 / 64 x 784 values), the output is huge (8 x 512 x 1024 x 64) -- so the
// weights live in REGISTERS for the lifetime of a block (one wave per SIMD, <= 200 of the 512 unified registers; the matrix
// cores read them in place), a block walks over many 8 x 32-pixel output patches, and the only thing that moves per patch is
// its 21 x 69 input window: global -> LDS by LDS-DMA (zero padding from the buffer bounds check), double buffered, next to the
// MFMAs of the current patch.  The k loop of a patch has no barrier, no global access and no address arithmetic: a tap is a
// compile-time offset into the window.
//
//   window : [21 rows][CP/8 channel halves][2 column parities][36] x 16 B.  Column parity planes make the stride-2 gather of a
//            fragment read (32 lanes = 32 output columns, input column 2*ox + kw) 512 contiguous bytes -> conflict-free ds_read_b128
//   CP = 8 : wave = 2 output rows x 64 channels; k-slab of 16 = two taps (lane halves), 25 slabs; 32 fragment reads per 100 MFMAs
//   CP = 16: wave = 4 output rows x 32 channels (2 x 2 waves); k-slab = one tap, 49 slabs; 91 fragment reads per 196 MFMAs
//            (a fragment of a window row serves every (output row, kernel row) pair that meets on it)
//   stores : 16-byte buffer stores (out-of-range pixels dropped by the bounds check: the instruction count per patch is
//            constant, so `s_waitcnt vmcnt(8)` at the top of the next patch waits for the window DMA but not for the stores)
//   stats  : per-lane running sums over ALL patches of the block, one partial row per block at the end
//
// CP = 4 (round 5, the backbone stem at its dense K): the input is [N, IH, IW, 4] (3 real channels, 8 bytes per pixel, IW even) and a
// kernel ROW is one dense run of 8 column slots x 4 channels = 32 k-values: slot kw' = kw + 1 (slot 0 has zero weights), so that the
// window starts at the EVEN image column 2 ox0 - 4 and a 16-byte LDS element is an aligned PAIR of pixels = the eight k-values
// (slots 2j, 2j + 1) a lane feeds to one MFMA.  Output column ox and slot pair j read element (ox - ox0) + j of the window row: no
// parity planes, 32 lanes x 16 contiguous bytes.  K = 7 x 32 = 224 instead of 49 x 8 = 392 (147 real): 14 slabs instead of 25, a
// 21 x 36-element window (12 KB) instead of 24 KB.  A wave owns 4 output rows x 32 channels (56 weight registers, 64 accumulators):
// under 256 registers, so TWO blocks share a CU and one block's window DMA latency hides behind the other's MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mgnet_hip.h"

namespace {

#include "h16.h"
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct StemParams {
    const uint16_t* in;   // [N, IH, IW, CP]
    const uint16_t* w;    // [64][kpad]
    uint16_t* out;        // [N, OH, OW, 64]
    float* stat_part;     // [gridDim.x][64][2] sums of r, r^2 over the rounded outputs, or null
    int N, IH, IW, OH, OW;
    int py, px, npatch;   // patches per image (rows of 8, columns of 32) and in total
    int kpad;
};
MGN_PLAN_RO_CONV(StemParams, MGN_RO(in) MGN_RO(w))   // pointers the kernels only read through (launch-plan dependency analysis, csrc/mgn_launch.h)

constexpr int SPH = 8, SPW = 32;   // output patch
constexpr int WROWS = 2 * SPH + 5; // 21 input rows
constexpr int PLANE = 36;          // entries per column-parity plane (35 even / 34 odd columns are real)
constexpr int OOB = (int)0x80000000;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int CP> struct Stem {
    static constexpr int NH = CP / 8;                  // 16-byte channel halves per pixel
    static constexpr int S = CP == 4 ? 14 : (CP == 8 ? 25 : 49);   // k-slabs of 16
    static constexpr int SPR = CP == 4 ? 4 : 7;        // slabs two window rows further down (CP = 4 | 8: the shared fragment stream)
    static constexpr int CB = CP == 8 ? 2 : 1;         // 32-channel blocks per wave
    static constexpr int RW = CP == 8 ? 2 : 4;         // output rows per wave  (CB = 1: 2 row groups x 2 channel halves of waves)
    static constexpr int BPC = CP == 4 ? 2 : 1;        // resident blocks per CU the kernel is built for
    static constexpr int ROWP = CP == 4 ? PLANE : NH * 2 * PLANE;   // 16-byte elements per window row
    static constexpr int ELEMS = WROWS * ROWP;
    static constexpr int CHUNKS = (ELEMS + 63) / 64;   // 1-KB DMA pieces: 24 / 48
    static constexpr int CPW = CHUNKS / 4;             // per wave
    static constexpr int WBUF = CHUNKS * 1024;
    static constexpr int LDS = 2 * WBUF + 4 * 2 * 64 * 2 * 4;   // + the statistics scratch
    static_assert(CHUNKS % 4 == 0, "window pieces are dealt to four waves");
};

// byte offset of tap t inside the window (relative to the wave's first row and the lane's column)
template <int CP> __device__ __forceinline__ constexpr int tap_off(int t) {
    const int kh = t / 7, kw = t % 7;
    return (kh * Stem<CP>::ROWP + (kw & 1) * PLANE + (kw >> 1)) * 16;
}

template <int CP>
__device__ __forceinline__ void stem_body(const StemParams& p) {
    using G = Stem<CP>;
    constexpr int S = G::S, CB = G::CB, RW = G::RW;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int wr = CB == 1 ? (wave >> 1) : wave, wc = CB == 1 ? (wave & 1) : 0;   // row group / channel half of the wave

    const uint32_t in_bytes = (uint32_t)((size_t)p.N * p.IH * p.IW * CP * 2), out_bytes = (uint32_t)((size_t)p.N * p.OH * p.OW * 64 * 2);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.in), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, out_bytes, 0x00020000);

    // the window elements this lane fetches (piece = wave + 4*j): (row << 16) | (column << 1) | channel half, or -1
    int desc[G::CPW];
#pragma unroll
    for (int j = 0; j < G::CPW; ++j) {
        const int e = (wave + 4 * j) * 64 + lane;
        const int rowl = e / G::ROWP, rem = e - rowl * G::ROWP;
        if (CP == 4) {   // element = the pixel pair at window columns 2 rem, 2 rem + 1 (35 pairs are real)
            desc[j] = (e < G::ELEMS && rem <= SPW + 2) ? ((rowl << 16) | (2 * rem << 1)) : -1;
            continue;
        }
        const int h = rem / (2 * PLANE), r2 = rem - h * 2 * PLANE;
        const int plane = r2 / PLANE, idx = r2 - plane * PLANE;
        const int lc = 2 * idx + plane;
        desc[j] = (e < G::ELEMS && lc <= 2 * SPW + 4) ? ((rowl << 16) | (lc << 1) | h) : -1;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int ppi = p.py * p.px;
    auto issue = [&](int patch, int buf) {
        const int n = patch / ppi, prem = patch - n * ppi, pyi = prem / p.px, pxi = prem - pyi * p.px;
        const int iy0 = 2 * SPH * pyi - 3, ix0 = 2 * SPW * pxi - (CP == 4 ? 4 : 3);
        unsigned char* dst = sm + buf * G::WBUF + wave * 1024;
#pragma unroll
        for (int j = 0; j < G::CPW; ++j) {
            const int d = desc[j];
            const int iy = iy0 + (d >> 16), ix = ix0 + ((d >> 1) & 0x7fff);
            const bool ok = d >= 0 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
            const int off = ok ? ((n * p.IH + iy) * p.IW + ix) * (CP * 2) + (d & 1) * 16 : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(dst + j * 4096), 16, off, 0, 0, 0);
        }
    };

    int patch = blockIdx.x;
    if (patch >= p.npatch) return;   // (the host launches at most npatch blocks)
    issue(patch, 0);

    // weights -> registers: fragment of slab s, channel block cb = 8 k-values (k = 16 s + 8 hi ..) of channel 32 (wc*CB + cb) + l31
    h16x8 wf[S][CB];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
            wf[s][cb] = *reinterpret_cast<const h16x8*>(p.w + (size_t)((wc * CB + cb) * 32 + l31) * p.kpad + s * 16 + hi * 8);

    // per-lane running statistics of the block: channels (cb, q, e) -> 32 cb + 8 q + 4 hi + e
    const bool stats = p.stat_part != nullptr;
    float s1[CB][4][4], s2[CB][4][4];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) s1[cb][q][e] = s2[cb][q][e] = 0.f;

    // lane's window base: first row of the wave, own column (and, CP = 16, own channel half)
    const int lbase = (2 * wr * RW * G::ROWP + l31 + (CP == 16 ? hi * 2 * PLANE : 0)) * 16;
    int buf = 0;
    bool first = true;
    while (true) {
        if (first) wait_vmcnt<0>();
        else wait_vmcnt<RW * CB * 2>();   // the stores of the previous patch are younger than this window's DMA
        first = false;
        __builtin_amdgcn_s_barrier();     // window `buf` complete for every wave; everybody has left window buf^1
        const int next = patch + gridDim.x;
        if (next < p.npatch) issue(next, buf ^ 1);

        f32x16 acc[RW][CB];
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][cb][e] = 0.f;
        const unsigned char* win = sm + buf * G::WBUF + lbase;
        if (CP != 16) {
            // slab = two taps (lane halves); CP = 4: two slot pairs of a kernel row (element (t & 1) * 2 + hi of window row t >> 1).  Output row i at slab s reads what output row 0 reads at slab s + 7 (two window rows =
            // 14 taps further down): ONE fragment stream F(t), t = s + 7 i, serves all rows -- 25 + 7 (RW - 1) reads for 25 RW CB
            // MFMAs.  The odd tap of the last slab (tap 49) has zero weights: any finite window element will do.
            constexpr int SPR = G::SPR;
            constexpr int T = S + SPR * (RW - 1), PF = 2;   // PF: fragments in flight ahead of the matrix cores
            constexpr int RMAX = 6 + 2 * (RW - 1);        // last window row of a wave
            auto sfrag = [&](int t) {
                if (CP == 4) return *reinterpret_cast<const h16x8*>(win + ((t >> 1) * G::ROWP + (t & 1) * 2 + hi) * 16);
                const int u0 = 2 * t, u1 = (2 * t + 1) / 7 <= RMAX ? 2 * t + 1 : 2 * t;
                return *reinterpret_cast<const h16x8*>(win + (hi ? tap_off<CP>(u1) : tap_off<CP>(u0)));
            };
            h16x8 a[PF + 1];
#pragma unroll
            for (int d = 0; d < PF; ++d) a[d] = sfrag(d);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (t + PF < T) a[(t + PF) % (PF + 1)] = sfrag(t + PF);
#pragma unroll
                for (int i = 0; i < RW; ++i) {
                    const int sl = t - SPR * i;
                    if (sl >= 0 && sl < S) {
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb) acc[i][cb] = MGN_MFMA_32x32x16(wf[sl][cb], a[t % (PF + 1)], acc[i][cb]);
                    }
                }
            }
        } else {
            // one tap per slab: the fragment of window row r = kh + 2 i and column tap kw serves every (output row i, kernel row kh)
            // pair on that window row -- 91 fragment reads for the 196 MFMAs of a patch, each consumed at once by up to four
            // independent accumulators
            constexpr int NR = 7 + 2 * (RW - 1);   // window rows a wave touches
            constexpr int PF = 3;
            auto rfrag = [&](int f) {              // f = r * 7 + kw
                const int r = f / 7, kw = f % 7;
                return *reinterpret_cast<const h16x8*>(win + (r * G::ROWP + (kw & 1) * PLANE + (kw >> 1)) * 16);
            };
            h16x8 a[PF + 1];
#pragma unroll
            for (int d = 0; d < PF; ++d) a[d] = rfrag(d);
#pragma unroll
            for (int f = 0; f < NR * 7; ++f) {
                if (f + PF < NR * 7) a[(f + PF) % (PF + 1)] = rfrag(f + PF);
                const int r = f / 7, kw = f % 7;
#pragma unroll
                for (int i = 0; i < RW; ++i) {
                    const int kh = r - 2 * i;
                    if (kh >= 0 && kh < 7) acc[i][0] = MGN_MFMA_32x32x16(wf[kh * 7 + kw][0], a[f % (PF + 1)], acc[i][0]);
                }
            }
        }

        // D = W-rows x pixels: column = lane & 31 -> output column, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) -> channel
        {
            const int n = patch / ppi, prem = patch - n * ppi, pyi = prem / p.px, pxi = prem - pyi * p.px;
            const int ox = pxi * SPW + l31;
#pragma unroll
            for (int i = 0; i < RW; ++i) {
                const int oy = pyi * SPH + wr * RW + i;
                const bool ok = ox < p.OW && oy < p.OH;
                const float msk = ok ? 1.f : 0.f;
                const int obase = ok ? (((n * p.OH + oy) * p.OW + ox) * 64 + wc * 32) * 2 : OOB;
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int qp = 0; qp < 2; ++qp) {
                        uint32_t pk[2][2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int q = 2 * qp + u;
                            pk[u][0] = mgn_pack2(acc[i][cb][q * 4 + 0], acc[i][cb][q * 4 + 1]);
                            pk[u][1] = mgn_pack2(acc[i][cb][q * 4 + 2], acc[i][cb][q * 4 + 3]);
                            if (stats) {
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    const float d0 = mgn_lo2f(pk[u][h]) * msk, d1 = mgn_hi2f(pk[u][h]) * msk;
                                    s1[cb][q][2 * h] += d0; s2[cb][q][2 * h] = fmaf(d0, d0, s2[cb][q][2 * h]);
                                    s1[cb][q][2 * h + 1] += d1; s2[cb][q][2 * h + 1] = fmaf(d1, d1, s2[cb][q][2 * h + 1]);
                                }
                            }
                        }
                        // v_permlane32_swap: lane l and lane l + 32 exchange 4-channel groups -> 8 consecutive channels per lane
                        const auto w0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                        const auto w1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                        const u32x4 v = {w0[0], w1[0], w0[1], w1[1]};
                        __builtin_amdgcn_raw_buffer_store_b128(v, rsO, obase == OOB ? OOB : obase + (cb * 32 + 16 * qp + 8 * hi) * 2, 0, 0);
                    }
            }
        }
        if (next >= p.npatch) break;
        patch = next;
        buf ^= 1;
    }

    if (stats) {
        // 16-lane DPP butterflies, then the NP = (row groups x 2 lane rows) parts of every channel through LDS in a fixed order
        auto row_sum = [](float v) {
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
            return v;
        };
        constexpr int NP = (CB == 1 ? 2 : 4) * 2;
        float* red = reinterpret_cast<float*>(sm + 2 * G::WBUF);   // [NP][64 channels][2]
        const int rw = (lane >> 4) & 1;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = row_sum(s1[cb][q][e]), b = row_sum(s2[cb][q][e]);
                    if ((lane & 15) == 0) {
                        const int ch = wc * 32 + cb * 32 + 8 * q + 4 * hi + e;
                        *reinterpret_cast<float2*>(red + ((wr * 2 + rw) * 64 + ch) * 2) = make_float2(a, b);
                    }
                }
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < NP; ++k) t += red[k * 128 + tid];
            p.stat_part[(size_t)blockIdx.x * 128 + tid] = t;
        }
    }
}

__global__ __launch_bounds__(256, 2) void conv_stem7_d4(StemParams p) { stem_body<4>(p); }
__global__ __launch_bounds__(256, 1) void conv_stem7_c8(StemParams p) { stem_body<8>(p); }
__global__ __launch_bounds__(256, 1) void conv_stem7_c16(StemParams p) { stem_body<16>(p); }

}  // namespace

extern "C" {

#ifndef MGN_F16
/* number of blocks (= statistics rows) the persistent stem kernel uses for this layer, 0 = not a layer for it */
int mgn_conv_stem7_blocks(int N, int IH, int IW, int Cin, int OH, int OW, int Cout) {
    if ((Cin != 4 && Cin != 8 && Cin != 16) || Cout != 64 || N < 1 || IH < 1 || IW < 1) return 0;
    if (Cin == 4 && ((IW & 1) || getenv("MGN_CONV_NOSTEM4"))) return 0;   // (pixel pairs: even rows; the caller then pads to 8 channels)
    if (OH != (IH + 6 - 7) / 2 + 1 || OW != (IW + 6 - 7) / 2 + 1) return 0;
    if ((size_t)N * IH * IW * Cin * 2 >= 0x7fffffffu || (size_t)N * OH * OW * 64 * 2 >= 0x7fffffffu) return 0;
    if (getenv("MGN_CONV_NOSTEM7")) return 0;
    const long np = (long)N * ((OH + SPH - 1) / SPH) * ((OW + SPW - 1) / SPW);
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    const long cap = (long)cus * (Cin == 4 ? Stem<4>::BPC : 1);
    return (int)(np < cap ? np : cap);
}
#endif

/* 7x7 / stride 2 / pad 3, Cin = 4 | 8 | 16 (channel-padded), Cout = 64: see the header of this file.  MGN_ENOTSUP for other shapes. */
int MGN_SYM(mgn_conv_stem7)(const void* in, const void* w_packed, void* out, int N, int IH, int IW, int Cin, int OH, int OW, int Cout,
                            float* stat_partials, void* stream) {
    const int blocks = mgn_conv_stem7_blocks(N, IH, IW, Cin, OH, OW, Cout);
    if (blocks <= 0) return MGN_ENOTSUP;
    if (!in || !w_packed || !out) return MGN_EINVAL;
    StemParams p;
    p.in = (const uint16_t*)in; p.w = (const uint16_t*)w_packed; p.out = (uint16_t*)out; p.stat_part = stat_partials;
    p.N = N; p.IH = IH; p.IW = IW; p.OH = OH; p.OW = OW;
    p.py = (OH + SPH - 1) / SPH; p.px = (OW + SPW - 1) / SPW; p.npatch = N * p.py * p.px;
    p.kpad = (49 * Cin + 31) / 32 * 32;
    hipStream_t st = (hipStream_t)stream;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem7_d4), hipFuncAttributeMaxDynamicSharedMemorySize, Stem<4>::LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem7_c8), hipFuncAttributeMaxDynamicSharedMemorySize, Stem<8>::LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem7_c16), hipFuncAttributeMaxDynamicSharedMemorySize, Stem<16>::LDS);
        attr = true;
    }
    if (Cin == 4) hipLaunchKernelGGL(conv_stem7_d4, dim3(blocks), dim3(256), Stem<4>::LDS, st, p);
    else if (Cin == 8) hipLaunchKernelGGL(conv_stem7_c8, dim3(blocks), dim3(256), Stem<8>::LDS, st, p);
    else hipLaunchKernelGGL(conv_stem7_c16, dim3(blocks), dim3(256), Stem<16>::LDS, st, p);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

}  // extern "C"

// conv_win.hip -- 3x3 / stride 1 / pad 1 convolutions with >= 128 output channels as a WINDOWED implicit GEMM (gfx950).
//
// Replaces, for the 128/256/512-channel layers of mgnet/modeling/res_net.py:28-60 (BasicBlock conv1/conv2 of res3..res5) and
// layers.py:53-72,110-118,201-210,283-311 (decoder / head 3x3 convs), the generic implicit GEMM of conv.hip.  Same contract as
// mgn_conv_igemm (NHWC 16-bit activations, weights [Cout][3][3][Cin], fp32 accumulation, optional 16-bit residual); the data
// gradient of those layers is this kernel on the flipped / transposed weights.
//
// Why: the implicit GEMM gathers the A operand once per TAP (every input pixel travels L2 -> LDS nine times) and the
// global -> LDS fill rate of a CU (~12 B/clk) bounds it at ~36 % MFMA utilisation for a 256 x 256 tile, ~25 % for the 128-channel
// layers.  Here a block owns a 2-D PATCH of PH x 32 output pixels and keeps the (PH+2) x 34 input window
// of one 32-channel chunk in LDS for all nine taps: a tap is an address offset into the window, so A is fetched ~1.2x instead
// of 9x.  With A nearly free the tile is tall and narrow -- 512 pixels x 128 output channels -- which halves the weight bytes
// per flop as well: 8 KB of weights + 4.4 KB of window per k-step (512 x 128 x 32 MACs) instead of 32 KB.
//
//   block : 8 waves = 4 (pixel rows) x 2 (64 output channels); wave tile (PH/4 rows x 32 px) x 64 co = PH/4 x 2 MFMA tiles
//           (v_mfma_f32_32x32x16, operands swapped: D rows = output channels, so a lane holds 4 consecutive channels of a pixel)
//   k loop: channel chunk outer (32 channels), tap inner; one barrier per k-step
//   LDS   : window [2 buffers][640 px][64 B] (pixel-major, 16-byte slots XOR-swizzled by (px >> 2) & 3: every fragment read is
//           conflict-free for ANY tap shift), weights [8 stages][128 co][64 B]; everything arrives by LDS-DMA
//           (buffer_load_dwordx4 ... lds, zero padding from the buffer bounds check), the next chunk's window is fetched while the
//           nine taps of the current one run; counted vmcnt waits.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "mgnet_hip.h"

namespace {

#include "h16.h"
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct WinParams {
    const uint16_t* in;        // [N, H, W, Cin]
    const uint16_t* w;         // [Cout, 3, 3, Cin]
    uint16_t* out;             // [N, H, W, Cout]
    const uint16_t* residual;  // [N, H, W, Cout] added before rounding, or null
    int N, H, W, Cin, Cout;
    float* stat_part;          // [N*py*px][Cout][2] per-patch sums of (r - shift), (r - shift)^2 of the rounded outputs, or null
    const float* stat_shift;   // [Cout] or null (= 0)
    int py, px;                // patches per image (rows, columns)
    int xcd;                   // 1: deal contiguous bands of patches to the XCDs
    const float* bias;         // [Cout] added before the residual, or null (the eval-mode fold of the InPlaceABNSync that follows, see
    int act;                   // mgn_conv_igemm_act); act after bias and residual: 0 none, 1 ReLU, 2 leaky ReLU with `slope`
    float slope;
};
MGN_PLAN_RO_CONV(WinParams, MGN_RO(in) MGN_RO(w) MGN_RO(residual) MGN_RO(stat_shift) MGN_RO(bias))   // pointers the kernels only read through (launch-plan dependency analysis, csrc/mgn_launch.h)

template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

constexpr int PW = 32, WW = PW + 2;
constexpr int WIN_PIECES = 40;                 // 1-KB DMA pieces of 16 pixels x 64 B: 5 per wave (612 of the 640 pixels are real)
constexpr int WIN_BYTES = WIN_PIECES * 1024;
constexpr int WST_BYTES = 128 * 64;            // one weight stage: 128 output channels x 32 input channels
constexpr int NWST = 8;                        // weight stages (power of two)
constexpr int PD = 6;                          // W(s + PD) is issued at k-step s (PD + 1 <= NWST)
constexpr int WIN_LDS = 2 * WIN_BYTES + NWST * WST_BYTES;

// MGN_WIN_DBG (compile-time, race hunting only; tools/dbg_win_race.py): 1 = every counted wait becomes vmcnt(0); 2 = all of a wave's LDS
// reads have returned before it enters a barrier; 3 = both
#ifndef MGN_WIN_DBG
#define MGN_WIN_DBG 0
#endif
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if (MGN_WIN_DBG & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    if (MGN_WIN_DBG & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ int xcd_tile(int b, int nb) {   // block b -> XCD b % 8; XCD k works on a contiguous band of tiles
    const int k = b & 7, j = b >> 3, q = nb >> 3, r = nb & 7;
    return k * q + (k < r ? k : r) + j;
}

// PH = 16: patch 16 x 32 (4 rows per wave); PH = 8: patch 8 x 32 (2 rows per wave) for layers with few pixels -- 8 waves each (NWM = 4
// pixel-row groups x 2 channel halves), one block per CU.
// NWM = 2 (round 4, `conv3x3_win8f`): the 8 x 32 patch with FOUR waves of 4 rows x 32 px x 64 co (the 16-row kernel's wave tile) in
// 80 KB of LDS, so that TWO independent blocks share a CU, one wave of each per SIMD: what a lone 8-wave block spends outside its
// k loop (~10 of ~62 us: first window + weight stages, epilogue, block turn-around) runs beside the other
// block's MFMAs, and a wave stuck issuing LDS-DMA pieces leaves the matrix pipe to its neighbour.  Per k-step a wave issues
// [window piece, while the next chunk has one][two weight pieces]; one barrier per k-step (the two waves of a SIMD are not in the same block, so they do
// not sit in the same bubble), weight ring of 4 stages issued 3 steps ahead.
template <int PH, int NWM>
__device__ __forceinline__ void conv_win_body(const WinParams& p) {
    constexpr int NW = 2 * NWM;               // waves
    constexpr int RPW = PH / NWM;             // pixel rows (= MFMA tiles) per wave
    constexpr int WPX = (PH + 2) * WW;        // real window pixels
    constexpr int NWP = (WPX + 16 * NW - 1) / (16 * NW);   // window pieces per wave and chunk (5 for PH = 16, 3 for PH = 8; 6 for 4 waves)
    constexpr int WINB = NWM == 4 ? WIN_BYTES : NWP * NW * 1024;          // one window buffer
    constexpr int NST = NWM == 4 ? NWST : 4, PDW = NWM == 4 ? PD : 3;     // weight stages, prefetch distance in k-steps
    constexpr int WPS = 8 / NW;               // weight pieces per wave and k-step
    constexpr int BAR2 = NWM == 4 ? 1 : 0;    // barrier every second k-step (8 waves) / every k-step (4 waves)
    constexpr int KEEP = BAR2 ? 3 : 1;        // k-steps whose loads may still be in flight behind a wait (everything older has landed)
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    unsigned char* const winb = wsm;
    unsigned char* const wst = wsm + 2 * WINB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5, l31 = lane & 31;
    const int patch = p.xcd ? xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x, bn = blockIdx.y;
    const int ppi = p.py * p.px;
    const int n = patch / ppi, prem = patch - n * ppi, pyi = prem / p.px, pxi = prem - pyi * p.px;
    const int y0 = pyi * PH, x0 = pxi * PW;

    const uint32_t in_bytes = (uint32_t)((size_t)p.N * p.H * p.W * p.Cin * 2), w_bytes = (uint32_t)((size_t)p.Cout * 9 * p.Cin * 2);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.in), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0, w_bytes, 0x00020000);
    constexpr int OOB = (int)0x80000000;
    typedef __attribute__((address_space(3))) void* lds_ptr;

    // window pieces of this wave: piece q = wave + 8 i covers window pixels 16 q .. 16 q + 15 (lane >> 2), 16-byte slot lane & 3
    int wvoff[NWP];
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
        const int pp = (wave + NW * i) * 16 + (lane >> 2);
        const int sseg = (lane & 3) ^ ((pp >> 2) & 3);    // source segment that lands in slot lane & 3
        const int wy = pp / WW, wx = pp - wy * WW;
        const int iy = y0 - 1 + wy, ix = x0 - 1 + wx;
        const bool ok = pp < WPX && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        wvoff[i] = ok ? (((n * p.H + iy) * p.W + ix) * p.Cin + sseg * 8) * 2 : OOB;
    }
    int wbase[WPS];
#pragma unroll
    for (int k = 0; k < WPS; ++k) {
        const int row = (wave * WPS + k) * 16 + (lane >> 2);
        const int sseg = (lane & 3) ^ ((row >> 2) & 3);
        wbase[k] = ((bn * 128 + row) * 9 * p.Cin + sseg * 8) * 2;
    }
    const int nch = p.Cin / 32;
    auto issue_w = [&](int c, int t, int stage) {
#pragma unroll
        for (int k = 0; k < WPS; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(wst + stage * WST_BYTES + (wave * WPS + k) * 1024), 16, wbase[k],
                                                     (t * p.Cin + c * 32) * 2, 0, 0);
    };
    auto issue_win = [&](int c, int i, int buf) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(winb + buf * WINB + (wave + NW * i) * 1024), 16, wvoff[i], c * 64, 0, 0);
    };

    // fragment byte offsets (kk = 0; kk = 1 flips bit 5): A from the window at (row 4*wm*RPW/4.. + dy, column l31 + kw)
    int aoff[RPW + 2][3];
#pragma unroll
    for (int dy = 0; dy < RPW + 2; ++dy)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int pp = (wm * RPW + dy) * WW + l31 + kw;
            aoff[dy][kw] = pp * 64 + ((hi ^ ((pp >> 2) & 3)) << 4);
        }
    int boff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wn * 64 + j * 32 + l31;
        boff[j] = row * 64 + ((hi ^ ((row >> 2) & 3)) << 4);
    }

    f32x16 acc[RPW][2];
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Pipeline.  k-step s issues the LDS-DMA loads [window piece s of the next chunk, while it has one][W(s + PDW), while there is one];
    // the wait in front of a barrier retires every load issued KEEP or more k-steps ago, i.e. it leaves exactly the loads of the last
    // KEEP steps in flight -- a compile-time count per (tap, last chunk or not): WinCount::younger().  8 waves: barriers stand at the even
    // taps of a chunk (5 per 9 k-steps: with a barrier per step both waves of a SIMD sit in the same bubble), KEEP = 3: the wait of even
    // step s retires W(s + 1), W(s + 2); the two steps up to the next barrier read W(s), W(s + 1) and -- the first half (kk = 0) of the
    // NEXT step's fragments is fetched from LDS while the second half of a step's MFMAs run -- W(s + 2).  4 waves: a barrier per step,
    // KEEP = 1.  Ring hazards: step s writes the stage of step s + PDW - NST and, at taps 0 .. NWP-1, the window buffer last read in
    // the previous chunk; a barrier separates both from their last readers (tap 0 always has one).
    // (Round 6: until then every step issued a CONSTANT number of loads, padded with out-of-range "dummy" loads into a scratch KB, and
    //  every wait used one constant.  The compiler merges adjacent identical dummy loads -- stores to the same LDS bytes with nothing it
    //  can see reading them in between -- so the last taps of the last chunk issued ONE load instead of three, the constant wait then
    //  retired too little, and the last weight stages could be read before they had landed: rare wrong tiles under memory contention,
    //  found by tools/dbg_win_race.py at the benchmark's shapes.  No dummies any more: the counts are exact.)
    struct WinCount {
        static constexpr int loads(int t, bool last) {   // loads a wave issues at tap t (t < 0: the previous chunk's last taps / the prologue)
            return t < 0 ? WPS : ((!last && t < NWP) ? 1 : 0) + ((!last || t + PDW <= 8) ? WPS : 0);
        }
        static constexpr int younger(int t, bool last) {
            int n = 0;
            for (int j = 1; j <= KEEP; ++j) n += loads(t - j, last);
            return n;
        }
    };
    static_assert(NWP <= 9 - KEEP, "the previous chunk's last KEEP taps must carry weight pieces only");
#pragma unroll
    for (int i = 0; i < NWP; ++i) issue_win(0, i, 0);
#pragma unroll
    for (int k = 0; k < PDW; ++k) issue_w(0, k, k);   // what steps -PDW .. -1 would have issued (ksteps >= 9 > PDW)
    // (8 waves: W(0..2) landed; 4 waves: W(0), W(1) landed -- the first step prefetches W(1))
    wait_vmcnt<WPS * KEEP>();
    __builtin_amdgcn_s_barrier();

    auto load_frags = [&](const unsigned char* win, const unsigned char* ws, int kh, int kw, int kk, h16x8 (&a)[RPW], h16x8 (&b)[2]) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) a[i] = *reinterpret_cast<const h16x8*>(win + (aoff[i + kh][kw] ^ (kk << 5)));
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const h16x8*>(ws + (boff[j] ^ (kk << 5)));
    };
    auto mma = [&](const h16x8 (&a)[RPW], const h16x8 (&b)[2]) {
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = MGN_MFMA_32x32x16(b[j], a[i], acc[i][j]);
    };

    h16x8 a0[RPW], b0[2], a1[RPW], b1[2];
    load_frags(winb, wst, 0, 0, 0, a0, b0);
    int stage = 0;   // stage of k-step s = s % NWST
    auto chunk = [&](int c, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const unsigned char* win = winb + (c & 1) * WINB;
        const unsigned char* win_next = winb + ((c + 1) & 1) * WINB;
        static_for<9>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (!BAR2 || t % 2 == 0) {   // 8 waves: one barrier per TWO k-steps (taps 0|1, 2|3, 4|5, 6|7, 8): see the pipeline comment
                wait_vmcnt<WinCount::younger(t, LAST)>();
                __builtin_amdgcn_s_barrier();
            }
            if constexpr (!LAST && t < NWP) issue_win(c + 1, t, (c + 1) & 1);
            if constexpr (!LAST || t + PDW <= 8) {   // (a chunk that is not the last always has a next one: c + 1 < nch)
                constexpr int tn = t + PDW >= 9 ? t + PDW - 9 : t + PDW;
                issue_w(t + PDW >= 9 ? c + 1 : c, tn, (stage + PDW) & (NST - 1));
            }
            const int kh = t / 3, kw = t - kh * 3;
            const unsigned char* ws = wst + stage * WST_BYTES;
            const unsigned char* ws_next = wst + ((stage + 1) & (NST - 1)) * WST_BYTES;
            // (sched_barrier: keep the two fragment sets in separate registers and each batch of LDS reads a full batch of MFMAs
            //  ahead of its use -- left alone, the scheduler re-serialises read -> wait -> MFMA to save registers)
            load_frags(win, ws, kh, kw, 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (t < 8) load_frags(win, ws_next, (t + 1) / 3, (t + 1) % 3, 0, a0, b0);
            else if (!LAST) load_frags(win_next, ws_next, 0, 0, 0, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            stage = (stage + 1) & (NST - 1);
        });
    };
    for (int c = 0; c + 1 < nch; ++c) chunk(c, std::false_type{});
    chunk(nch - 1, std::true_type{});
    wait_vmcnt<0>();   // (the dummies of the last steps still write their zeros: nothing may land after the block has ended)

    // D = W-rows x pixels: column = lane & 31 -> pixel of the row, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) -> output channel
    const int ox = x0 + l31;
    // optional per-channel statistics of the ROUNDED output (what the following InPlaceABNSync normalises with): a lane sums
    // (r - shift) and (r - shift)^2 of its 32 channels over its RPW pixels; see the reduction below the stores
    const bool stats = p.stat_part != nullptr;
    float s1[2][4][4], s2[2][4][4], sh[2][4][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (stats && p.stat_shift) t = *reinterpret_cast<const float4*>(p.stat_shift + bn * 128 + wn * 64 + j * 32 + 8 * q + 4 * hi);
            if (!stats && p.bias) t = *reinterpret_cast<const float4*>(p.bias + bn * 128 + wn * 64 + j * 32 + 8 * q + 4 * hi);   // (never both)
            sh[j][q][0] = t.x; sh[j][q][1] = t.y; sh[j][q][2] = t.z; sh[j][q][3] = t.w;
#pragma unroll
            for (int e = 0; e < 4; ++e) s1[j][q][e] = s2[j][q][e] = 0.f;
        }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int oy = y0 + wm * RPW + i;
        if (oy >= p.H) break;   // wave-uniform
        const bool ok = ox < p.W;
        const size_t m = ((size_t)n * p.H + oy) * p.W + (ok ? ox : 0);
        uint16_t* opix = p.out + m * p.Cout + bn * 128 + wn * 64;
        // v_permlane32_swap exchanges the 4-channel groups of lane l and lane l + 32: every lane then owns 8 consecutive channels of its
        // pixel (16-byte stores, half the store instructions).  An optional residual is read with ONE 16-byte load in that same layout
        // and brought back to the accumulator layout by the same exchange (it is its own inverse), then added in fp32 before the
        // rounding -- the 8-byte loads / stores this replaces cost four times the memory instructions.
        const float msk = ok ? 1.f : 0.f;
        const uint16_t* rpix = p.residual ? p.residual + m * p.Cout + bn * 128 + wn * 64 : nullptr;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                uint32_t rp[2][2] = {{0u, 0u}, {0u, 0u}};
                if (p.residual) {
                    const uint4 R = *reinterpret_cast<const uint4*>(rpix + j * 32 + 16 * qp + 8 * hi);
                    const auto u0 = __builtin_amdgcn_permlane32_swap(R.x, R.z, false, false);
                    const auto u1 = __builtin_amdgcn_permlane32_swap(R.y, R.w, false, false);
                    rp[0][0] = u0[0]; rp[1][0] = u0[1]; rp[0][1] = u1[0]; rp[1][1] = u1[1];
                }
                uint32_t pk[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int q = 2 * qp + u;
                    float v0 = acc[i][j][q * 4 + 0], v1 = acc[i][j][q * 4 + 1], v2 = acc[i][j][q * 4 + 2], v3 = acc[i][j][q * 4 + 3];
                    if (!stats && p.bias) { v0 += sh[j][q][0]; v1 += sh[j][q][1]; v2 += sh[j][q][2]; v3 += sh[j][q][3]; }
                    if (p.residual) {
                        v0 += mgn_lo2f(rp[u][0]); v1 += mgn_hi2f(rp[u][0]);
                        v2 += mgn_lo2f(rp[u][1]); v3 += mgn_hi2f(rp[u][1]);
                    }
                    if (p.act) {
                        const float sl = p.act == 1 ? 0.f : p.slope;
                        v0 = v0 > 0.f ? v0 : v0 * sl; v1 = v1 > 0.f ? v1 : v1 * sl; v2 = v2 > 0.f ? v2 : v2 * sl; v3 = v3 > 0.f ? v3 : v3 * sl;
                    }
                    pk[u][0] = mgn_pack2(v0, v1);
                    pk[u][1] = mgn_pack2(v2, v3);
                    if (stats) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const float d0 = (mgn_lo2f(pk[u][h]) - sh[j][q][2 * h]) * msk, d1 = (mgn_hi2f(pk[u][h]) - sh[j][q][2 * h + 1]) * msk;
                            s1[j][q][2 * h] += d0; s2[j][q][2 * h] = fmaf(d0, d0, s2[j][q][2 * h]);
                            s1[j][q][2 * h + 1] += d1; s2[j][q][2 * h + 1] = fmaf(d1, d1, s2[j][q][2 * h + 1]);
                        }
                    }
                }
                const auto w0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                const auto w1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                if (ok) *reinterpret_cast<uint4*>(opix + j * 32 + 16 * qp + 8 * hi) = make_uint4(w0[0], w1[0], w0[1], w1[1]);
            }
    }
    if (stats) {
        // 16-lane butterflies (DPP: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror): every lane of a 16-lane row
        // then holds the row's sum; lanes 0 / 16 / 32 / 48 park it in LDS (rows 0,1 = channels 4*hi.., two pixel halves), and
        // after a block barrier 256 threads add the 8 parts (4 pixel waves x 2 rows) of one (channel, moment) each in a fixed
        // order and store the block's partial row: stat_part[patch][Cout][2].  The window memory is free by then.
        auto row_sum = [](float v) {
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
            return v;
        };
        float* red = reinterpret_cast<float*>(wsm);   // [wm 4][row 2][128 channels][2]
        __syncthreads();                               // every wave has left the k loop: the window buffers are dead
        const int rw = (lane >> 4) & 1;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = row_sum(s1[j][q][e]), b = row_sum(s2[j][q][e]);
                    if ((lane & 15) == 0) {
                        const int ch = wn * 64 + j * 32 + 8 * q + 4 * hi + e;
                        *reinterpret_cast<float2*>(red + ((wm * 2 + rw) * 128 + ch) * 2) = make_float2(a, b);
                    }
                }
        __syncthreads();
        if (tid < 256) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 2 * NWM; ++k) t += red[k * 256 + tid];
            p.stat_part[((size_t)patch * p.Cout + bn * 128) * 2 + tid] = t;
        }
    }
}

constexpr int WIN8F_LDS = 2 * 24 * 1024 + 4 * WST_BYTES;   // 80 KB: two blocks per CU
__global__ __launch_bounds__(512, 1) void conv3x3_win16(WinParams p) { conv_win_body<16, 4>(p); }
__global__ __launch_bounds__(512, 1) void conv3x3_win8(WinParams p) { conv_win_body<8, 4>(p); }
__global__ __launch_bounds__(256, 2) void conv3x3_win8f(WinParams p) { conv_win_body<8, 2>(p); }

}  // namespace

extern "C" {

#ifndef MGN_F16
/* patch height the dispatcher uses for this layer: 16-row patches when they fill the chip, 8-row patches for the low-resolution
 * layers, 0 = not a layer for this kernel (MGN_CONV_WIN = "0" | "8" | "16" overrides: tests, A/B runs) */
int mgn_conv_win_patch_rows(int N, int OH, int OW, int Cin, int Cout) {
    if (N < 1 || OH < 1 || OW < 1 || Cin < 32 || Cin % 32 != 0 || Cin == 64 || Cout < 128 || Cout % 128 != 0) return 0;
    if ((size_t)N * OH * OW * (Cin > Cout ? Cin : Cout) * 2 >= 0x7fffffffu) return 0;
    const char* ewin = getenv("MGN_CONV_WIN");
    if (ewin) {
        const int pr = atoi(ewin);
        return pr == 8 || pr == 16 ? pr : 0;
    }
    const long pc = (long)N * ((OW + 31) / 32) * (Cout / 128);
    if (pc * ((OH + 7) / 8) >= 1024) return 8;   // enough 8-row patches for two rounds of the two-blocks-per-CU kernel (conv3x3_win8f)
    return pc * ((OH + 15) / 16) >= 200 ? 16 : (pc * ((OH + 7) / 8) >= 64 ? 8 : 0);
}
#endif

int MGN_SYM(mgn_conv3x3_win_act)(const void* in, const void* w, void* out, int N, int H, int W, int Cin, int Cout, const void* residual,
                                 int patch_rows, float* stat_partials, const float* stat_shift, const float* bias, int act, float slope,
                                 void* stream);
int MGN_SYM(mgn_conv3x3_win)(const void* in, const void* w, void* out, int N, int H, int W, int Cin, int Cout, const void* residual,
                             int patch_rows, float* stat_partials, const float* stat_shift, void* stream) {
    return MGN_SYM(mgn_conv3x3_win_act)(in, w, out, N, H, W, Cin, Cout, residual, patch_rows, stat_partials, stat_shift, nullptr, 0, 0.f, stream);
}

/* the same with an epilogue out = act(conv + bias + residual) (act: 0 none, 1 ReLU, 2 leaky ReLU with `slope`): inference with the
 * InPlaceABNSync that follows folded into the weights and this bias (mgn_conv_igemm_act).  Not together with the statistics rows. */
int MGN_SYM(mgn_conv3x3_win_act)(const void* in, const void* w, void* out, int N, int H, int W, int Cin, int Cout, const void* residual,
                                 int patch_rows, float* stat_partials, const float* stat_shift, const float* bias, int act, float slope,
                                 void* stream) {
    if (!in || !w || !out || N < 1 || H < 1 || W < 1 || act < 0 || act > 2) return MGN_EINVAL;
    if (stat_partials && (bias || act)) return MGN_ENOTSUP;
    if (Cin < 32 || Cin % 32 != 0 || Cout < 128 || Cout % 128 != 0) return MGN_ENOTSUP;
    if ((size_t)N * H * W * (Cin > Cout ? Cin : Cout) * 2 >= 0x7fffffffu) return MGN_ENOTSUP;   // 32-bit byte offsets
    if (patch_rows != 8 && patch_rows != 16) return MGN_EINVAL;
    WinParams p;
    p.in = (const uint16_t*)in; p.w = (const uint16_t*)w; p.out = (uint16_t*)out; p.residual = (const uint16_t*)residual;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.stat_part = stat_partials; p.stat_shift = stat_shift;
    p.bias = bias; p.act = act; p.slope = slope;
    if (stat_partials && residual) return MGN_ENOTSUP;
    p.py = (H + patch_rows - 1) / patch_rows; p.px = (W + PW - 1) / PW;
    const long npatch = (long)N * p.py * p.px;
    if (npatch > 0x7fffffffL) return MGN_EINVAL;
    p.xcd = (npatch >= 16 && !getenv("MGN_CONV_NOXCD")) ? 1 : 0;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_win16), hipFuncAttributeMaxDynamicSharedMemorySize, WIN_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_win8), hipFuncAttributeMaxDynamicSharedMemorySize, WIN_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_win8f), hipFuncAttributeMaxDynamicSharedMemorySize, WIN8F_LDS);
        attr = true;
    }
    const dim3 grid((unsigned)npatch, (unsigned)(Cout / 128));
    // 8-row patches: the 4-wave kernel (two blocks per CU) from 1024 blocks on, the 8-wave kernel for the low-resolution layers whose
    // few blocks do not fill the chip twice (measured, profiles/r04_conv_win8f.txt); MGN_CONV_WIN8F = 0 | 1 forces one (tests, A/B)
    const char* e8 = getenv("MGN_CONV_WIN8F");
    const int fat8 = e8 ? atoi(e8) : (npatch * (Cout / 128) >= 1024);
    if (patch_rows == 16) hipLaunchKernelGGL(conv3x3_win16, grid, dim3(512), WIN_LDS, (hipStream_t)stream, p);
    else if (fat8) hipLaunchKernelGGL(conv3x3_win8f, grid, dim3(256), WIN8F_LDS, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(conv3x3_win8, grid, dim3(512), WIN_LDS, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

}  // extern "C"

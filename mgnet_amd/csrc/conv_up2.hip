// conv_up2.hip -- data gradient of the 3x3 / stride 2 / pad 1 convolutions as a WINDOWED implicit GEMM over the low-resolution
// gradient (gfx950).
//
// Replaces, for the first conv of the down-sampling BasicBlocks (mgnet/modeling/res_net.py:28-60 with stride 2: res3 / res4 / res5
// conv1), the parity-class launch of the generic implicit GEMM (conv.hip `conv_igemm_glds`, up = 2): same tensors and the same sums
// as mgn_conv_igemm(stride 1, pad 1, up 2) on the flipped / transposed weights [Cin_fwd][3][3][Cout_fwd].
//
// Why: an output pixel (2i + a, 2j + b) of the data gradient meets only the taps whose parity matches -- 1, 2, 2 or 4 of the 9 --
// so as four separate dense convolutions (one per parity class) the K loops are 4 .. 16 steps long and a block spends its life
// in prologue / epilogue: 319-532 TFLOP/s, 1.4 TB/s for a layer whose HBM floor is three times lower (profiles/r04_conv_table.txt).
// Here a block owns a patch of 8 x 32 LOW-resolution pixels = 16 x 64 output pixels and keeps the 9 x 33 gradient window of one
// 32-channel chunk in LDS: the nine (tap, shift) pairs of the four classes are nine MFMA steps on that window, exactly the shape of
// a 3x3 window kernel's k loop, accumulating into four accumulator sets.  All nine weight taps of the chunk travel with the window
// (36 KB for a 64-channel output tile): window one chunk ahead (two buffers), weights two chunks ahead (three stages), one counted
// vmcnt + barrier per chunk, the LDS-DMA of a chunk issued by one wave of each SIMD while the other computes (roles alternate).  Blocks are persistent --
// work item = (patch, 64-channel output tile), dealt round-robin -- so the first chunk of the next item is in flight while the
// epilogue of the current one stores; nothing but the first fetch of a block is exposed.
//
//   block : 8 waves = 4 (pairs of low-res pixel rows) x 2 (32 output channels); a wave holds 2 rows x 4 classes x (32 px x 32 co)
//   LDS   : 2 x window 19 KB [304 px][64 B] + 3 x weights 36 KB [9 taps][64 co][64 B] = 146 KB, everything by LDS-DMA, 16-byte slots
//           XOR-swizzled by (row >> 2) & 3 like conv_win.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mgnet_hip.h"

namespace {

#include "h16.h"
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct Up2Params {
    const uint16_t* in;        // [N, H, W, Cin]    gradient of the strided conv's output
    const uint16_t* w;         // [Cout, 3, 3, Cin] flipped / transposed forward weights (mgn_weight_layout mode 1)
    uint16_t* out;             // [N, OH, OW, Cout] OH in {2H - 1, 2H}
    const uint16_t* residual;  // [N, OH, OW, Cout] added before rounding, or null
    int N, H, W, Cin, Cout, OH, OW;
    int py, px, cot;           // patches per image (rows, columns), 64-channel output tiles
    int nitems;                // N * py * px * cot
};
MGN_PLAN_RO_CONV(Up2Params, MGN_RO(in) MGN_RO(w) MGN_RO(residual))   // pointers the kernels only read through (launch-plan dependency analysis, csrc/mgn_launch.h)

constexpr int PH = 8, PW = 32, RPW = 2, WW = PW + 1, WPX = (PH + 1) * WW;   // 297 window pixels
constexpr int NWP = (WPX + 15) / 16;                                          // 19 window pieces of 16 pixels
constexpr int WINB = NWP * 1024;                                              // 19 KB
constexpr int NWS = 3;                                                        // weight stages: the weights travel TWO steps ahead
constexpr int LWP = (NWP + 3) / 4;                                            // window pieces per LOADER wave and step
// KS = 3: the 3x3 conv's gradient; KS = 1: the 1x1 / stride-2 shortcut conv's gradient (res_net.py:52-60 `downsample`) -- one tap, only
// the class (0, 0) pixels receive a product, the other three quarters of the output are zeros (+ residual) written by the same pass
template <int KS> struct Up2 {
    static constexpr int NT = KS * KS, NWT = NT * 4, WTB = NWT * 1024;        // weight pieces per chunk: taps x 64 channels x 64 B
    static constexpr int LWT = NWT / 4;                                       // weight pieces per loader wave and step (9 | 1)
    static constexpr int LDS = 2 * WINB + NWS * WTB;                          // 146 KB | 50 KB
};

// the nine MFMA steps of a chunk: (class = 2 a + b, window shift (sy, sx), tap kh' * 3 + kw' of the flipped weights).  Output row
// 2 i + a reads the zero-upsampled gradient at row 2 i + a - 1 + kh' = 2 (i + sy): a = 0 -> kh' = 1 (sy 0); a = 1 -> kh' = 0 (sy 0),
// kh' = 2 (sy 1); columns alike.
struct Step { int cls, sy, sx, tap; };
__device__ constexpr Step STEP1[1] = {{0, 0, 0, 0}};
__device__ constexpr Step STEPS[9] = {{0, 0, 0, 4}, {1, 0, 0, 3}, {1, 0, 1, 5}, {2, 0, 0, 1}, {2, 1, 0, 7},
                                      {3, 0, 0, 0}, {3, 0, 1, 2}, {3, 1, 0, 6}, {3, 1, 1, 8}};

// RES: 0 none, 1 a full-resolution tensor added to every output pixel, 2 a LOW-resolution tensor [N, H, W, Cout] added to the class (0, 0)
// pixels only -- the data gradient of the block's 1x1 / stride-2 shortcut conv, which is non-zero exactly there (its zeros are never written
// nor read back: ops._ConvFn with_skip = 2)
template <int RES, int KS>
__device__ __forceinline__ void up2_body(const Up2Params& p) {
    constexpr int NT = Up2<KS>::NT, WTB = Up2<KS>::WTB, LWT = Up2<KS>::LWT;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5, l31 = lane & 31;
    if ((int)blockIdx.x >= p.nitems) return;
    const uint32_t in_bytes = (uint32_t)((size_t)p.N * p.H * p.W * p.Cin * 2), w_bytes = (uint32_t)((size_t)p.Cout * NT * p.Cin * 2);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.in), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0, w_bytes, 0x00020000);
    constexpr int OOB = (int)0x80000000;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int nch = p.Cin / 32, ppi = p.py * p.px;

    // ---- loader ----------------------------------------------------------------------------------------------------------------------
    // Waves w and w + 4 share a SIMD.  Per step ONE of the two issues the LDS-DMA of both (pieces lw + 4 k, lw = wave & 3) while the
    // other starts on the matrix cores at once; the roles swap every step.  (All eight waves issuing their share behind the barrier
    // leaves the matrix pipe idle for the ~1900 clk the CU needs to issue a chunk's 55 pieces: measured 34 % MFMA time with 4-row
    // patches.)
    const int lw = wave & 3, grp = wave >> 2;
    int wvoff[LWP], wboff[LWT];
    auto decode = [&](int item, int& n, int& y0, int& x0, int& bn) {
        bn = item % p.cot;
        const int patch = item / p.cot;
        n = patch / ppi;
        const int prem = patch - n * ppi, pyi = prem / p.px;
        y0 = pyi * PH;
        x0 = (prem - pyi * p.px) * PW;
    };
    auto set_window = [&](int item) {
        int n, y0, x0, bn;
        decode(item, n, y0, x0, bn);
#pragma unroll
        for (int i = 0; i < LWP; ++i) {
            const int pp = (lw + 4 * i) * 16 + (lane >> 2);
            const int sseg = (lane & 3) ^ ((pp >> 2) & 3);      // source segment that lands in slot lane & 3
            const int wy = pp / WW, wx = pp - wy * WW;
            const int iy = y0 + wy, ix = x0 + wx;
            const bool ok = pp < WPX && iy < p.H && ix < p.W;
            wvoff[i] = ok ? (((n * p.H + iy) * p.W + ix) * p.Cin + sseg * 8) * 2 : OOB;
        }
    };
    auto set_weights = [&](int bn) {
#pragma unroll
        for (int k = 0; k < LWT; ++k) {
            const int idx = lw + 4 * k, tap = idx >> 2, row = (idx & 3) * 16 + (lane >> 2);
            const int sseg = (lane & 3) ^ ((row >> 2) & 3);
            wboff[k] = (((bn * 64 + row) * NT + tap) * p.Cin + sseg * 8) * 2;
        }
    };
    unsigned char* const wts = sm + 2 * WINB;
    auto issue_window = [&](int c, int wbuf) {
#pragma unroll
        for (int i = 0; i < LWP; ++i)
            if (lw + 4 * i < NWP)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(sm + wbuf * WINB + (lw + 4 * i) * 1024), 16, wvoff[i], c * 64, 0, 0);
    };
    auto issue_weights = [&](int c, int stage) {
#pragma unroll
        for (int k = 0; k < LWT; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(wts + stage * WTB + (lw + 4 * k) * 1024), 16, wboff[k], c * 64, 0, 0);
    };

    // ---- fragment byte offsets (kk = 0; kk = 1 flips bit 5): window rows RPW * wm + 0 .. RPW, columns l31 + 0 | 1 -------------------
    int aoff[RPW + 1][2];
#pragma unroll
    for (int dy = 0; dy < RPW + 1; ++dy)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            const int pp = (wm * RPW + dy) * WW + l31 + sx;
            aoff[dy][sx] = pp * 64 + ((hi ^ ((pp >> 2) & 3)) << 4);
        }
    const int brow = wn * 32 + l31;
    const int boff = brow * 64 + ((hi ^ ((brow >> 2) & 3)) << 4);

    f32x16 acc[RPW][4];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][k][e] = 0.f;

    // step = (item, chunk); items of this block = blockIdx.x + k * gridDim.x.  Two issue cursors that every wave keeps (either wave of a
    // SIMD can be the loader of a step): the window one step ahead of the compute cursor, the weights two steps ahead (a loader's
    // LDS-DMA is issued in front of its OWN MFMAs of the step, so the pieces it issues last would have less than a step to land;
    // measured on the C4 shapes: the same time as with everything one step ahead -- 0.6-0.9 PFLOP/s, the MFMA + LDS-read rate of the
    // step decides there -- kept for the lead it gives layers with short steps).
    struct Cursor { int item, c, set; };
    Cursor cw = {(int)blockIdx.x, 0, -1}, cb = {(int)blockIdx.x, 0, -1};
    auto advance = [&](Cursor& q) { if (++q.c == nch) { q.c = 0; q.item += (int)gridDim.x; } };
    auto next_window = [&](bool loader, int wbuf) {
        if (cw.item >= p.nitems) return false;
        if (loader) {
            if (cw.set != cw.item) { set_window(cw.item); cw.set = cw.item; }   // (the other wave may have issued the item's first chunks)
            issue_window(cw.c, wbuf);
        }
        advance(cw);
        return true;
    };
    auto next_weights = [&](bool loader, int stage) {
        if (cb.item >= p.nitems) return false;
        if (loader) {
            if (cb.set != cb.item % p.cot) { set_weights(cb.item % p.cot); cb.set = cb.item % p.cot; }
            issue_weights(cb.c, stage);
        }
        advance(cb);
        return true;
    };
    int item = blockIdx.x, c = 0, step = 0, stage = 0;
    next_window(grp == 0, 0);                 // step 0: window + weights, and the weights of step 1
    next_weights(grp == 0, 0);
    next_weights(grp == 0, 1);
    bool counted = false;                     // this wave was the loader of the step before AND issued weights there
    for (;;) {
        // the loader of the step before has [window of this step][weights of the next step] in flight: the window must have landed;
        // the other wave's loads (a step older) carry this step's weights: everything
        if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LWT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // (item, c) has landed; every wave is done reading the buffers of the step before
        const bool last_chunk = c + 1 == nch;
        const bool loader = grp == ((step + 1) & 1);
        const bool more = next_window(loader, (step + 1) & 1);                      // window of step + 1 -> the other window buffer
        const bool wmore = next_weights(loader, stage == 0 ? 2 : stage - 1);       // weights of step + 2 -> stage (step + 2) % 3
        counted = loader && wmore;
        const unsigned char* win = sm + (step & 1) * WINB;
        const unsigned char* b0 = wts + stage * WTB;
        h16x8 a[RPW + 1][2][2];
#pragma unroll
        for (int dy = 0; dy < RPW + 1; ++dy)
#pragma unroll
            for (int sx = 0; sx < 2; ++sx)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) a[dy][sx][kk] = *reinterpret_cast<const h16x8*>(win + (aoff[dy][sx] ^ (kk << 5)));
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const Step st = KS == 3 ? STEPS[s] : STEP1[0];
            const int cls = st.cls, sy = st.sy, sx = st.sx, tap = st.tap;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const h16x8 b = *reinterpret_cast<const h16x8*>(b0 + tap * 4096 + (boff ^ (kk << 5)));
#pragma unroll
                for (int r = 0; r < RPW; ++r)   // D rows = output channels, columns = pixels
                    acc[r][cls] = MGN_MFMA_32x32x16(b, a[r + sy][sx][kk], acc[r][cls]);
            }
        }
        if (last_chunk) {
            // epilogue of `item` (the next item's first chunk is in flight): lane = low-res pixel l31 of row RPW wm + r; class (a, b) ->
            // output pixel (2 row + a, 2 (x0 + l31) + b); registers e -> channel (e & 3) + 8 (e >> 2) + 4 hi of the wave's 32
            int n, y0, x0, bn;
            decode(item, n, y0, x0, bn);
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int cls = 0; cls < 4; ++cls) {
                    const int oy = 2 * (y0 + wm * RPW + r) + (cls >> 1), ox = 2 * (x0 + l31) + (cls & 1);
                    const bool ok = oy < p.OH && ox < p.OW;
                    const size_t m = ((size_t)n * p.OH + (oy < p.OH ? oy : 0)) * p.OW + (ox < p.OW ? ox : 0);
                    uint16_t* opix = p.out + m * p.Cout + bn * 64 + wn * 32;
                    const bool add = RES == 1 || (RES == 2 && cls == 0);
                    const size_t mr = RES == 2 ? ((size_t)n * p.H + (y0 + wm * RPW + r < p.H ? y0 + wm * RPW + r : 0)) * p.W + (x0 + l31 < p.W ? x0 + l31 : 0) : m;
                    const uint16_t* rpix = add ? p.residual + mr * p.Cout + bn * 64 + wn * 32 : nullptr;
#pragma unroll
                    for (int qp = 0; qp < 2; ++qp) {
                        // v_permlane32_swap exchanges the 4-channel groups of lane l and l + 32: a lane then owns 8 consecutive channels
                        // (16-byte stores); the residual is read in that layout and brought to the accumulator layout by the same exchange
                        uint32_t rp[2][2] = {{0u, 0u}, {0u, 0u}};
                        if (add) {
                            const uint4 R = *reinterpret_cast<const uint4*>(rpix + 16 * qp + 8 * hi);
                            const auto u0 = __builtin_amdgcn_permlane32_swap(R.x, R.z, false, false);
                            const auto u1 = __builtin_amdgcn_permlane32_swap(R.y, R.w, false, false);
                            rp[0][0] = u0[0]; rp[1][0] = u0[1]; rp[0][1] = u1[0]; rp[1][1] = u1[1];
                        }
                        uint32_t pk[2][2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int q = 2 * qp + u;
                            float v0 = acc[r][cls][q * 4 + 0], v1 = acc[r][cls][q * 4 + 1], v2 = acc[r][cls][q * 4 + 2], v3 = acc[r][cls][q * 4 + 3];
                            if (add) {
                                v0 += mgn_lo2f(rp[u][0]); v1 += mgn_hi2f(rp[u][0]);
                                v2 += mgn_lo2f(rp[u][1]); v3 += mgn_hi2f(rp[u][1]);
                            }
                            pk[u][0] = mgn_pack2(v0, v1);
                            pk[u][1] = mgn_pack2(v2, v3);
                        }
                        const auto w0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                        const auto w1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                        if (ok) *reinterpret_cast<uint4*>(opix + 16 * qp + 8 * hi) = make_uint4(w0[0], w1[0], w0[1], w1[1]);
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[r][cls][e] = 0.f;
                }
        }
        if (!more) break;
        if (last_chunk) { item += (int)gridDim.x; c = 0; }
        else ++c;
        ++step;
        stage = stage == NWS - 1 ? 0 : stage + 1;
    }
}

__global__ __launch_bounds__(512, 1) void conv3x3_up2_win(Up2Params p) { up2_body<0, 3>(p); }
__global__ __launch_bounds__(512, 1) void conv3x3_up2_win_res(Up2Params p) { up2_body<1, 3>(p); }
__global__ __launch_bounds__(512, 1) void conv3x3_up2_win_reslo(Up2Params p) { up2_body<2, 3>(p); }
__global__ __launch_bounds__(512, 1) void conv1x1_up2_win(Up2Params p) { up2_body<0, 1>(p); }
__global__ __launch_bounds__(512, 1) void conv1x1_up2_win_res(Up2Params p) { up2_body<1, 1>(p); }

}  // namespace

extern "C" {

int MGN_SYM(mgn_conv3x3_up2_win)(const void* in, const void* w, void* out, int N, int H, int W, int Cin, int Cout, int OH, int OW,
                                 int ksize, const void* residual, int residual_lowres, void* stream) {
    if (!in || !w || !out || N < 1 || H < 1 || W < 1 || (ksize != 3 && ksize != 1)) return MGN_EINVAL;
    if (residual_lowres && (!residual || ksize != 3)) return MGN_EINVAL;
    if (Cin < 32 || Cin % 32 != 0 || Cout < 64 || Cout % 64 != 0) return MGN_ENOTSUP;
    if ((OH != 2 * H && OH != 2 * H - 1) || (OW != 2 * W && OW != 2 * W - 1)) return MGN_ENOTSUP;
    if ((size_t)N * H * W * Cin * 2 >= 0x7fffffffu || (size_t)Cout * ksize * ksize * Cin * 2 >= 0x7fffffffu) return MGN_ENOTSUP;   // 32-bit byte offsets
    Up2Params p;
    p.in = (const uint16_t*)in; p.w = (const uint16_t*)w; p.out = (uint16_t*)out; p.residual = (const uint16_t*)residual;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.OH = OH; p.OW = OW;
    p.py = (H + PH - 1) / PH; p.px = (W + PW - 1) / PW; p.cot = Cout / 64;
    const long nitems = (long)N * p.py * p.px * p.cot;
    if (nitems > 0x7fffffffL) return MGN_EINVAL;
    p.nitems = (int)nitems;
    static int cus = 0;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_up2_win), hipFuncAttributeMaxDynamicSharedMemorySize, Up2<3>::LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_up2_win_res), hipFuncAttributeMaxDynamicSharedMemorySize, Up2<3>::LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_up2_win_reslo), hipFuncAttributeMaxDynamicSharedMemorySize, Up2<3>::LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_up2_win), hipFuncAttributeMaxDynamicSharedMemorySize, Up2<1>::LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_up2_win_res), hipFuncAttributeMaxDynamicSharedMemorySize, Up2<1>::LDS);
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
            cus = 256;
        attr = true;
    }
    // persistent 8-wave blocks: one per CU for the 3x3 kernel (146 KB of LDS), two for the 1x1 kernel (50 KB; it is a stream of stores)
    const long want = ksize == 3 ? cus : 2L * cus;
    const dim3 grid((unsigned)(nitems < want ? nitems : want)), block(512);
    hipStream_t st = (hipStream_t)stream;
    if (ksize == 3 && residual && residual_lowres) hipLaunchKernelGGL(conv3x3_up2_win_reslo, grid, block, Up2<3>::LDS, st, p);
    else if (ksize == 3 && residual) hipLaunchKernelGGL(conv3x3_up2_win_res, grid, block, Up2<3>::LDS, st, p);
    else if (ksize == 3) hipLaunchKernelGGL(conv3x3_up2_win, grid, block, Up2<3>::LDS, st, p);
    else if (residual) hipLaunchKernelGGL(conv1x1_up2_win_res, grid, block, Up2<1>::LDS, st, p);
    else hipLaunchKernelGGL(conv1x1_up2_win, grid, block, Up2<1>::LDS, st, p);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

}  // extern "C"

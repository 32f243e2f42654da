// postproc.hip -- panoptic fusion of the inference path on the device (the step right AFTER the hot path).
//
// Replaces mgnet/postprocessing/panoptic_post_proc.py:9-147 (get_panoptic_prediction + _group_instances_and_fuse_logits).
// The reference builds an [N centres] x [P thing pixels] x 2 float tensor to take `norm(...).argmin(0)` (1.6 GB for 200
// centres and a megapixel of things), then a chain of masked_fill / masked_scatter / bincount passes with two host syncs.
// Here: six small launches, ~40 B/px of HBM traffic, no intermediate larger than 3 bytes per pixel, no host sync:
//
//   pp_nms      threshold + (k x k) max-pool NMS as a SPARSE test: only pixels above the threshold look at their window
//               (flat maxima tie and all survive, like `heat != max_pool(heat)`); flag byte per pixel + count per block
//   pp_scan     one block: exclusive scan of the block counts (=> torch.nonzero's row-major order), N, zeroes the votes
//   pp_compact  block-local scan of the flags -> centre list in row-major order
//   pp_group    every thing pixel: voted location = pixel + offset (fp32), nearest centre = FIRST minimum of the fp32
//               distance sqrt(dy*dy + dx*dx) evaluated without FMA contraction (what torch.norm(p=2).argmin returns);
//               centres are staged through LDS in chunks; instance id (u16) per pixel; class votes and stuff areas as
//               INTEGER atomics (order-independent)
//   pp_vote     per instance: majority class (first maximum)
//   pp_fuse     panoptic id = instance + (class + last_stuff) * divisor | stuff area filter -> void | * divisor
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgnet_hip.h"

namespace {

constexpr int CAP = MGN_PANOPTIC_MAX_CENTERS;   // 65534: the reference's own limit (65535 is its sentinel, :112-126)
constexpr int PXT = 8;                          // pixels per thread
constexpr int SEG = 256 * PXT;                  // pixels per block (one row segment)
constexpr int HSTRIDE = 32;                     // ints between two stuff-area counters

struct PPParams {
    mgn_panoptic_cfg c;
    const long long* sem;     // [H,W]
    const float* center;      // [H,W]
    const float* offsets;     // [2,H,W]
    long long* pan;           // [H,W]
    int* info;                // [2]: centres found, overflow flag
    uint8_t* flags;           // [H*W]
    int* blk_count;           // [nblk]
    int* blk_off;             // [nblk]
    float2* centres;          // [CAP] (y, x)
    int* votes;               // [CAP][num_thing + 1]
    uint8_t* cls_of;          // [CAP]
    uint16_t* cluster;        // [H*W]
    int* stuff_hist;          // [256][HSTRIDE]: one counter per 128-byte line (same-line global atomics serialise)
    int* meta;                // [0] = N (clamped)
    int nbx;                  // blocks per row
};

// 8 consecutive elements of a row segment: one or two 16-byte accesses when the row length keeps them aligned (W % 8 == 0),
// element-wise with a bounds check otherwise
template <typename T> struct alignas(sizeof(T) * 8 >= 16 ? 16 : 8) Pack8 { T v[8]; };

template <typename T>
__device__ __forceinline__ void load8(const T* ptr, long base, int x0, int W, T (&v)[PXT], T fill) {
    if ((W & 7) == 0) {
        const Pack8<T> pk = *reinterpret_cast<const Pack8<T>*>(ptr + base);
#pragma unroll
        for (int k = 0; k < PXT; ++k) v[k] = pk.v[k];
    } else {
#pragma unroll
        for (int k = 0; k < PXT; ++k) v[k] = (x0 + k < W) ? ptr[base + k] : fill;
    }
}

template <typename T>
__device__ __forceinline__ void store8(T* ptr, long base, int x0, int W, const T (&v)[PXT]) {
    if ((W & 7) == 0) {
        Pack8<T> pk;
#pragma unroll
        for (int k = 0; k < PXT; ++k) pk.v[k] = v[k];
        *reinterpret_cast<Pack8<T>*>(ptr + base) = pk;
    } else {
#pragma unroll
        for (int k = 0; k < PXT; ++k)
            if (x0 + k < W) ptr[base + k] = v[k];
    }
}

__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {   // 256 threads
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    __syncthreads();   // wsum reuse
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wid) base += wsum[w];
    }
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return base + x - v;
}

__global__ __launch_bounds__(256) void pp_nms(PPParams p) {
    // pixel k of thread t is x = seg0 + k * 256 + t: candidates (pixels above the threshold) are spatially clustered, so
    // neighbouring candidates must sit in neighbouring LANES, not in one thread's private run of pixels
    const int H = p.c.H, W = p.c.W, y = blockIdx.x / p.nbx, seg0 = (blockIdx.x % p.nbx) * SEG;
    const int r = (p.c.nms_kernel - 1) / 2;
    const float thr = p.c.threshold;
    const float* own_row = p.center + (long)y * W;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < PXT; ++k) {
        const int x = seg0 + k * 256 + threadIdx.x;
        if (x >= W) continue;
        const float v = own_row[x];
        bool is_max = (v > thr) && (v > 0.f);   // F.threshold(x, thr, -1) then `> 0` (:53,59)
        if (is_max) {
            const int xa = max(x - r, 0), xb = min(x + r, W - 1);
            for (int yy = max(y - r, 0); yy <= min(y + r, H - 1) && is_max; ++yy) {   // max_pool2d pads with -inf
                const float* row = p.center + (long)yy * W;
                float m = v;
                for (int xx = xa; xx <= xb; ++xx) m = fmaxf(m, row[xx]);   // independent loads: one latency per window row
                is_max = !(m > v);   // (a neighbour above v is above the threshold too)
            }
        }
        p.flags[(long)y * W + x] = is_max;
        cnt += is_max;
    }
    int total;
    block_exclusive_scan(cnt, &total);
    if (threadIdx.x == 0) p.blk_count[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void pp_scan(PPParams p, int nblk) {
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const int v = i < nblk ? p.blk_count[i] : 0;
        int total;
        const int ex = block_exclusive_scan(v, &total);
        if (i < nblk) p.blk_off[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    const int found = carry, n = min(found, CAP);
    if (threadIdx.x == 0) { p.info[0] = found; p.info[1] = found > CAP; p.meta[0] = n; }
    const int m0 = p.c.num_thing_classes + 1;
    for (long i = threadIdx.x; i < (long)n * m0; i += 256) p.votes[i] = 0;
    for (int i = threadIdx.x; i < 256 * HSTRIDE; i += 256) p.stuff_hist[i] = 0;
}

__global__ __launch_bounds__(256) void pp_compact(PPParams p) {
    const int W = p.c.W, y = blockIdx.x / p.nbx, x0 = (blockIdx.x % p.nbx) * SEG + threadIdx.x * PXT;
    if (p.blk_count[blockIdx.x] == 0) return;
    const long base = (long)y * W + x0;
    uint8_t f[PXT];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < PXT; ++k) f[k] = 0;
    if (x0 < W) load8(p.flags, base, x0, W, f, (uint8_t)0);
#pragma unroll
    for (int k = 0; k < PXT; ++k) cnt += f[k];
    int total;
    int rank = p.blk_off[blockIdx.x] + block_exclusive_scan(cnt, &total);
#pragma unroll
    for (int k = 0; k < PXT; ++k)
        if (f[k]) { if (rank < CAP) p.centres[rank] = make_float2((float)y, (float)(x0 + k)); ++rank; }
}

constexpr int CHUNK = 2048;
constexpr int VSLOTS = 512;

__global__ __launch_bounds__(256) void pp_group(PPParams p) {
    __shared__ float2 cs[CHUNK];
    __shared__ int hist[256];
    __shared__ int vkey[VSLOTS], vcnt[VSLOTS];   // block-private vote cache: (instance, class) -> count
    const int W = p.c.W, y = blockIdx.x / p.nbx, x0 = (blockIdx.x % p.nbx) * SEG + threadIdx.x * PXT;
    const long hw = (long)p.c.H * W, base = (long)y * W + x0;
    const int n = p.meta[0], last_stuff = p.c.last_stuff_id, m0 = p.c.num_thing_classes + 1;
    hist[threadIdx.x] = 0;
    vkey[threadIdx.x] = vkey[threadIdx.x + 256] = -1;
    vcnt[threadIdx.x] = vcnt[threadIdx.x + 256] = 0;
    int sem[PXT];
    float ly[PXT], lx[PXT], bd[PXT], bq[PXT];
    int best[PXT];
    bool thing[PXT];
    bool any = false;
    const bool in_row = x0 < W;
    long long sem64[PXT];
    float oy[PXT], ox[PXT];
    if (in_row) {
        load8(p.sem, base, x0, W, sem64, 0LL);
        load8(p.offsets, base, x0, W, oy, 0.f);
        load8(p.offsets + hw, base, x0, W, ox, 0.f);
    }
#pragma unroll
    for (int k = 0; k < PXT; ++k) {
        thing[k] = false; best[k] = -1; bd[k] = bq[k] = __int_as_float(0x7f800000); sem[k] = 0; ly[k] = lx[k] = 0.f;
        if (!in_row || x0 + k >= W) continue;
        sem[k] = (int)sem64[k];
        thing[k] = sem[k] > last_stuff;
        if (thing[k]) {
            ly[k] = __fadd_rn(oy[k], (float)y);             // offsets += xy (:108), fp32
            lx[k] = __fadd_rn(ox[k], (float)(x0 + k));
            any = true;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PXT; ++k)
        if (x0 + k < W && sem[k] >= 0 && sem[k] <= last_stuff && sem[k] < 256) atomicAdd(&hist[sem[k]], 1);
    for (int c0 = 0; c0 < n; c0 += CHUNK) {
        const int m = min(CHUNK, n - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += 256) cs[i] = p.centres[c0 + i];
        __syncthreads();
        if (!any) continue;
        for (int i = 0; i < m; ++i) {
            const float2 c = cs[i];
#pragma unroll
            for (int k = 0; k < PXT; ++k) {
                const float dy = __fsub_rn(c.x, ly[k]), dx = __fsub_rn(c.y, lx[k]);
                const float q = __fadd_rn(__fmul_rn(dy, dy), __fmul_rn(dx, dx));   // torch.norm(p=2) = sqrt(q), no FMA
                // argmin of sqrt(q), first minimum.  sqrt is monotonic, so q >= (best q) can never be strictly closer;
                // only an improving q pays for the correctly rounded sqrt (two q may round to the same distance: a tie,
                // which the earlier centre keeps)
                if (q < bq[k]) {
                    const float d = __fsqrt_rn(q);
                    if (d < bd[k]) { bd[k] = d; bq[k] = q; best[k] = c0 + i; }
                }
            }
        }
    }
    // votes: runs of equal (instance, class) inside the thread's 8 pixels are merged, then counted in the block's LDS cache
    // (direct-mapped; a colliding key goes straight to the global counter) -- millions of pixels vote for a few hundred
    // counters, which as plain global atomics serialise in L2
    int run_key = -1, run_n = 0;
    auto cast = [&](int key, int cnt) {
        const int slot = (int)(((unsigned)key * 2654435761u) >> 23) & (VSLOTS - 1);
        const int old = atomicCAS(&vkey[slot], -1, key);
        if (old == -1 || old == key) atomicAdd(&vcnt[slot], cnt);
        else atomicAdd(&p.votes[key], cnt);
    };
    uint16_t clv[PXT];
#pragma unroll
    for (int k = 0; k < PXT; ++k) {
        clv[k] = 0;
        if (x0 + k >= W) continue;
        int cl = 0;
        if (thing[k] && n > 0) {
            // (a NaN location never compares below: index 0, like argmin over an all-NaN column)
            cl = (best[k] < 0 ? 0 : best[k]) + 1;
            const int cls = sem[k] - last_stuff;
            if (cls < m0) {
                const int key = (cl - 1) * m0 + cls;
                if (key == run_key) ++run_n;
                else { if (run_n) cast(run_key, run_n); run_key = key; run_n = 1; }
            }
        }
        clv[k] = (uint16_t)cl;
    }
    if (in_row) store8(p.cluster, base, x0, W, clv);
    if (run_n) cast(run_key, run_n);
    __syncthreads();
    for (int sidx = threadIdx.x; sidx < VSLOTS; sidx += 256)
        if (vcnt[sidx]) atomicAdd(&p.votes[vkey[sidx]], vcnt[sidx]);
    if (threadIdx.x <= last_stuff && threadIdx.x < 256 && hist[threadIdx.x]) atomicAdd(&p.stuff_hist[threadIdx.x * HSTRIDE], hist[threadIdx.x]);
}

__global__ __launch_bounds__(256) void pp_vote(PPParams p) {
    const int i = blockIdx.x * 256 + threadIdx.x, n = p.meta[0], m0 = p.c.num_thing_classes + 1;
    if (i >= n) return;
    int bestc = 0, bestv = p.votes[(long)i * m0];
    for (int c = 1; c < m0; ++c) {
        const int v = p.votes[(long)i * m0 + c];
        if (v > bestv) { bestv = v; bestc = c; }   // bins.max(1)[1]: first maximum (:137)
    }
    p.cls_of[i] = (uint8_t)bestc;
}

__global__ __launch_bounds__(256) void pp_fuse(PPParams p) {
    const int W = p.c.W, y = blockIdx.x / p.nbx, x0 = (blockIdx.x % p.nbx) * SEG + threadIdx.x * PXT;
    const long base = (long)y * W + x0;
    const int last_stuff = p.c.last_stuff_id, ld = p.c.label_divisor, vd = p.c.void_label;
    if (x0 >= W) return;
    long long sem64[PXT], out[PXT];
    uint16_t clv[PXT];
    load8(p.sem, base, x0, W, sem64, 0LL);
    load8(p.cluster, base, x0, W, clv, (uint16_t)0);
#pragma unroll
    for (int k = 0; k < PXT; ++k) {
        const long long s = sem64[k];
        const int cl = clv[k];
        long long v = s;
        if (cl > 0) v = (long long)cl + (long long)((int)p.cls_of[cl - 1] + last_stuff) * ld;        // :138-144
        else if (s >= 0 && s <= last_stuff && s < 256 && p.stuff_hist[s * HSTRIDE] < p.c.stuff_area) v = vd;  // :64-66
        if (v < ld && v != vd) v *= ld;                                                             // :68-69
        out[k] = v;
    }
    store8(p.pan, base, x0, W, out);
}

bool cfg_ok(const mgn_panoptic_cfg* c) {
    return c && c->H >= 1 && c->W >= 1 && (long)c->H * c->W < (1L << 31) && c->num_thing_classes >= 0 &&
           c->num_thing_classes < 255 && c->last_stuff_id >= 0 && c->last_stuff_id < 255 && c->label_divisor >= 1 &&
           c->nms_kernel >= 1 && (c->nms_kernel & 1) && c->nms_kernel <= 63;
}
size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

struct Layout { size_t flags, blk_count, blk_off, centres, votes, cls_of, cluster, stuff_hist, meta, total; int nbx, nblk; };
Layout layout(const mgn_panoptic_cfg* c) {
    Layout L;
    const size_t hw = (size_t)c->H * c->W;
    L.nbx = (c->W + SEG - 1) / SEG;
    L.nblk = L.nbx * c->H;
    size_t o = 0;
    L.flags = o; o += align16(hw);
    L.blk_count = o; o += align16((size_t)L.nblk * 4);
    L.blk_off = o; o += align16((size_t)L.nblk * 4);
    L.centres = o; o += align16((size_t)CAP * 8);
    L.votes = o; o += align16((size_t)CAP * (c->num_thing_classes + 1) * 4);
    L.cls_of = o; o += align16(CAP);
    L.cluster = o; o += align16(hw * 2);
    L.stuff_hist = o; o += 256 * HSTRIDE * 4;
    L.meta = o; o += 16;
    L.total = o;
    return L;
}


// =====================================================================================================================
// Depth post-processing: metric rescaling with the DGC module (depth_post_proc.py:11-185)
//
//   dp_heights  per pixel: back-project the 3x3 neighbourhood (camera.py:107-141, frame "c"), four cross-product normals ->
//               mean -> unit normal (replicate padding = the normal of the nearest interior pixel), camera height
//               |P . n|, ground test (panoptic == road id, or |cos(n, vertical)| > cos 5 deg and y > 0); writes the height as
//               an ordered 32-bit key (0xffffffff for non-ground pixels) and counts the ground pixels
//   dp_hist / dp_pick  x4: radix select of the LOWER median (torch.median) among the keys -- no sort, 4 B/px per pass
//   dp_apply    scale = real height / median; depth * scale, points * scale, filtered classes -> 0 / NaN
// Nothing but the depth map (and the panoptic ids) is read from HBM: the 3-D points are recomputed where needed.
struct DPParams {
    mgn_depth_post_cfg c;
    const float* depth;       // [H,W]
    const long long* pan;     // [H,W] or null
    float* out_depth;         // [H,W]
    float* xyz;               // [3,H,W] or null
    float* scale;             // [1]
    unsigned* keys;           // [H*W]
    unsigned* hist;           // [256]
    unsigned* state;          // [0] ground count, [1] prefix, [2] k
    float ifx, ify, cxi, cyi; // rows of Kinv: 1/fx, 1/fy, -cx/fx, -cy/fy (fp32, computed once on the host)
};

__device__ __forceinline__ void dp_point(const DPParams& p, int y, int x, float (&P)[3]) {
    const float d = p.depth[(long)y * p.c.W + x];
    // Kinv = [[1/fx, 0, -cx/fx], [0, 1/fy, -cy/fy], [0, 0, 1]] (camera.py:74-81) applied to (u, v, 1), times depth
    P[0] = __fmul_rn(__fadd_rn(__fmul_rn(p.ifx, (float)x), p.cxi), d);
    P[1] = __fmul_rn(__fadd_rn(__fmul_rn(p.ify, (float)y), p.cyi), d);
    P[2] = d;
}

__device__ __forceinline__ void dp_unit_cross(const float (&a)[3], const float (&b)[3], float (&acc)[3]) {
    const float cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2], cz = a[0] * b[1] - a[1] * b[0];
    const float inv = 1.0f / fmaxf(sqrtf(cx * cx + cy * cy + cz * cz), 1e-12f);   // F.normalize (eps 1e-12)
    acc[0] += cx * inv; acc[1] += cy * inv; acc[2] += cz * inv;
}

constexpr int DPX = 4;   // pixels per thread (block-strided: coalesced)

__global__ __launch_bounds__(256) void dp_heights(DPParams p) {
    __shared__ unsigned wcnt[4];
    const int H = p.c.H, W = p.c.W;
    const long hw = (long)H * W;
    unsigned mine = 0;
#pragma unroll 1
    for (int it = 0; it < DPX; ++it) {
    const long i = ((long)blockIdx.x * DPX + it) * 256 + threadIdx.x;
    bool ground = false;
    if (i < hw) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        const int yc = min(max(y, 1), H - 2), xc = min(max(x, 1), W - 2);   // replicate padding of the interior normals
        float c[3], q[3], v[8][3];
        dp_point(p, yc, xc, c);
        const int dy8[8] = {0, -1, 0, 1, -1, 1, -1, 1}, dx8[8] = {-1, 0, 1, 0, -1, -1, 1, 1};   // x0 y0 x1 y1 x0y0 x0y1 x1y0 x1y1
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            dp_point(p, yc + dy8[k], xc + dx8[k], q);
            v[k][0] = q[0] - c[0]; v[k][1] = q[1] - c[1]; v[k][2] = q[2] - c[2];
        }
        float n[3] = {0.f, 0.f, 0.f};
        dp_unit_cross(v[0], v[1], n);   // (:138-141)
        dp_unit_cross(v[2], v[3], n);
        dp_unit_cross(v[4], v[5], n);
        dp_unit_cross(v[6], v[7], n);
        n[0] *= 0.25f; n[1] *= 0.25f; n[2] *= 0.25f;
        const float nl = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), inv = 1.0f / fmaxf(nl, 1e-12f);
        n[0] *= inv; n[1] *= inv; n[2] *= inv;
        float own[3];
        dp_point(p, y, x, own);
        const float h = fabsf(own[0] * n[0] + own[1] * n[1] + own[2] * n[2]);   // (:98)
        if (p.pan) ground = p.pan[i] == p.c.road_class_id;
        else {
            const float nn = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            const float cs = n[1] / fmaxf(nn, 1e-6f), t = 0.99619469809174555f;   // cos(5 deg) (:168-176)
            ground = (cs > t || cs < -t) && !(own[1] <= 0.f);
        }
        p.keys[i] = ground ? __float_as_uint(h) : 0xffffffffu;   // h >= 0: the bit pattern orders like the value
    }
    mine += (unsigned)__popcll(__ballot(ground));
    }
    // one counter update per block (a per-wave atomic on a single address serialises 32k updates per frame)
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) { const unsigned t = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]; if (t) atomicAdd(&p.state[0], t); }
}

__global__ void dp_begin(DPParams p) {   // after dp_heights: k of the lower median, clear the histogram
    if (threadIdx.x < 256) p.hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) { const unsigned n = p.state[0]; p.state[1] = 0; p.state[2] = n ? (n - 1) / 2 : 0; }
}

__global__ __launch_bounds__(256) void dp_hist(DPParams p, int pass) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const long hw = (long)p.c.H * p.c.W;
    const unsigned prefix = p.state[1];
    const int shift = 8 * pass;
    const int lane = threadIdx.x & 63;
    for (long i0 = (long)blockIdx.x * 256; i0 < hw; i0 += (long)gridDim.x * 256) {
        const long i = i0 + threadIdx.x;
        const unsigned k = i < hw ? p.keys[i] : 0u;
        const bool act = i < hw && (pass == 3 || (k >> (shift + 8)) == prefix);
        const unsigned d = (k >> shift) & 255u;
        // most keys of a wave share the digit (the non-ground sentinel, one exponent byte): one LDS update for that group
        const unsigned long long todo = __ballot(act);
        if (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const unsigned dl = __shfl(d, leader);
            const unsigned long long same = __ballot(act && d == dl);
            if (lane == leader) atomicAdd(&h[dl], (unsigned)__popcll(same));
            else if (act && d != dl) atomicAdd(&h[d], 1u);
        }
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&p.hist[threadIdx.x], h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void dp_pick(DPParams p) {   // 256 threads: the digit whose cumulative count passes k
    const unsigned k = p.state[2], prefix = p.state[1];
    const int c = (int)p.hist[threadIdx.x];
    int total;
    const int ex = block_exclusive_scan(c, &total);
    __syncthreads();
    if ((int)k >= ex && (int)k < ex + c) { p.state[1] = (prefix << 8) | threadIdx.x; p.state[2] = k - (unsigned)ex; }
    else if (threadIdx.x == 255 && (int)k >= total) { p.state[1] = (prefix << 8) | 255u; p.state[2] = 0; }   // (empty set)
    p.hist[threadIdx.x] = 0;
}

__global__ __launch_bounds__(256) void dp_apply(DPParams p) {
    const long hw = (long)p.c.H * p.c.W, i = (long)blockIdx.x * 256 + threadIdx.x;
    float scale = 1.0f;
    if (p.c.use_dgc_scaling) {
        const float med = p.state[0] ? __uint_as_float(p.state[1]) : __int_as_float(0x7fc00000);   // no ground pixel: NaN
        scale = __fmul_rn(1.0f / med, p.c.real_camera_height);   // reciprocal().mul_() (:102)
        if (i == 0) p.scale[0] = scale;
    } else if (i == 0) p.scale[0] = 1.0f;
    if (i >= hw) return;
    const int y = (int)(i / p.c.W), x = (int)(i - (long)y * p.c.W);
    bool drop = false;
    if (p.pan) {
        const long long id = p.pan[i];
        for (int f = 0; f < p.c.n_filter; ++f) drop |= id == p.c.filter_ids[f];   // (:62-68)
    }
    float P[3];
    dp_point(p, y, x, P);
    p.out_depth[i] = drop ? 0.f : P[2] * scale;
    if (p.xyz) {
        const float nan = __int_as_float(0x7fc00000);
        p.xyz[i] = drop ? nan : P[0] * scale;
        p.xyz[hw + i] = drop ? nan : P[1] * scale;
        p.xyz[2 * hw + i] = drop ? nan : P[2] * scale;
    }
}

bool dcfg_ok(const mgn_depth_post_cfg* c) {
    return c && c->H >= 3 && c->W >= 3 && (long)c->H * c->W < (1L << 31) && c->n_filter >= 0 && c->n_filter <= MGN_DEPTH_MAX_FILTER_IDS;
}

// =====================================================================================================================
// Depth metrics (mgnet/evaluation/depth_evaluation.py:48-112, DepthEvaluator.process for one frame)
//   dm_keys    mask = min < label < max (and the Eigen crop); ordered keys of label / prediction for the median scaling
//   dp_hist / dm_pick  radix selects of the two middle elements of each (np.median averages them)
//   dm_reduce  per masked pixel: scaled + clamped prediction, the seven error terms, fp64 block partials
//   dm_final   fixed-order sum of the partials -> means, sqrt
struct DMParams {
    const float* pred; const float* label;
    int H, W, y0, y1, x0, x1;      // crop window (whole frame when the Eigen crop is off)
    float min_depth, max_depth;
    int use_gt_scale;
    unsigned* keys_l; unsigned* keys_p; unsigned* hist; unsigned* state;   // state: [0] n, [1] prefix, [2] k
    float* med;                    // [4]: label lower/upper middle, prediction lower/upper middle
    double* partials;              // [nblk][8]
    double* out;                   // [9]
    int nblk;
};

__device__ __forceinline__ unsigned ord_key(float f) {   // order-preserving for all finite floats
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_ord(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__device__ __forceinline__ bool dm_mask(const DMParams& p, long i, float l) {
    const int y = (int)(i / p.W), x = (int)(i - (long)y * p.W);
    return l > p.min_depth && l < p.max_depth && y >= p.y0 && y < p.y1 && x >= p.x0 && x < p.x1;
}

__global__ __launch_bounds__(256) void dm_keys(DMParams p) {
    __shared__ unsigned wcnt[4];
    const long hw = (long)p.H * p.W;
    unsigned mine = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
        const float l = p.label[i];
        const bool m = dm_mask(p, i, l);
        p.keys_l[i] = m ? ord_key(l) : 0xffffffffu;
        p.keys_p[i] = m ? ord_key(p.pred[i]) : 0xffffffffu;
        mine += (unsigned)__popcll(__ballot(m));
    }
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) { const unsigned t = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]; if (t) atomicAdd(&p.state[0], t); }
}

__global__ void dm_begin(DMParams p, int upper) {   // k of the lower / upper middle element
    if (threadIdx.x < 256) p.hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) { const unsigned n = p.state[0]; p.state[1] = 0; p.state[2] = n ? (upper ? n / 2 : (n - 1) / 2) : 0; }
}

__global__ void dm_store(DMParams p, int slot) {
    if (threadIdx.x == 0) p.med[slot] = p.state[0] ? key_ord(p.state[1]) : __int_as_float(0x7fc00000);
}

__global__ __launch_bounds__(256) void dm_reduce(DMParams p) {
    __shared__ double red[8][4];
    const long hw = (long)p.H * p.W;
    float ratio = 1.0f;
    if (p.use_gt_scale) {   // np.median = mean of the two middle elements, in float32 (:88-90)
        const float ml = (p.med[0] + p.med[1]) * 0.5f, mp = (p.med[2] + p.med[3]) * 0.5f;
        ratio = ml / mp;
    }
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
        const float l = p.label[i];
        if (!dm_mask(p, i, l)) continue;
        float q = p.pred[i];
        if (p.use_gt_scale) q = __fmul_rn(q, ratio);
        q = q < p.min_depth ? p.min_depth : q;   // (:92-93)
        q = q > p.max_depth ? p.max_depth : q;
        const float th = fmaxf(l / q, q / l);
        const float d = l - q, d2 = d * d, dl = logf(l) - logf(q);
        a[0] += (double)(fabsf(d) / l);            // abs rel
        a[1] += (double)(d2 / l);                  // sq rel
        a[2] += (double)d2;                        // rmse^2
        a[3] += (double)(dl * dl);                 // rmse log^2
        a[4] += th < 1.25f ? 1.0 : 0.0;
        a[5] += th < (float)(1.25 * 1.25) ? 1.0 : 0.0;
        a[6] += th < (float)(1.25 * 1.25 * 1.25) ? 1.0 : 0.0;
        a[7] += 1.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double v = a[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8) p.partials[(long)blockIdx.x * 8 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ void dm_final(DMParams p) {
    if (threadIdx.x >= 8) return;
    double s = 0;
    for (int b = 0; b < p.nblk; ++b) s += p.partials[(long)b * 8 + threadIdx.x];
    __shared__ double tot[8];
    tot[threadIdx.x] = s;
    __syncthreads();
    const double n = tot[7];
    if (threadIdx.x < 7) {
        double m = s / n;
        if (threadIdx.x == 2 || threadIdx.x == 3) m = sqrt(m);
        p.out[threadIdx.x] = m;
    } else {
        p.out[8] = n;
        float ratio = 1.0f;
        if (p.use_gt_scale) ratio = ((p.med[0] + p.med[1]) * 0.5f) / ((p.med[2] + p.med[3]) * 0.5f);
        p.out[7] = (double)ratio;
    }
}

}  // namespace

extern "C" int mgn_panoptic_post_workspace_bytes(const mgn_panoptic_cfg* cfg, size_t* bytes) {
    if (!cfg_ok(cfg) || !bytes) return MGN_EINVAL;
    *bytes = layout(cfg).total;
    return MGN_OK;
}

extern "C" int mgn_panoptic_post(const mgn_panoptic_cfg* cfg, const int64_t* sem_seg, const float* center_heatmap,
                                 const float* offsets, int64_t* panoptic, int32_t* info, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (!cfg_ok(cfg) || !sem_seg || !center_heatmap || !offsets || !panoptic || !info || !workspace) return MGN_EINVAL;
    const Layout L = layout(cfg);
    if (workspace_bytes < L.total) return MGN_ENOSPC;
    char* w = (char*)workspace;
    PPParams p;
    p.c = *cfg; p.sem = (const long long*)sem_seg; p.center = center_heatmap; p.offsets = offsets; p.pan = (long long*)panoptic;
    p.info = info; p.flags = (uint8_t*)(w + L.flags); p.blk_count = (int*)(w + L.blk_count); p.blk_off = (int*)(w + L.blk_off);
    p.centres = (float2*)(w + L.centres); p.votes = (int*)(w + L.votes); p.cls_of = (uint8_t*)(w + L.cls_of);
    p.cluster = (uint16_t*)(w + L.cluster); p.stuff_hist = (int*)(w + L.stuff_hist); p.meta = (int*)(w + L.meta); p.nbx = L.nbx;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(L.nblk), blk(256);
    hipLaunchKernelGGL(pp_nms, grid, blk, 0, st, p);
    hipLaunchKernelGGL(pp_scan, dim3(1), blk, 0, st, p, L.nblk);
    hipLaunchKernelGGL(pp_compact, grid, blk, 0, st, p);
    hipLaunchKernelGGL(pp_group, grid, blk, 0, st, p);
    hipLaunchKernelGGL(pp_vote, dim3((CAP + 255) / 256), blk, 0, st, p);
    hipLaunchKernelGGL(pp_fuse, grid, blk, 0, st, p);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

extern "C" int mgn_depth_post_workspace_bytes(const mgn_depth_post_cfg* cfg, size_t* bytes) {
    if (!dcfg_ok(cfg) || !bytes) return MGN_EINVAL;
    *bytes = align16((size_t)cfg->H * cfg->W * 4) + 1024 + 16;
    return MGN_OK;
}

extern "C" int mgn_depth_post(const mgn_depth_post_cfg* cfg, const float* depth, const int64_t* panoptic, float* depth_out,
                              float* xyz, float* scale, void* workspace, size_t workspace_bytes, void* stream) {
    size_t need = 0;
    if (mgn_depth_post_workspace_bytes(cfg, &need) != MGN_OK || !depth || !depth_out || !scale || !workspace) return MGN_EINVAL;
    if (cfg->has_panoptic != (panoptic != nullptr)) return MGN_EINVAL;
    if (cfg->use_dgc_scaling && (!xyz || !(cfg->fx != 0.f) || !(cfg->fy != 0.f))) return MGN_EINVAL;
    if (workspace_bytes < need) return MGN_ENOSPC;
    DPParams p;
    p.c = *cfg; p.depth = depth; p.pan = (const long long*)panoptic; p.out_depth = depth_out;
    p.xyz = cfg->use_dgc_scaling ? xyz : nullptr; p.scale = scale;
    p.ifx = 1.0f / cfg->fx; p.ify = 1.0f / cfg->fy; p.cxi = -cfg->cx / cfg->fx; p.cyi = -cfg->cy / cfg->fy;
    const size_t hw = (size_t)cfg->H * cfg->W;
    p.keys = (unsigned*)workspace;
    p.hist = (unsigned*)((char*)workspace + align16(hw * 4));
    p.state = p.hist + 256;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((hw + 255) / 256)), blk(256);
    if (cfg->use_dgc_scaling) {
        if (hipMemsetAsync(p.state, 0, 16, st) != hipSuccess) return MGN_ELAUNCH;
        hipLaunchKernelGGL(dp_heights, dim3((unsigned)((hw + 256 * DPX - 1) / (256 * DPX))), blk, 0, st, p);
        hipLaunchKernelGGL(dp_begin, dim3(1), blk, 0, st, p);
        const unsigned hb = (unsigned)((hw + 256 * 16 - 1) / (256 * 16));
        for (int pass = 3; pass >= 0; --pass) {
            hipLaunchKernelGGL(dp_hist, dim3(hb < 2048 ? hb : 2048), blk, 0, st, p, pass);
            hipLaunchKernelGGL(dp_pick, dim3(1), blk, 0, st, p);
        }
    }
    hipLaunchKernelGGL(dp_apply, grid, blk, 0, st, p);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

static const int DM_BLOCKS = 1024;

extern "C" int mgn_depth_metrics_workspace_bytes(int H, int W, size_t* bytes) {
    if (H < 1 || W < 1 || (long)H * W >= (1L << 31) || !bytes) return MGN_EINVAL;
    *bytes = 2 * align16((size_t)H * W * 4) + 1024 + 16 + 16 + (size_t)DM_BLOCKS * 8 * sizeof(double);
    return MGN_OK;
}

extern "C" int mgn_depth_metrics(const float* prediction, const float* label, int H, int W, float min_depth, float max_depth,
                                 int use_gt_scale, int crop_y0, int crop_y1, int crop_x0, int crop_x1, double* out9,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    size_t need = 0;
    if (mgn_depth_metrics_workspace_bytes(H, W, &need) != MGN_OK || !prediction || !label || !out9 || !workspace) return MGN_EINVAL;
    if (workspace_bytes < need) return MGN_ENOSPC;
    const size_t hw = (size_t)H * W;
    char* w = (char*)workspace;
    DMParams p;
    p.pred = prediction; p.label = label; p.H = H; p.W = W; p.y0 = crop_y0; p.y1 = crop_y1; p.x0 = crop_x0; p.x1 = crop_x1;
    p.min_depth = min_depth; p.max_depth = max_depth; p.use_gt_scale = use_gt_scale;
    p.keys_l = (unsigned*)w; w += align16(hw * 4);
    p.keys_p = (unsigned*)w; w += align16(hw * 4);
    p.hist = (unsigned*)w; w += 1024;
    p.state = (unsigned*)w; w += 16;
    p.med = (float*)w; w += 16;
    p.partials = (double*)w;
    p.out = out9;
    const unsigned nb = (unsigned)((hw + 256 * 16 - 1) / (256 * 16));
    p.nblk = (int)(nb < (unsigned)DM_BLOCKS ? nb : (unsigned)DM_BLOCKS);
    hipStream_t st = (hipStream_t)stream;
    const dim3 blk(256), grid(p.nblk);
    if (use_gt_scale) {
        if (hipMemsetAsync(p.state, 0, 16, st) != hipSuccess) return MGN_ELAUNCH;
        hipLaunchKernelGGL(dm_keys, grid, blk, 0, st, p);
        DPParams d;   // the select kernels of the depth post-processing, pointed at each key array in turn
        d.c.H = H; d.c.W = W; d.hist = p.hist; d.state = p.state;
        for (int slot = 0; slot < 4; ++slot) {
            d.keys = slot < 2 ? p.keys_l : p.keys_p;
            hipLaunchKernelGGL(dm_begin, dim3(1), blk, 0, st, p, slot & 1);
            for (int pass = 3; pass >= 0; --pass) {
                hipLaunchKernelGGL(dp_hist, grid, blk, 0, st, d, pass);
                hipLaunchKernelGGL(dp_pick, dim3(1), blk, 0, st, d);
            }
            hipLaunchKernelGGL(dm_store, dim3(1), dim3(64), 0, st, p, slot);
        }
    }
    hipLaunchKernelGGL(dm_reduce, grid, blk, 0, st, p);
    hipLaunchKernelGGL(dm_final, dim3(1), dim3(64), 0, st, p);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

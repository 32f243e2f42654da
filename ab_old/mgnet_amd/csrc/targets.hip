// targets.hip -- panoptic training targets on the device (the step right BEFORE the hot path).
//
// Replaces mgnet/data/target_generator.py:54-158 (PanopticDeepLabTargetGenerator.__call__), the `rgb2id` decode in front of
// it (dataset_mapper.py:178) and the class part of the reprojection mask (dataset_mapper.py:214-216).  The reference walks
// the segment list on the host and re-scans the whole label image 3-6 times per segment (`panoptic == seg["id"]`,
// `np.where`): O(segments x pixels) numpy work per frame plus 32 bytes per pixel of host->device copies for the finished
// maps.  Here only the label image crosses PCIe (3-4 B/px) and three small launches build every map:
//
//   pt_stats    one pass over the labels: pixel -> segment by binary search in the sorted id table (LDS), per-segment
//               area / sum(y) / sum(x) accumulated as INTEGERS (LDS atomics per block, one 64-bit global atomic per touched
//               segment and block) => bit-identical for any launch shape
//   pt_finalize per segment: centre = sums / area in fp64 (np.mean of integer indices is exact below 2^53), rounded
//               half-to-even (np.round) to the Gaussian's anchor, flags
//   pt_emit     second pass, 16 x 128 pixel tiles: each block first lists the Gaussians whose (6 sigma + 3)^2 window touches
//               its tile, then every thread produces 8 consecutive pixels of all six maps (+ the reprojection mask):
//               a GATHER formulation of the reference's scatter (`center[window] = max(center[window], g)`), so there
//               are no atomics on the float maps either
//
// All of it is HBM-bound byte/integer work: algorithmic bytes = 2 reads of the labels + one write of each map =
// 2 x 4 + 8 + 4 + 8 + 4 + 4 + 4 (+1) = 41 B/px with int32 labels and an int64 `sem_seg` like the reference's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgnet_hip.h"

namespace {

constexpr int MAXS = MGN_TARGETS_MAX_SEGMENTS;

// per-segment result of pt_finalize (workspace)
struct SegFin {
    double cy, cx;   // mean row / column of the segment's pixels
    int ulx, uly;    // upper-left corner of the Gaussian window (target_generator.py:124)
    int flags;       // F_*
    int pad;
};
enum { F_CENTER = 1, F_SMALL = 2 };
// seg_attr bits (host): category id (0..255) | crowd << 8 | thing << 9
__device__ __forceinline__ int attr_cat(int a) { return a & 255; }
__device__ __forceinline__ bool attr_crowd(int a) { return (a >> 8) & 1; }
__device__ __forceinline__ bool attr_thing(int a) { return (a >> 9) & 1; }

struct TgtParams {
    mgn_targets_cfg c;
    const void* pan;          // int32 [B,H,W] or uint8 [B,H,W,3]
    const int* seg_ids;       // [B, max_segments] ascending per image
    const int* seg_attr;      // [B, max_segments]
    const int* seg_count;     // [B]
    const float* gauss;       // [(6 sigma + 3)^2]
    unsigned long long* stats;  // [B][max_segments][3]: area, sum y, sum x
    SegFin* fin;              // [B][max_segments]
    long long* sem_seg; float* center; float* offset; float* sem_w; float* center_w; float* offset_w;
    uint8_t* reproj_mask;     // optional
    double* center_points;    // optional [B, max_segments, 2] (cy, cx), NaN where the segment has no centre
    long long* seg_area;      // optional [B, max_segments]
};

// label of 8 consecutive pixels starting at pixel index `pix` of image b (pix % 8 == 0 and W % 8 == 0 => aligned), or of
// `n` (< 8) pixels for the ragged tail of a row
template <bool VEC>
__device__ __forceinline__ void load_ids(const TgtParams& p, long base_px, int n, int (&id)[8]) {
    if (p.c.pan_rgb) {
        const uint8_t* s = (const uint8_t*)p.pan + base_px * 3;
        if (VEC) {
            const uint2 a = *reinterpret_cast<const uint2*>(s), b = *reinterpret_cast<const uint2*>(s + 8),
                        c = *reinterpret_cast<const uint2*>(s + 16);
            const uint32_t w[6] = {a.x, a.y, b.x, b.y, c.x, c.y};
#pragma unroll
            for (int k = 0; k < 8; ++k) {   // id = R + 256 G + 65536 B (panopticapi rgb2id) = the 3 bytes read little-endian
                const int bit = k * 24, lo = bit >> 5, sh = bit & 31;
                const uint64_t two = (uint64_t)w[lo] | ((uint64_t)(lo + 1 < 6 ? w[lo + 1] : 0u) << 32);
                id[k] = (int)((two >> sh) & 0xffffffu);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                id[k] = k < n ? (int)s[3 * k] | ((int)s[3 * k + 1] << 8) | ((int)s[3 * k + 2] << 16) : -1;
        }
    } else {
        const int* s = (const int*)p.pan + base_px;
        if (VEC) {
            const int4 a = *reinterpret_cast<const int4*>(s), b = *reinterpret_cast<const int4*>(s + 4);
            id[0] = a.x; id[1] = a.y; id[2] = a.z; id[3] = a.w; id[4] = b.x; id[5] = b.y; id[6] = b.z; id[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) id[k] = k < n ? s[k] : -1;
        }
    }
}

// index of `id` in the ascending table t[0..n) or -1
__device__ __forceinline__ int find_seg(const int* t, int n, int id) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (t[mid] < id) lo = mid + 1; else hi = mid;
    }
    return (lo < n && t[lo] == id) ? lo : -1;
}

constexpr int TILE_H = 16, TILE_W = 128;   // 256 threads x 8 pixels

// ---------------------------------------------------------------------------------------------------------------------
// pass 1: per-segment area and coordinate sums
__global__ __launch_bounds__(256) void pt_stats(TgtParams p) {
    __shared__ int ids[MAXS];
    __shared__ unsigned int cnt[MAXS], sy[MAXS], sx[MAXS];
    const int b = blockIdx.z, H = p.c.H, W = p.c.W;
    const int n = min(p.seg_count[b], p.c.max_segments);
    for (int s = threadIdx.x; s < n; s += 256) { ids[s] = p.seg_ids[(long)b * p.c.max_segments + s]; cnt[s] = 0; sy[s] = 0; sx[s] = 0; }
    __syncthreads();
    const int y = blockIdx.y * TILE_H + (threadIdx.x >> 4), x0 = blockIdx.x * TILE_W + (threadIdx.x & 15) * 8;
    if (y < H && x0 < W) {
        const int nvalid = min(8, W - x0);
        int id[8];
        const long base = ((long)b * H + y) * W + x0;
        if ((W & 7) == 0) load_ids<true>(p, base, 8, id); else load_ids<false>(p, base, nvalid, id);
        // run-length merge: one atomic triple per run of equal labels (segments are spatially coherent)
        int run_s = -1, run_n = 0, run_x = 0, prev_id = 0;
        bool have_prev = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k >= nvalid) break;
            int s;
            if (have_prev && id[k] == prev_id) s = run_s;
            else {
                if (run_s >= 0 && run_n) { atomicAdd(&cnt[run_s], run_n); atomicAdd(&sy[run_s], run_n * y); atomicAdd(&sx[run_s], run_x); }
                s = find_seg(ids, n, id[k]);
                run_s = s; run_n = 0; run_x = 0; prev_id = id[k]; have_prev = true;
            }
            if (s >= 0) { run_n += 1; run_x += x0 + k; }
        }
        if (run_s >= 0 && run_n) { atomicAdd(&cnt[run_s], run_n); atomicAdd(&sy[run_s], run_n * y); atomicAdd(&sx[run_s], run_x); }
    }
    __syncthreads();
    unsigned long long* st = p.stats + (long)b * p.c.max_segments * 3;
    for (int s = threadIdx.x; s < n; s += 256)
        if (cnt[s]) {
            atomicAdd(&st[s * 3 + 0], (unsigned long long)cnt[s]);
            atomicAdd(&st[s * 3 + 1], (unsigned long long)sy[s]);
            atomicAdd(&st[s * 3 + 2], (unsigned long long)sx[s]);
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// per-segment centre (target_generator.py:105-124)
__global__ __launch_bounds__(256) void pt_finalize(TgtParams p) {
    const int b = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
    if (s >= p.c.max_segments) return;
    const int n = min(p.seg_count[b], p.c.max_segments);
    const long i = (long)b * p.c.max_segments + s;
    SegFin f;
    f.cy = f.cx = __longlong_as_double(0x7ff8000000000000LL);
    f.ulx = f.uly = 0; f.flags = 0; f.pad = 0;
    unsigned long long area = 0;
    if (s < n) {
        const int a = p.seg_attr[i];
        area = p.stats[i * 3];
        if (attr_thing(a) && !attr_crowd(a) && area > 0) {   // :105-110 (a completely cropped instance has no centre)
            f.cy = (double)p.stats[i * 3 + 1] / (double)area;   // np.mean over integer indices: exact sum, one division
            f.cx = (double)p.stats[i * 3 + 2] / (double)area;
            const int iy = (int)rint(f.cy), ix = (int)rint(f.cx);   // np.round: half to even (:121)
            f.ulx = ix - 3 * p.c.sigma - 1;
            f.uly = iy - 3 * p.c.sigma - 1;
            f.flags = F_CENTER | ((long long)area < (long long)p.c.small_instance_area ? F_SMALL : 0);   // :113-115
        }
    }
    p.fin[i] = f;
    if (p.center_points) { p.center_points[i * 2] = f.cy; p.center_points[i * 2 + 1] = f.cx; }
    if (p.seg_area) p.seg_area[i] = (long long)area;
}

// ---------------------------------------------------------------------------------------------------------------------
// pass 2: all maps
template <bool VEC>
__device__ __forceinline__ void store8(float* dst, const float (&v)[8], int n) {
    if (VEC) {
        reinterpret_cast<float4*>(dst)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(dst)[1] = make_float4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < n) dst[k] = v[k];
    }
}

template <bool VEC>
__device__ __forceinline__ void emit_body(const TgtParams& p, const int* ids, const int* attr, const int2* wins, int nwin, int n) {
    const int b = blockIdx.z, H = p.c.H, W = p.c.W;
    const int y = blockIdx.y * TILE_H + (threadIdx.x >> 4), x0 = blockIdx.x * TILE_W + (threadIdx.x & 15) * 8;
    if (y >= H || x0 >= W) return;
    const int nvalid = VEC ? 8 : min(8, W - x0);
    const long hw = (long)H * W, px = (long)y * W + x0, base = (long)b * hw + px;
    int id[8];
    load_ids<VEC>(p, base, nvalid, id);
    const SegFin* fin = p.fin + (long)b * p.c.max_segments;
    const int G = 6 * p.c.sigma + 3;
    float ctr[8], oy[8], ox[8], sw[8], cw[8], ow[8];
    long long sem[8];
    uint8_t rm[8];
    int prev_id = 0, s = -1;
    bool have_prev = false;
    double cy = 0, cx = 0;
    int a = 0, fl = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        ctr[k] = oy[k] = ox[k] = 0.f; sw[k] = 1.f; cw[k] = ow[k] = 0.f; sem[k] = p.c.ignore_label; rm[k] = 1;
        if (k >= nvalid) continue;
        if (!have_prev || id[k] != prev_id) {
            s = find_seg(ids, n, id[k]);
            prev_id = id[k]; have_prev = true;
            if (s >= 0) { a = attr[s]; fl = fin[s].flags; if (fl & F_CENTER) { cy = fin[s].cy; cx = fin[s].cx; } }
        }
        int semantic = p.c.ignore_label;
        if (s >= 0) {
            const bool crowd = attr_crowd(a), thing = attr_thing(a);
            if (!(p.c.ignore_crowd_in_semantic && crowd)) semantic = attr_cat(a);                 // :96-97
            if (!crowd && (!p.c.ignore_stuff_in_offset || thing)) { ow[k] = 1.f; cw[k] = 1.f; }   // :98-103
            if (fl & F_CENTER) {
                if (fl & F_SMALL) sw[k] = (float)(p.c.small_instance_weight & 255);               // :113-115 (uint8 map)
                const int x = x0 + k;
                if (p.c.legacy_promotion) {   // NumPy < 2: np.float64 scalar - float32 array is evaluated in float32
                    oy[k] = (float)cy - (float)y;
                    ox[k] = (float)cx - (float)x;
                } else {                      // NEP 50: evaluated in float64, rounded when stored into the float32 map (:143-144)
                    oy[k] = (float)(cy - (double)y);
                    ox[k] = (float)(cx - (double)x);
                }
            }
        }
        if (semantic < p.c.first_thing_id) cw[k] = 1.f;                                           // :146
        sem[k] = semantic;
        rm[k] = (p.c.depth_ignore_mask[(semantic >> 5) & 7] >> (semantic & 31)) & 1u ? 0 : 1;    // dataset_mapper.py:214-216
    }
    // centre heat map: max over the Gaussians whose window covers the pixel (:117-139 as a gather)
    for (int w = 0; w < nwin; ++w) {
        const int2 ul = wins[w];
        const unsigned dy = (unsigned)(y - ul.y);
        if (dy >= (unsigned)G) continue;
        const float* grow = p.gauss + dy * G;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned dx = (unsigned)(x0 + k - ul.x);
            if (dx < (unsigned)G) ctr[k] = fmaxf(ctr[k], grow[dx]);
        }
    }
    store8<VEC>(p.center + base, ctr, nvalid);
    store8<VEC>(p.offset + (long)b * 2 * hw + px, oy, nvalid);
    store8<VEC>(p.offset + (long)b * 2 * hw + hw + px, ox, nvalid);
    store8<VEC>(p.sem_w + base, sw, nvalid);
    store8<VEC>(p.center_w + base, cw, nvalid);
    store8<VEC>(p.offset_w + base, ow, nvalid);
    if (VEC) {
        longlong2* d = reinterpret_cast<longlong2*>(p.sem_seg + base);
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = make_longlong2(sem[2 * k], sem[2 * k + 1]);
        if (p.reproj_mask) {
            uint2 m;
            m.x = rm[0] | (rm[1] << 8) | (rm[2] << 16) | ((uint32_t)rm[3] << 24);
            m.y = rm[4] | (rm[5] << 8) | (rm[6] << 16) | ((uint32_t)rm[7] << 24);
            *reinterpret_cast<uint2*>(p.reproj_mask + base) = m;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < nvalid) { p.sem_seg[base + k] = sem[k]; if (p.reproj_mask) p.reproj_mask[base + k] = rm[k]; }
    }
}

__global__ __launch_bounds__(256) void pt_emit(TgtParams p) {
    __shared__ int ids[MAXS], attr[MAXS];
    __shared__ int2 wins[MAXS];
    __shared__ int nwin;
    const int b = blockIdx.z;
    const int n = min(p.seg_count[b], p.c.max_segments);
    if (threadIdx.x == 0) nwin = 0;
    __syncthreads();
    const int ty0 = blockIdx.y * TILE_H, tx0 = blockIdx.x * TILE_W, G = 6 * p.c.sigma + 3;
    for (int s = threadIdx.x; s < n; s += 256) {
        const long i = (long)b * p.c.max_segments + s;
        ids[s] = p.seg_ids[i];
        attr[s] = p.seg_attr[i];
        const SegFin f = p.fin[i];
        if ((f.flags & F_CENTER) && f.ulx < tx0 + TILE_W && f.ulx + G > tx0 && f.uly < ty0 + TILE_H && f.uly + G > ty0)
            wins[atomicAdd(&nwin, 1)] = make_int2(f.ulx, f.uly);   // order is irrelevant: max is commutative and exact
    }
    __syncthreads();
    if ((p.c.W & 7) == 0) emit_body<true>(p, ids, attr, wins, nwin, n);
    else emit_body<false>(p, ids, attr, wins, nwin, n);
}

bool cfg_ok(const mgn_targets_cfg* c) {
    return c && c->B >= 1 && c->H >= 1 && c->W >= 1 && c->sigma >= 1 && c->sigma <= 64 && c->max_segments >= 1 &&
           c->max_segments <= MAXS && c->ignore_label >= 0 && c->ignore_label <= 255 && (long)c->H * c->W < (1L << 31) &&
           c->H <= 65536 && c->W <= 65536;
}
size_t stats_bytes(const mgn_targets_cfg* c) { return (size_t)c->B * c->max_segments * 3 * sizeof(unsigned long long); }

}  // namespace

extern "C" int mgn_panoptic_targets_workspace_bytes(const mgn_targets_cfg* cfg, size_t* bytes) {
    if (!cfg_ok(cfg) || !bytes) return MGN_EINVAL;
    *bytes = stats_bytes(cfg) + (size_t)cfg->B * cfg->max_segments * sizeof(SegFin);
    return MGN_OK;
}

extern "C" int mgn_panoptic_targets(const mgn_targets_cfg* cfg, const void* panoptic, const int32_t* seg_ids,
                                    const int32_t* seg_attr, const int32_t* seg_count, const float* gauss,
                                    int64_t* sem_seg, float* center, float* offset, float* sem_seg_weights,
                                    float* center_weights, float* offset_weights, uint8_t* reprojection_mask,
                                    double* center_points, int64_t* seg_area, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    size_t need = 0;
    if (mgn_panoptic_targets_workspace_bytes(cfg, &need) != MGN_OK) return MGN_EINVAL;
    if (!panoptic || !seg_ids || !seg_attr || !seg_count || !gauss || !sem_seg || !center || !offset || !sem_seg_weights ||
        !center_weights || !offset_weights || !workspace)
        return MGN_EINVAL;
    if (workspace_bytes < need) return MGN_ENOSPC;
    TgtParams p;
    p.c = *cfg; p.pan = panoptic; p.seg_ids = seg_ids; p.seg_attr = seg_attr; p.seg_count = seg_count; p.gauss = gauss;
    p.stats = (unsigned long long*)workspace;
    p.fin = (SegFin*)((char*)workspace + stats_bytes(cfg));
    p.sem_seg = (long long*)sem_seg; p.center = center; p.offset = offset; p.sem_w = sem_seg_weights;
    p.center_w = center_weights; p.offset_w = offset_weights; p.reproj_mask = reprojection_mask;
    p.center_points = center_points; p.seg_area = (long long*)seg_area;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(p.stats, 0, stats_bytes(cfg), st) != hipSuccess) return MGN_ELAUNCH;
    const dim3 grid((cfg->W + TILE_W - 1) / TILE_W, (cfg->H + TILE_H - 1) / TILE_H, cfg->B);
    if (grid.y > 65535 || grid.z > 65535) return MGN_EINVAL;
    hipLaunchKernelGGL(pt_stats, grid, dim3(256), 0, st, p);
    hipLaunchKernelGGL(pt_finalize, dim3((cfg->max_segments + 255) / 256, cfg->B), dim3(256), 0, st, p);
    hipLaunchKernelGGL(pt_emit, grid, dim3(256), 0, st, p);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

// instances.hip -- instance predictions from a panoptic prediction on the device (TEST.EVAL_INSTANCE).
//
// Replaces mgnet/postprocessing/instance_post_proc.py:11-72 get_instance_predictions: for every thing segment of the panoptic image
// (panoptic id // label_divisor in thing_ids; ids ascending like np.unique) the class, the mask, the score
//     mean over the mask of softmax(sem_seg)[class]  *  center_heatmap[int(mean y), int(mean x)]
// and the tight box [x_min, y_min, x_max + 1, y_max + 1] (detectron2 BitMasks.get_bounding_boxes).  The reference loops over the
// segments on the host (np.unique on a device->host copy, one boolean mask, nonzero and three reductions per segment, .item() twice);
// here ONE pass over the pixels accumulates every segment's integer moments, box and probability sum at once:
//   inst_accum   : block-private LDS hash table (label -> count, sum y, sum x, box, sum of the class probability as 32.32 fixed
//                  point: integer atomics, so the sums do not depend on the order) flushed with one global atomic per field and
//                  touched segment; the class probability is evaluated only at thing pixels (C logits per pixel, fp32 softmax)
//   inst_finalize: one block compacts the table, sorts the labels (bitonic, LDS) and emits labels / classes / scores / boxes
//   inst_masks   : pixel -> index of its label (binary search) -> mask[k][pixel] = 1   (masks zero-initialised by the caller)
// HBM-bound integer / byte work; no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgnet_hip.h"

namespace {

constexpr int TPB = 256;
constexpr int PXT = 8;                 // pixels per thread
constexpr int LSLOTS = 256;            // block-private table
constexpr int GSLOTS = 16384;          // global table (power of two); MGN_INSTANCE_MAX segments fit at load <= 1/4
constexpr unsigned long long EMPTY = 0xffffffffffffffffull;

struct Seg {                 // global accumulators of one segment
    unsigned long long cnt, sy, sx, prob;   // prob: sum of p * 2^32
    unsigned ymin, ymax, xmin, xmax;
};

__device__ __forceinline__ unsigned hash64(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33;
    return (unsigned)k;
}

__device__ __forceinline__ bool is_thing(const mgn_instance_cfg& c, long long label, int* cls) {
    if (label < 0) return false;
    const long long k = label / c.label_divisor;
    *cls = (int)k;
    return k < 64 && ((c.thing_mask >> k) & 1ull);
}

// slot of `key` in an open-addressing table of n (power of two) 64-bit keys; inserts it if absent; -1 when full
template <typename K>
__device__ __forceinline__ int find_or_insert(K* keys, int n, unsigned long long key) {
    unsigned h = hash64(key) & (n - 1);
    for (int probe = 0; probe < n; ++probe) {
        const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&keys[h]), EMPTY, key);
        if (old == EMPTY || old == key) return (int)h;
        h = (h + 1) & (n - 1);
    }
    return -1;
}

__global__ __launch_bounds__(TPB) void inst_clear(unsigned long long* gkeys, Seg* gseg, int* info) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i < GSLOTS) {
        gkeys[i] = EMPTY;
        Seg s;
        s.cnt = s.sy = s.sx = s.prob = 0; s.ymin = s.xmin = 0xffffffffu; s.ymax = s.xmax = 0;
        gseg[i] = s;
    }
    if (i < 2) info[i] = 0;
}

__global__ __launch_bounds__(TPB) void inst_accum(mgn_instance_cfg c, const float* __restrict__ logits, const long long* __restrict__ pan,
                                                  unsigned long long* gkeys, Seg* gseg, int* info) {
    __shared__ unsigned long long lkeys[LSLOTS];
    __shared__ unsigned lcnt[LSLOTS], lsy[LSLOTS], lsx[LSLOTS], lbox[4][LSLOTS];
    __shared__ unsigned long long lprob[LSLOTS];
    for (int i = threadIdx.x; i < LSLOTS; i += TPB) {
        lkeys[i] = EMPTY; lcnt[i] = lsy[i] = lsx[i] = 0; lprob[i] = 0;
        lbox[0][i] = lbox[2][i] = 0xffffffffu; lbox[1][i] = lbox[3][i] = 0;
    }
    __syncthreads();
    const long npx = (long)c.H * c.W;
    const long base = (long)blockIdx.x * (TPB * PXT);
#pragma unroll
    for (int u = 0; u < PXT; ++u) {
        const long p = base + (long)u * TPB + threadIdx.x;   // consecutive lanes = consecutive pixels
        if (p >= npx) continue;
        const long long label = pan[p];
        int cls;
        if (!is_thing(c, label, &cls) || cls >= c.C) continue;
        // softmax(sem_seg)[cls] at this pixel (F.softmax over dim 0, fp32)
        float mx = -3.0e38f;
        for (int k = 0; k < c.C; ++k) mx = fmaxf(mx, logits[(long)k * npx + p]);
        float den = 0.f, num = 0.f;
        for (int k = 0; k < c.C; ++k) {
            const float e = expf(logits[(long)k * npx + p] - mx);
            den += e;
            if (k == cls) num = e;
        }
        const float pr = num / den;
        const unsigned long long fx = (unsigned long long)((double)pr * 4294967296.0 + 0.5);
        const unsigned y = (unsigned)(p / c.W), x = (unsigned)(p - (long)y * c.W);
        const int s = find_or_insert(lkeys, LSLOTS, (unsigned long long)label);
        if (s >= 0) {
            atomicAdd(&lcnt[s], 1u); atomicAdd(&lsy[s], y); atomicAdd(&lsx[s], x); atomicAdd(&lprob[s], fx);
            atomicMin(&lbox[0][s], y); atomicMax(&lbox[1][s], y); atomicMin(&lbox[2][s], x); atomicMax(&lbox[3][s], x);
        } else {   // (more than 256 segments inside one 2048-pixel run: straight to the global table)
            const int g = find_or_insert(gkeys, GSLOTS, (unsigned long long)label);
            if (g < 0) { atomicExch(&info[1], 1); continue; }
            atomicAdd(&gseg[g].cnt, 1ull); atomicAdd(&gseg[g].sy, (unsigned long long)y); atomicAdd(&gseg[g].sx, (unsigned long long)x);
            atomicAdd(&gseg[g].prob, fx);
            atomicMin(&gseg[g].ymin, y); atomicMax(&gseg[g].ymax, y); atomicMin(&gseg[g].xmin, x); atomicMax(&gseg[g].xmax, x);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LSLOTS; i += TPB) {
        if (lkeys[i] == EMPTY) continue;
        const int g = find_or_insert(gkeys, GSLOTS, lkeys[i]);
        if (g < 0) { atomicExch(&info[1], 1); continue; }
        atomicAdd(&gseg[g].cnt, (unsigned long long)lcnt[i]); atomicAdd(&gseg[g].sy, (unsigned long long)lsy[i]);
        atomicAdd(&gseg[g].sx, (unsigned long long)lsx[i]); atomicAdd(&gseg[g].prob, lprob[i]);
        atomicMin(&gseg[g].ymin, lbox[0][i]); atomicMax(&gseg[g].ymax, lbox[1][i]);
        atomicMin(&gseg[g].xmin, lbox[2][i]); atomicMax(&gseg[g].xmax, lbox[3][i]);
    }
}

// one block of 1024 threads: occupied slots -> LDS list -> bitonic sort by label -> outputs
__global__ __launch_bounds__(1024) void inst_finalize(mgn_instance_cfg c, const float* __restrict__ heat, const unsigned long long* __restrict__ gkeys,
                                                      const Seg* __restrict__ gseg, long long* labels, int* classes, float* scores, float* boxes,
                                                      int* info) {
    __shared__ unsigned long long key[MGN_INSTANCE_MAX];
    __shared__ int slot[MGN_INSTANCE_MAX];
    __shared__ int n;
    if (threadIdx.x == 0) n = 0;
    for (int i = threadIdx.x; i < MGN_INSTANCE_MAX; i += 1024) { key[i] = EMPTY; slot[i] = -1; }
    __syncthreads();
    for (int i = threadIdx.x; i < GSLOTS; i += 1024) {
        if (gkeys[i] == EMPTY) continue;
        const int k = atomicAdd(&n, 1);
        if (k < MGN_INSTANCE_MAX) { key[k] = gkeys[i]; slot[k] = i; }
    }
    __syncthreads();
    const int total = n;
    if (total > MGN_INSTANCE_MAX && threadIdx.x == 0) info[1] = 1;
    for (int k = 2; k <= MGN_INSTANCE_MAX; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < MGN_INSTANCE_MAX; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const unsigned long long a = key[i], b = key[l];
                    if ((a > b) == up) {
                        key[i] = b; key[l] = a;
                        const int t = slot[i]; slot[i] = slot[l]; slot[l] = t;
                    }
                }
            }
            __syncthreads();
        }
    const int cnt = total < MGN_INSTANCE_MAX ? total : MGN_INSTANCE_MAX;
    if (threadIdx.x == 0) info[0] = cnt;
    for (int i = threadIdx.x; i < cnt; i += 1024) {
        const Seg s = gseg[slot[i]];
        const long long label = (long long)key[i];
        labels[i] = label;
        classes[i] = (int)(label / c.label_divisor);
        // torch.mean of the float32 row / column indices, then int(): correctly rounded float32 quotient, truncated
        const float my = (float)((double)s.sy / (double)s.cnt), mxx = (float)((double)s.sx / (double)s.cnt);
        const int cy = (int)my, cx = (int)mxx;
        const float sem = (float)(((double)s.prob / 4294967296.0) / (double)s.cnt);
        scores[i] = sem * heat[(long)cy * c.W + cx];
        boxes[4 * i + 0] = (float)s.xmin; boxes[4 * i + 1] = (float)s.ymin;
        boxes[4 * i + 2] = (float)(s.xmax + 1); boxes[4 * i + 3] = (float)(s.ymax + 1);
    }
}

__global__ __launch_bounds__(TPB) void inst_masks(mgn_instance_cfg c, const long long* __restrict__ pan, const long long* __restrict__ labels, int n,
                                                  uint8_t* masks) {
    const long npx = (long)c.H * c.W;
    const long p = (long)blockIdx.x * TPB + threadIdx.x;
    if (p >= npx) return;
    const long long label = pan[p];
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const long long v = labels[mid];
        if (v == label) { masks[(long)mid * npx + p] = 1; return; }
        if (v < label) lo = mid + 1; else hi = mid - 1;
    }
}

// tools/generate_pseudo_labels.py:100-118 -- panoptic prediction (train ids) -> the dataset's instanceIds image:
//   stuff segments (id % divisor == 0) -> id_map[class]; thing segments -> id_map[class] * divisor + instance; the three masked
//   numpy assignments of the reference applied in their order to one pixel (incl. numpy's negative index for the void label -1)
__global__ __launch_bounds__(TPB) void pseudo_ids(const long long* __restrict__ pan, long n, int div, const unsigned char* __restrict__ id_map, uint16_t* out) {
    __shared__ unsigned char lut[256];
    lut[threadIdx.x] = id_map[threadIdx.x];
    __syncthreads();
    const long p = (long)blockIdx.x * TPB + threadIdx.x;
    if (p >= n) return;
    long long v = pan[p];
    const long long m = ((v % div) + div) % div;                      // python's non-negative remainder
    if (m == 0) v = (v - m) / div;                                    // (floor division)
    if (v < div) v = lut[(int)(((v % 256) + 256) % 256)];             // id_map[v], numpy wrap-around for v = -1
    else v = (long long)lut[(int)((v / div) & 255)] * div + v % div;
    out[p] = (uint16_t)(unsigned long long)v;                         // .astype(np.uint16)
}

}  // namespace

extern "C" {

int mgn_pseudo_label_ids(const int64_t* panoptic, long n_pixels, int label_divisor, const uint8_t* id_map256, uint16_t* out, void* stream) {
    if (!panoptic || !id_map256 || !out || n_pixels < 1 || label_divisor < 257) return MGN_EINVAL;
    hipLaunchKernelGGL(pseudo_ids, dim3((unsigned)((n_pixels + TPB - 1) / TPB)), dim3(TPB), 0, (hipStream_t)stream, (const long long*)panoptic,
                       n_pixels, label_divisor, id_map256, out);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_instance_post_workspace_bytes(const mgn_instance_cfg* cfg, size_t* bytes) {
    if (!cfg || !bytes || cfg->H < 1 || cfg->W < 1 || cfg->C < 1 || cfg->C > 64 || cfg->label_divisor < 1) return MGN_EINVAL;
    *bytes = (size_t)GSLOTS * (sizeof(unsigned long long) + sizeof(Seg));
    return MGN_OK;
}

int mgn_instance_post(const mgn_instance_cfg* cfg, const float* sem_logits, const float* center_heatmap, const int64_t* panoptic,
                      int64_t* labels, int32_t* classes, float* scores, float* boxes, int32_t* info, void* workspace,
                      size_t workspace_bytes, void* stream) {
    size_t need;
    const int rc = mgn_instance_post_workspace_bytes(cfg, &need);
    if (rc != MGN_OK) return rc;
    if (!sem_logits || !center_heatmap || !panoptic || !labels || !classes || !scores || !boxes || !info || !workspace) return MGN_EINVAL;
    if (workspace_bytes < need) return MGN_ENOSPC;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* gkeys = (unsigned long long*)workspace;
    Seg* gseg = (Seg*)(gkeys + GSLOTS);
    const long npx = (long)cfg->H * cfg->W;
    hipLaunchKernelGGL(inst_clear, dim3(GSLOTS / TPB), dim3(TPB), 0, s, gkeys, gseg, info);
    hipLaunchKernelGGL(inst_accum, dim3((unsigned)((npx + TPB * PXT - 1) / (TPB * PXT))), dim3(TPB), 0, s, *cfg, sem_logits,
                       (const long long*)panoptic, gkeys, gseg, info);
    hipLaunchKernelGGL(inst_finalize, dim3(1), dim3(1024), 0, s, *cfg, center_heatmap, (const unsigned long long*)gkeys, (const Seg*)gseg,
                       (long long*)labels, classes, scores, boxes, info);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_instance_masks(const mgn_instance_cfg* cfg, const int64_t* panoptic, const int64_t* labels, int n, uint8_t* masks_zeroed, void* stream) {
    if (!cfg || !panoptic || !labels || !masks_zeroed || n < 1 || n > MGN_INSTANCE_MAX || cfg->H < 1 || cfg->W < 1) return MGN_EINVAL;
    const long npx = (long)cfg->H * cfg->W;
    hipLaunchKernelGGL(inst_masks, dim3((unsigned)((npx + TPB - 1) / TPB)), dim3(TPB), 0, (hipStream_t)stream, *cfg, (const long long*)panoptic,
                       (const long long*)labels, n, masks_zeroed);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

}  // extern "C"

// boundary_probe -- does a kernel that is RUNNING on one queue lose stores when other queues start / finish kernels beside it?
// (Kernel boundaries carry cache maintenance: write-back of the L2s at the end of a kernel, invalidate at the start of the next.  The replayed
// training step issues ~650 boundaries per 25 ms on three queues; profiles/r06_determinism.txt section 4: the differing launches' OUTPUT holds
// the previous step's values in a few cache lines, their inputs are intact.)
//   stream 1: Long(out, tag)   -- writes every word of a big buffer (value = tag ^ index), stretched by a little arithmetic per word
//             Check(out, tag)  -- behind it on the same stream: counts wrong words, and how many of them hold the PREVIOUS tag's value
//   streams 2 .. Q: small kernels back to back, each writing its own small buffer (their boundaries are what is being tested);
//             optionally chained across streams by events (barrier packets, as the launch plan's edges are)
//   store mode 0: plain stores   1: write-through stores (sc0 sc1)   2: non-temporal stores
// usage: boundary_probe [iters=200] [words_log2=25] [side queues=2] [small kernels per iteration=60] [store mode=0] [events=0] [spin=8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>
__device__ __forceinline__ void put(unsigned* p, unsigned v) {
    if (MODE == 1) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (MODE == 2) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <int MODE>
__global__ __launch_bounds__(256) void Long(unsigned* out, size_t n, unsigned tag, int spin) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float a = (float)(i & 1023);
        for (int k = 0; k < spin; ++k) a = a * 1.0001f + 0.25f;   // stretch the kernel: the lines stay dirty for a while
        put<MODE>(out + i, (tag ^ (unsigned)i) + (a < 0.f ? 1u : 0u));
    }
}
__global__ __launch_bounds__(256) void Check(const unsigned* out, size_t n, unsigned tag, unsigned prev, unsigned long long* err) {
    unsigned long long bad = 0, stale = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned v = out[i];
        if (v != (tag ^ (unsigned)i)) { ++bad; stale += v == (prev ^ (unsigned)i); }
    }
    if (bad) { atomicAdd(err, bad); atomicAdd(err + 1, stale); atomicAdd(err + 2, 1ull); }
}
__global__ __launch_bounds__(256) void Small(unsigned* y, int n, unsigned v) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = y[i] * 3u + v;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200, lg = argc > 2 ? atoi(argv[2]) : 25, nq = argc > 3 ? atoi(argv[3]) : 2;
    const int nsmall = argc > 4 ? atoi(argv[4]) : 60, mode = argc > 5 ? atoi(argv[5]) : 0, events = argc > 6 ? atoi(argv[6]) : 0;
    const int spin = argc > 7 ? atoi(argv[7]) : 8;
    const size_t n = (size_t)1 << lg;
    unsigned* out; unsigned long long* err;
    hipMalloc(&out, n * 4); hipMemset(out, 0, n * 4);
    hipMalloc(&err, 24); hipMemset(err, 0, 24);
    hipStream_t s1; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    std::vector<hipStream_t> sq(nq); std::vector<unsigned*> yb(nq); std::vector<hipEvent_t> ev(nq);
    const int ysz = 1 << 18;
    for (int q = 0; q < nq; ++q) {
        hipStreamCreateWithFlags(&sq[q], hipStreamNonBlocking);
        hipMalloc(&yb[q], (size_t)ysz * 4); hipMemset(yb[q], 0, (size_t)ysz * 4);
        hipEventCreateWithFlags(&ev[q], hipEventDisableTiming);
    }
    hipDeviceSynchronize();
    unsigned prev = 0;
    for (int it = 1; it <= iters; ++it) {
        const unsigned tag = (unsigned)it * 2654435761u;
        if (mode == 1) Long<1><<<2048, 256, 0, s1>>>(out, n, tag, spin);
        else if (mode == 2) Long<2><<<2048, 256, 0, s1>>>(out, n, tag, spin);
        else Long<0><<<2048, 256, 0, s1>>>(out, n, tag, spin);
        for (int k = 0; k < nsmall; ++k)
            for (int q = 0; q < nq; ++q) {
                if (events && nq > 1 && k % 4 == 0) hipStreamWaitEvent(sq[q], ev[(q + 1) % nq], 0);
                Small<<<(k % 3 == 0) ? 256 : 16, 256, 0, sq[q]>>>(yb[q], ysz, tag + k);
                if (events && nq > 1 && k % 4 == 0) hipEventRecord(ev[q], sq[q]);
            }
        Check<<<1024, 256, 0, s1>>>(out, n, tag, prev, err);
        prev = tag;
    }
    hipDeviceSynchronize();
    unsigned long long h[3] = {0, 0, 0};
    hipMemcpy(h, err, 24, hipMemcpyDeviceToHost);
    printf("store mode %d, %d iterations of a %zu-MB kernel beside %d side queues x %d small kernels (events %d, spin %d): %llu wrong words, "
           "%llu of them the previous iteration's value, in %llu (block, iteration) cases\n",
           mode, iters, n * 4 >> 20, nq, nsmall, events, spin, h[0], h[1], h[2]);
    return 0;
}

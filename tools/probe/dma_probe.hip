// probe: achievable read bandwidth of the row-march access pattern (256 blocks x 8 waves, 17 KB row pieces at 64 KB pitch)
// with LDS-DMA vs plain loads, against a linear streaming read.  hipcc --offload-arch=gfx950 -O3 dma_probe.hip -o dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
constexpr int N = 8, H = 256, W = 512, C = 64;           // bf16 NHWC
constexpr int INROW = 136 * 128;

template <int F, int MODE>   // MODE 0: LDS-DMA, 1: plain 16-byte loads
__global__ __launch_bounds__(512, 1) void march(const uint16_t* __restrict__ in, float* out, int rows_per_chunk, int strips, int chunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int s = blockIdx.x;
    const int strip = s % strips; s /= strips;
    const int chunk = s % chunks, n = s / chunks;
    const int ow0 = strip * 128, r0 = chunk * rows_per_chunk, r1 = min(r0 + rows_per_chunk, H);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(in), 0, (uint32_t)((size_t)N * H * W * C * 2), 0x00020000);
    constexpr int OOB = (int)0x80000000;
    int voff[3];
    for (int i = 0; i < 3; ++i) {
        const int j = wave + 8 * i, px = 8 * j + (lane >> 3), iw = ow0 - 1 + px;
        voff[i] = (j < 17 && iw >= 0 && iw < W) ? (iw * 64 + (lane & 7) * 8) * 2 : OOB;
    }
    float acc = 0.f;
    auto issue = [&](int ih, int slot) {
        const bool ok = ih >= 0 && ih < H;
        const int soff = ok ? ((n * H + ih) * W) * 128 : 0;
        for (int i = 0; i < 3; ++i) {
            const int j = wave + 8 * i;
            if (j < 17) {
                if (MODE == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(sm + slot * INROW + j * 1024), 16, ok ? voff[i] : OOB, soff, 0, 0);
                else { auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? voff[i] : OOB, soff, 0); acc += __uint_as_float(v[0]) + __uint_as_float(v[3]); }
            }
        }
    };
    constexpr int NR = F + 4;
    for (int k = 0; k < F + 3; ++k) issue(r0 - 1 + k, k);
    int si = 0;
    for (int r = r0; r < r1; ++r) {
        if (MODE == 0) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * F) : "memory"); __builtin_amdgcn_s_barrier(); }
        issue(r + 2 + F, (si + F + 3) % NR);
        if (MODE == 0) acc += ((float*)sm)[(si * INROW) / 4 + tid];
        si = (si + 1) % NR;
    }
    if (acc == 12345.f) out[0] = acc;
}

__global__ void linear(const uint4* __restrict__ in, float* out, long nvec) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) { const uint4 v = in[i]; acc += __uint_as_float(v[0]) + __uint_as_float(v[3]); }
    if (acc == 12345.f) out[0] = acc;
}

template <typename Fn> float timeit(Fn f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a); for (int i = 0; i < 20; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 20;
}

int main() {
    const size_t bytes = (size_t)N * H * W * C * 2;
    uint16_t* d; float* o; hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 0, bytes);
    const int strips = 4;
    for (int rpc : {32, 16, 64}) {
        const int chunks = H / rpc, nb = N * strips * chunks;
        hipFuncSetAttribute(reinterpret_cast<const void*>(&march<1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * INROW);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&march<3, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * INROW);
        float t1 = timeit([&] { march<1, 0><<<nb, 512, 5 * INROW>>>(d, o, rpc, strips, chunks); });
        float t3 = timeit([&] { march<3, 0><<<nb, 512, 7 * INROW>>>(d, o, rpc, strips, chunks); });
        float tp = timeit([&] { march<1, 1><<<nb, 512, 5 * INROW>>>(d, o, rpc, strips, chunks); });
        printf("rows/chunk %2d blocks %4d : LDS-DMA F=1 %.1f us (%.2f TB/s)  F=3 %.1f us (%.2f TB/s)  plain loads %.1f us (%.2f TB/s)\n", rpc, nb,
               t1 * 1e3, bytes / t1 / 1e9, t3 * 1e3, bytes / t3 / 1e9, tp * 1e3, bytes / tp / 1e9);
    }
    float tl = timeit([&] { linear<<<2048, 256>>>((const uint4*)d, o, bytes / 16); });
    printf("linear streaming read: %.1f us (%.2f TB/s)\n", tl * 1e3, bytes / tl / 1e9);
    return 0;
}

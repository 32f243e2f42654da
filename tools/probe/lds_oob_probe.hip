// lds_oob_probe -- is the LDS destination of an LDS-DMA load (`buffer_load ... lds`) checked against the block's LDS allocation?
// An "attacker" block with 4 KB of LDS issues LDS-DMA loads whose destination lies 4 .. 64 KB BEHIND its own allocation while canary
// blocks (8 KB of LDS each, other stream) keep re-checking a pattern in theirs.  If the hardware does not clamp, a canary that shares a
// CU with an attacker sees its pattern change.  Control: the same loads aimed inside the attacker's own 4 KB.
// (profiles/r06_determinism.txt: the kernels whose output differs in a replay -- reproj_march, conv3x3_c64_res, conv_wgrad_reduce_batch --
//  all keep state in LDS; every convolution kernel of the library fills its LDS by DMA.)
// usage: lds_oob_probe [rounds=20] [mode: 0 = control, 1 = out of range]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void* lds_ptr;

__global__ __launch_bounds__(64) void canary(unsigned* bad, int words, long long ticks, unsigned seed) {
    extern __shared__ unsigned lds[];
    const unsigned tag = seed ^ (blockIdx.x * 2654435761u);
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = tag + i;
    __syncthreads();
    const long long t0 = wall_clock64();
    unsigned nb = 0;
    while (wall_clock64() - t0 < ticks) {
        for (int i = threadIdx.x; i < words; i += blockDim.x) nb += lds[i] != tag + i;
        __builtin_amdgcn_s_sleep(8);
    }
    if (nb) atomicAdd(bad, 1u);
}

__global__ __launch_bounds__(64) void attacker(const unsigned* src, int lo, int hi, long long ticks, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char own[];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, 1 << 20, 0x00020000);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
        for (int off = lo; off < hi; off += 256)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(own + off), 4, (int)threadIdx.x * 4, 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (reinterpret_cast<unsigned*>(own)[threadIdx.x] == 0x12345u) sink[0] = 1;   // keep the LDS alive
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20, mode = argc > 2 ? atoi(argv[2]) : 1;
    unsigned *bad, *src, *sink;
    hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    hipMalloc(&sink, 4); hipMemset(sink, 0, 4);
    hipMalloc(&src, 1 << 20); hipMemset(src, 0xEE, 1 << 20);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const long long ticks = 20000;   // 100 MHz wall clock: 200 us
    for (int r = 0; r < rounds; ++r) {
        canary<<<4096, 64, 8192, s1>>>(bad, 2048, ticks, 77u + r);
        if (mode == 1) attacker<<<4096, 64, 4096, s2>>>(src, 4096, 65536, ticks, sink);
        else attacker<<<4096, 64, 4096, s2>>>(src, 0, 4096, ticks, sink);
    }
    hipDeviceSynchronize();
    unsigned h = 0;
    hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("%s: %d rounds of 4096 canary blocks beside 4096 attacker blocks: %u canary blocks saw their LDS change (%s)\n",
           mode == 1 ? "LDS-DMA destinations 4 .. 64 KB behind the block's allocation" : "control (destinations inside the allocation)", rounds, h,
           hipGetErrorString(hipGetLastError()));
    return 0;
}

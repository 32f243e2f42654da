// oob_order_probe -- does `s_waitcnt vmcnt(N)` still mean "all but the N YOUNGEST LDS-DMA loads have landed" when some of the loads are
// out-of-range (all lanes beyond the buffer's num_records: no memory request, zeros written to LDS)?  csrc/conv_win.hip and conv.hip keep
// the number of loads per pipeline step constant with such "dummy" loads and wait with counted vmcnt.
//   test 0: [real cold load -> LDS A][dummy -> LDS B]  s_waitcnt vmcnt(1)  read A: must hold the loaded data
//   test 1: [real cold load -> LDS A][real hot load -> LDS B]  vmcnt(1)  read A   (control: two real loads)
//   test 2: [real cold -> A][dummy][dummy][dummy] vmcnt(3) read A
//   test 3: [real cold -> A][partially out-of-range load (lanes 32..63 beyond the buffer) -> B] vmcnt(1) read A
// A is pre-filled with a sentinel; `bad` counts lanes that still see the sentinel after the wait.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_ptr;
constexpr unsigned SENT = 0xDEADBEEFu;
template <int TEST>
__global__ __launch_bounds__(64) void probe(const uint32_t* src, size_t bytes, int iters, unsigned long long* bad, unsigned long long* seen) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src), 0, (uint32_t)bytes, 0x00020000);
    constexpr int OOB = (int)0x80000000;
    const int lane = threadIdx.x;
    unsigned long long nb = 0, ns = 0;
    const size_t lines = bytes / 1024;   // 1-KB pieces
    for (int it = 0; it < iters; ++it) {
        reinterpret_cast<uint4*>(lds)[lane] = make_uint4(SENT, SENT, SENT, SENT);
        reinterpret_cast<uint4*>(lds + 1024)[lane] = make_uint4(SENT, SENT, SENT, SENT);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // a cold piece: a different KB every iteration and block, scattered through the buffer
        const size_t piece = ((size_t)blockIdx.x * 2654435761ull + (size_t)it * 40503ull) % lines;
        const int voff = (int)(piece * 1024 + lane * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds, 16, voff, 0, 0, 0);
        if (TEST == 0) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + 1024), 16, OOB, 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        } else if (TEST == 1) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + 1024), 16, lane * 16, 0, 0, 0);   // the buffer's first KB: hot
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        } else if (TEST == 2) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + 1024), 16, OOB, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + 1024), 16, OOB, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + 1024), 16, OOB, 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + 1024), 16, lane < 32 ? lane * 16 : OOB, 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        }
        const uint4 v = reinterpret_cast<const uint4*>(lds)[lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        nb += (v.x == SENT) + (v.y == SENT) + (v.z == SENT) + (v.w == SENT);
        ns += (v.x == (unsigned)(piece * 256 + lane * 4));   // the source holds its own word index
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (nb) atomicAdd(bad, nb);
    atomicAdd(seen, ns);
}
__global__ void fill(uint32_t* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
__global__ void busy(float* y, size_t n) {   // memory traffic beside the probe
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = y[i] * 1.0001f + 0.5f;
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000, blocks = argc > 2 ? atoi(argv[2]) : 4096, with_busy = argc > 3 ? atoi(argv[3]) : 1;
    const size_t bytes = (size_t)1 << 30;   // 1 GiB source (< 2^31: 32-bit buffer offsets)
    uint32_t* src; float* y; unsigned long long* cnt;
    hipMalloc(&src, bytes); hipMalloc(&y, (size_t)1 << 30); hipMalloc(&cnt, 64);
    fill<<<4096, 256>>>(src, bytes / 4);
    hipMemset(y, 0, (size_t)1 << 30);
    hipStream_t s, t;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&t, hipStreamNonBlocking);
    hipDeviceSynchronize();
    const char* names[4] = {"real cold + 1 dummy, vmcnt(1)", "real cold + real hot, vmcnt(1) (control)", "real cold + 3 dummies, vmcnt(3)",
                            "real cold + half-out-of-range load, vmcnt(1)"};
    for (int test = 0; test < 4; ++test) {
        hipMemset(cnt, 0, 64);
        if (with_busy) for (int k = 0; k < 40; ++k) busy<<<2048, 256, 0, t>>>(y, ((size_t)1 << 30) / 4);
        if (test == 0) probe<0><<<blocks, 64, 2048, s>>>(src, bytes, iters, cnt, cnt + 1);
        if (test == 1) probe<1><<<blocks, 64, 2048, s>>>(src, bytes, iters, cnt, cnt + 1);
        if (test == 2) probe<2><<<blocks, 64, 2048, s>>>(src, bytes, iters, cnt, cnt + 1);
        if (test == 3) probe<3><<<blocks, 64, 2048, s>>>(src, bytes, iters, cnt, cnt + 1);
        hipDeviceSynchronize();
        unsigned long long h[2];
        hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost);
        printf("test %d (%s): %llu words still held the sentinel after the wait, %llu of %llu lane-loads verified\n", test, names[test], h[0], h[1],
               (unsigned long long)blocks * iters * 64);
    }
    return 0;
}

// l2_probe -- do two kernels launched back to back on ONE stream see each other's writes when the addresses are the same every iteration
// (as in a launch-plan replay: static memory) and other streams keep the chip busy?  W(it) writes X[i] = it; R(it) reads X through a block
// permutation (a block reads what ANOTHER block -- most likely on another XCD -- wrote) and counts values != it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void W(float* x, int n, float v) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = v;
}
__global__ void R(const float* x, int n, float v, unsigned* err, int shift) {
    const int nb = gridDim.x, b = (blockIdx.x + shift) % nb;   // read the region block b wrote
    unsigned bad = 0;
    for (int i = b * blockDim.x + threadIdx.x; i < n; i += nb * blockDim.x) bad += x[i] != v;
    if (bad) atomicAdd(err, bad);
}
// per-block partial rows of `row` floats (16 bytes for row = 4): neighbouring BLOCKS -- usually on different XCDs -- write into one 128-byte line
__global__ void Wp(float* x, int row, float v) {
    if (threadIdx.x < row) x[(size_t)blockIdx.x * row + threadIdx.x] = v + threadIdx.x;
}
__global__ void Rp(const float* x, int nblk, int row, float v, unsigned* err) {
    unsigned bad = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nblk * row; i += gridDim.x * blockDim.x) bad += x[i] != v + (i % row);
    if (bad) atomicAdd(err, bad);
}
struct WP { float* x; int n; float v; };
struct RP { const float* x; int n; float v; unsigned* err; int shift; };
// the same two kernels with their pointers INSIDE by-value structs (as most kernels of libmgnet_hip.so take them): the runtime's
// dependency tracker sees no memory object among the arguments
__global__ void Ws(WP p) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += gridDim.x * blockDim.x) p.x[i] = p.v;
}
__global__ void Rs(RP p) {
    const int nb = gridDim.x, b = (blockIdx.x + p.shift) % nb;
    unsigned bad = 0;
    for (int i = b * blockDim.x + threadIdx.x; i < p.n; i += nb * blockDim.x) bad += p.x[i] != p.v;
    if (bad) atomicAdd(p.err, bad);
}
__global__ void busy(float* y, int n, int rounds) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float a = y[i];
        for (int r = 0; r < rounds; ++r) a = a * 1.0001f + 0.5f;
        y[i] = a;
    }
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, n = argc > 2 ? atoi(argv[2]) : 1 << 16, with_busy = argc > 3 ? atoi(argv[3]) : 1;
    const int in_struct = argc > 4 ? atoi(argv[4]) : 0;
    float *x, *y; unsigned* err;
    hipMalloc(&x, (size_t)n * 128 * sizeof(float)); hipMalloc(&y, (4 << 20) * sizeof(float)); hipMalloc(&err, 4);
    hipMemset(err, 0, 4); hipMemset(y, 0, (4 << 20) * sizeof(float));
    hipStream_t s, t, u;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&t, hipStreamNonBlocking); hipStreamCreateWithFlags(&u, hipStreamNonBlocking);
    for (int it = 1; it <= iters; ++it) {
        if (with_busy && it % 4 == 0) { busy<<<300, 256, 0, t>>>(y, 4 << 20, 64); busy<<<77, 256, 0, u>>>(y + (2 << 20), 1 << 20, 200); }
        if (in_struct >= 2) {   // partial rows: in_struct = 2 + log2(row floats) -> 2: 1 float, 4: 4 floats (16 B), 7: 32 floats (128 B)
            const int row = 1 << (in_struct - 2), nblk = n;
            Wp<<<nblk, 64, 0, s>>>(x, row, (float)it);
            Rp<<<argc > 5 ? atoi(argv[5]) : 1, 256, 0, s>>>(x, nblk, row, (float)it, err);
        } else if (in_struct) {
            Ws<<<256, 256, 0, s>>>(WP{x, n, (float)it});
            Rs<<<256, 256, 0, s>>>(RP{x, n, (float)it, err, 37 + it % 101});
        } else {
            W<<<256, 256, 0, s>>>(x, n, (float)it);
            R<<<256, 256, 0, s>>>(x, n, (float)it, err, 37 + it % 101);
        }
    }
    hipDeviceSynchronize();
    unsigned h = 0; hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost);
    printf("iters %d n %d busy %d pointers %s: stale reads %u\n", iters, n, with_busy, in_struct ? "inside by-value structs" : "as kernel arguments", h);
    return 0;
}

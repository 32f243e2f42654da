// xstream_probe -- producer on stream A, consumer on stream B, ordered ONLY by hipEventRecord / hipStreamWaitEvent (and back), static
// addresses, launches issued back to back without host synchronisation, other streams busy: does the consumer ever read the value of the
// PREVIOUS iteration?  (engine/plan.py replays the training step this way: ~45 cross-stream events per step.)
//   mode 0: W(A) -> event -> R(B) -> event -> next W(A)                    (ping-pong, one pair)
//   mode 1: the same with NP independent pairs on 2*NP streams at once      (many queues busy)
//   mode 2: W(A) -> event -> {R1(B), R2(C)} both read; next W waits for both  (fan-out)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void W(unsigned* x, int n, unsigned v) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = v + (unsigned)i;
}
__global__ void R(const unsigned* x, int n, unsigned v, unsigned* err, int shift) {
    const int nb = gridDim.x, b = (blockIdx.x + shift) % nb;   // read what ANOTHER block (most likely on another XCD) wrote
    unsigned bad = 0;
    for (int i = b * blockDim.x + threadIdx.x; i < n; i += nb * blockDim.x) bad += x[i] != v + (unsigned)i;
    if (bad) atomicAdd(err, bad);
}
__global__ void busy(float* y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = y[i] * 1.0001f + 0.5f;
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, n = argc > 2 ? atoi(argv[2]) : 1 << 14, mode = argc > 3 ? atoi(argv[3]) : 0;
    const int np = mode == 1 ? 4 : 1, wblocks = argc > 4 ? atoi(argv[4]) : 8, with_busy = argc > 5 ? atoi(argv[5]) : 1;
    std::vector<unsigned*> x(np); unsigned* err; float* y;
    hipMalloc(&err, 4); hipMemset(err, 0, 4);
    hipMalloc(&y, (size_t)256 << 20); hipMemset(y, 0, (size_t)256 << 20);
    std::vector<hipStream_t> A(np), B(np), C(np);
    std::vector<hipEvent_t> ew(np), er(np), er2(np);
    for (int p = 0; p < np; ++p) {
        hipMalloc(&x[p], (size_t)n * 4); hipMemset(x[p], 0, (size_t)n * 4);
        hipStreamCreateWithFlags(&A[p], hipStreamNonBlocking); hipStreamCreateWithFlags(&B[p], hipStreamNonBlocking); hipStreamCreateWithFlags(&C[p], hipStreamNonBlocking);
        hipEventCreateWithFlags(&ew[p], hipEventDisableTiming); hipEventCreateWithFlags(&er[p], hipEventDisableTiming); hipEventCreateWithFlags(&er2[p], hipEventDisableTiming);
    }
    hipStream_t bs; hipStreamCreateWithFlags(&bs, hipStreamNonBlocking);
    hipDeviceSynchronize();
    for (int it = 1; it <= iters; ++it) {
        if (with_busy && it % 8 == 0) busy<<<1024, 256, 0, bs>>>(y, ((size_t)256 << 20) / 4);
        for (int p = 0; p < np; ++p) {
            const unsigned v = (unsigned)it * 1000003u;
            W<<<wblocks, 256, 0, A[p]>>>(x[p], n, v);
            hipEventRecord(ew[p], A[p]);
            hipStreamWaitEvent(B[p], ew[p], 0);
            R<<<wblocks, 256, 0, B[p]>>>(x[p], n, v, err, 1 + it % 5);
            hipEventRecord(er[p], B[p]);
            hipStreamWaitEvent(A[p], er[p], 0);
            if (mode == 2) {
                hipStreamWaitEvent(C[p], ew[p], 0);
                R<<<wblocks, 256, 0, C[p]>>>(x[p], n, v, err, 2 + it % 3);
                hipEventRecord(er2[p], C[p]);
                hipStreamWaitEvent(A[p], er2[p], 0);
            }
        }
    }
    hipDeviceSynchronize();
    unsigned h = 0;
    hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost);
    printf("mode %d, %d iterations, %d words, %d blocks per kernel, busy %d: %u stale words read\n", mode, iters, n, wblocks, with_busy, h);
    return 0;
}

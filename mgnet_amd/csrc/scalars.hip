// scalars.hip -- the scalar tail of MGNet.forward (gfx950): the uncertainty weighting of the task losses, mg_net.py:360-372
//     loss_k <- tau_k * exp(-log_vars[k]) * loss_k + 0.5 * log_vars[k],     tau = 1 for loss_sem_seg, else 0.5
// evaluated for all tasks in ONE launch (and one for the backward) straight from the device scalars the loss kernels left behind: the
// launch carries the pointers, nothing is stacked, sliced or scattered by tensor ops.  In the reference this is ~10 scalar ATen ops per
// task and two .item() host synchronisations per task for the event storage; here the "_raw" / "_uncertainty" values stay on the device.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgnet_hip.h"

namespace {

struct UncPtrs { const float* p[MGN_MAX_TASKS]; };
MGN_PLAN_RO(UncPtrs, MGN_RO(p))

__global__ void uncertainty_fwd(UncPtrs raw, int n, const float* __restrict__ log_vars, unsigned tau_one_mask, float* __restrict__ weighted,
                                float* __restrict__ unc) {
    const int k = threadIdx.x;
    if (k >= n) return;
    const float lv = log_vars[k], tau = ((tau_one_mask >> k) & 1u) ? 1.0f : 0.5f;
    weighted[k] = tau * expf(-lv) * raw.p[k][0] + 0.5f * lv;
    unc[k] = expf(lv);
}

// d raw_k = tau_k exp(-lv_k) g_k ;  d lv_k = (0.5 - tau_k exp(-lv_k) raw_k) g_k ;  g_k = 0 where the output was not used
__global__ void uncertainty_bwd(UncPtrs raw, UncPtrs g, int n, int n_lv, const float* __restrict__ log_vars, unsigned tau_one_mask,
                                float* __restrict__ d_raw, float* __restrict__ d_lv) {
    const int k = threadIdx.x;
    if (k >= n_lv) return;
    if (k >= n) { d_lv[k] = 0.f; return; }
    const float lv = log_vars[k], tau = ((tau_one_mask >> k) & 1u) ? 1.0f : 0.5f;
    const float gk = g.p[k] ? g.p[k][0] : 0.f, e = tau * expf(-lv);
    d_raw[k] = e * gk;
    d_lv[k] = (0.5f - e * raw.p[k][0]) * gk;
}

// small host -> device table uploads as a KERNEL that reads pinned host memory through the fabric (zero copy): an async hipMemcpy of a
// few KB goes through the copy path of the runtime, whose hand-over with the compute queue showed up as 0.3-0.7 ms holes in the step's
// kernel trace (profiles/r04_v1_kernel_trace.txt: "... -> __amd_rocclr_copyBuffer"); a 240 KB table takes ~15 us this way
__global__ __launch_bounds__(256) void copy_from_host_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L)
        dst[i] = __builtin_nontemporal_load(src + i);
}

}  // namespace

extern "C" {

int mgn_copy_from_host(void* dst_dev, const void* src_pinned_host, size_t nbytes, void* stream) {
    if (!dst_dev || !src_pinned_host || nbytes == 0 || nbytes % 4 || ((uintptr_t)dst_dev | (uintptr_t)src_pinned_host) % 4) return MGN_EINVAL;
    const long n = (long)(nbytes / 4);
    const unsigned blocks = (unsigned)((n + 255) / 256 < 64 ? (n + 255) / 256 : 64);
    hipLaunchKernelGGL(copy_from_host_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)src_pinned_host, (uint32_t*)dst_dev, n);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_uncertainty_fwd(const float* const* raw_losses, int n, const float* log_vars, unsigned tau_one_mask, float* weighted, float* uncertainty,
                        void* stream) {
    if (!raw_losses || n < 1 || n > MGN_MAX_TASKS || !log_vars || !weighted || !uncertainty) return MGN_EINVAL;
    UncPtrs r;
    for (int k = 0; k < MGN_MAX_TASKS; ++k) {
        r.p[k] = k < n ? raw_losses[k] : nullptr;
        if (k < n && !r.p[k]) return MGN_EINVAL;
    }
    hipLaunchKernelGGL(uncertainty_fwd, dim3(1), dim3(64), 0, (hipStream_t)stream, r, n, log_vars, tau_one_mask, weighted, uncertainty);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_uncertainty_bwd(const float* const* raw_losses, const float* const* grads, int n, int n_log_vars, const float* log_vars,
                        unsigned tau_one_mask, float* d_raw, float* d_log_vars, void* stream) {
    if (!raw_losses || !grads || n < 1 || n > MGN_MAX_TASKS || n_log_vars < n || n_log_vars > 64 || !log_vars || !d_raw || !d_log_vars) return MGN_EINVAL;
    UncPtrs r, g;
    for (int k = 0; k < MGN_MAX_TASKS; ++k) {
        r.p[k] = k < n ? raw_losses[k] : nullptr;
        g.p[k] = k < n ? grads[k] : nullptr;
        if (k < n && !r.p[k]) return MGN_EINVAL;
    }
    hipLaunchKernelGGL(uncertainty_bwd, dim3(1), dim3(64), 0, (hipStream_t)stream, r, g, n, n_log_vars, log_vars, tau_one_mask, d_raw, d_log_vars);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

}  // extern "C"

// p2p.hip -- the cross-rank exchange of the SyncBN statistics as peer-to-peer stores over xGMI (gfx950, one process per GPU of a node).
//
// Replaces, for the 68 InPlaceABNSync sites (mgnet/modeling/res_net.py:35,49,59,103, layers.py:63,71,117,209,242,253,291; inplace_abn's
// own all-gather / all-reduce in the reference), the two torch.distributed collectives per layer and step: one all_gather of 3*C floats
// in the forward, one all_reduce of 2*C floats in the backward = 136 blocking, latency-bound collectives per step, each a Python ->
// ProcessGroupNCCL -> RCCL kernel round trip with an event hand-shake on either side.  The payload is a few KB; xGMI is point-to-point
// (7 links per GPU), so the natural form is a PUSH: every rank stores its row into a mailbox in EVERY peer's memory and raises a flag
// there, then waits until the flags of all peers have arrived in its own mailbox -- one small kernel on the compute stream, no
// collective library, no host hand-shake, ~one xGMI store latency.
//
//   mailbox (per rank, fine-grained device memory, IPC-mapped by every peer):
//       data [CHANNELS][RING][world][SLOT] fp32      flags [CHANNELS][RING][world] u32 (64-byte apart)
//   channel = the stream the exchange is issued on (the step runs its branches on up to three streams; exchanges of ONE stream are
//   ordered, which is what bounds the ring: rank A can post exchange j of a stream only after it has seen every peer's flag of j-1, and a
//   peer raises j-1 only after it has finished READING j-2 -- so a slot of a ring of >= 2 is never overwritten while it is being read;
//   the ring has 4).  seq counts the exchanges of a channel on the host; every rank issues the same exchanges in the same order.
//   Memory model: payload stores, __threadfence_system(), flag store with release / system scope; the reader spins with acquire /
//   system-scope loads (s_sleep between polls), fences, then reads its own (local) mailbox.  The combination over ranks runs in rank
//   order on every rank: bit-identical results everywhere, as with the gather + local combine it replaces.
//   A wait that exceeds its budget (a peer that died) sets status[0] (host-visible) and fills the output with NaN: see the kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "mgnet_hip.h"

namespace {

constexpr int P2P_CHANNELS = MGN_P2P_CHANNELS, P2P_RING = 4, P2P_SLOT = MGN_P2P_SLOT_FLOATS, P2P_MAXW = MGN_P2P_MAX_WORLD;
constexpr int FLAG_STRIDE = 16;   // u32 words between flags (64 bytes)

struct Peers { float* mail[P2P_MAXW]; };

__host__ __device__ inline size_t data_floats() { return (size_t)P2P_CHANNELS * P2P_RING * P2P_MAXW * P2P_SLOT; }
__host__ __device__ inline size_t flag_words() { return (size_t)P2P_CHANNELS * P2P_RING * P2P_MAXW * FLAG_STRIDE; }

__device__ __forceinline__ float* slot_data(float* mail, int chan, int slot, int src) {
    return mail + (((size_t)chan * P2P_RING + slot) * P2P_MAXW + src) * P2P_SLOT;
}
__device__ __forceinline__ uint32_t* slot_flag(float* mail, int chan, int slot, int src) {
    return reinterpret_cast<uint32_t*>(mail + data_floats()) + (((size_t)chan * P2P_RING + slot) * P2P_MAXW + src) * FLAG_STRIDE;
}

// grid = world blocks: block b pushes this rank's payload into peer b's mailbox; block 0 then waits for every rank's flag in the OWN
// mailbox and writes  REDUCE ? out[i] = sum_r data_r[i] : out[r][i] = data_r[i]
//
// seq: the exchange number of the channel.  Either the host counts it (`seq`), or -- seq_ctr != null -- the DEVICE does: block b keeps
// its own counter seq_ctr[chan][b] (every launch has exactly `world` blocks and each advances its own, so all blocks of a launch, and
// the same launch on every rank, agree without talking to each other).  The device form has no per-launch host value, which is what
// lets a recorded step be replayed (csrc/plan.hip).
// A wait that runs out of its budget (a peer that died or diverged in its launch order) is LOUD: `status` (host-visible pinned memory,
// polled by the trainer every step without a device synchronisation) is set, and the output is filled with NaN so that the losses of
// this very step are non-finite -- never silently wrong statistics in the forward, the running means and the optimizer.
template <bool REDUCE>
__global__ __launch_bounds__(256) void p2p_exchange(Peers peers, int world, int rank, int chan, uint32_t seq, uint32_t* seq_ctr,
                                                    const float* __restrict__ payload, int n, float* __restrict__ out, int* status,
                                                    long long budget_ticks) {
    const int tid = threadIdx.x, b = blockIdx.x;
    __shared__ int bad;
    if (seq_ctr) {
        uint32_t* c = seq_ctr + chan * P2P_MAXW + b;
        seq = *c + 1u;          // (block-uniform: read by every thread before thread 0 advances it behind the barrier below)
    }
    const int slot = (int)(seq % P2P_RING);
    if (tid == 0) bad = 0;
    {   // ---- post to peer b ----
        float* dst = slot_data(peers.mail[b], chan, slot, rank);
        for (int i = tid; i < n; i += 256) dst[i] = payload[i];
        __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(slot_flag(peers.mail[b], chan, slot, rank), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (seq_ctr) seq_ctr[chan * P2P_MAXW + b] = seq;
        }
    }
    if (b != 0) return;
    // ---- wait for all ranks (own mailbox) ----
    float* mine = peers.mail[rank];
    if (tid < world) {
        const uint32_t* f = slot_flag(mine, chan, slot, tid);
        const long long t0 = wall_clock64();
        while ((int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > budget_ticks) {   // a peer never posted
                bad = 1;
                __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
    __threadfence_system();
    if (bad) {
        const float qnan = __builtin_nanf("");
        const int total = REDUCE ? n : world * n;
        for (int i = tid; i < total; i += 256) out[i] = qnan;
        return;
    }
    if (REDUCE) {
        for (int i = tid; i < n; i += 256) {
            float s = 0.f;
            for (int r = 0; r < world; ++r) s += __builtin_nontemporal_load(slot_data(mine, chan, slot, r) + i);
            out[i] = s;
        }
    } else {
        for (int r = 0; r < world; ++r) {
            const float* src = slot_data(mine, chan, slot, r);
            for (int i = tid; i < n; i += 256) out[(size_t)r * n + i] = __builtin_nontemporal_load(src + i);
        }
    }
}

}  // namespace

extern "C" {

size_t mgn_p2p_mailbox_bytes(void) { return data_floats() * sizeof(float) + flag_words() * sizeof(uint32_t); }

int mgn_p2p_alloc(void** mailbox) {
    if (!mailbox) return MGN_EINVAL;
    void* p = nullptr;
    // fine-grained device memory: coherent for system-scope atomics / fences across the GPUs of the node
    if (hipExtMallocWithFlags(&p, mgn_p2p_mailbox_bytes(), hipDeviceMallocFinegrained) != hipSuccess || !p) {
        (void)hipGetLastError();
        return MGN_ENOTSUP;
    }
    if (hipMemset(p, 0, mgn_p2p_mailbox_bytes()) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(p);
        return MGN_ELAUNCH;
    }
    *mailbox = p;
    return MGN_OK;
}

int mgn_p2p_free(void* mailbox) { return (!mailbox || hipFree(mailbox) == hipSuccess) ? MGN_OK : MGN_EINVAL; }

int mgn_p2p_export(void* mailbox, void* handle64) {
    if (!mailbox || !handle64) return MGN_EINVAL;
    static_assert(sizeof(hipIpcMemHandle_t) == MGN_P2P_HANDLE_BYTES, "handle size");
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, mailbox) != hipSuccess) {
        (void)hipGetLastError();
        return MGN_ENOTSUP;
    }
    memcpy(handle64, &h, sizeof(h));
    return MGN_OK;
}

int mgn_p2p_open(const void* handle64, void** peer_mailbox) {
    if (!handle64 || !peer_mailbox) return MGN_EINVAL;
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !p) {
        (void)hipGetLastError();
        return MGN_ENOTSUP;
    }
    *peer_mailbox = p;
    return MGN_OK;
}

int mgn_p2p_close(void* peer_mailbox) { return (!peer_mailbox || hipIpcCloseMemHandle(peer_mailbox) == hipSuccess) ? MGN_OK : MGN_EINVAL; }

int mgn_p2p_exchange(void* const* mailboxes, int world, int rank, int channel, unsigned seq, unsigned* seq_counters, const float* payload, int n,
                     int reduce, float* out, int* status, float timeout_s, void* stream) {
    if (!mailboxes || world < 1 || world > P2P_MAXW || rank < 0 || rank >= world || channel < 0 || channel >= P2P_CHANNELS ||
        (seq == 0 && !seq_counters) || !payload || n < 1 || n > P2P_SLOT || !out || !status || !(timeout_s > 0.f))
        return MGN_EINVAL;
    Peers pr;
    for (int r = 0; r < P2P_MAXW; ++r) {
        pr.mail[r] = r < world ? (float*)mailboxes[r] : nullptr;
        if (r < world && !pr.mail[r]) return MGN_EINVAL;
    }
    const long long ticks = (long long)((double)timeout_s * 100.0e6);   // wall_clock64: constant 100 MHz
    if (reduce)
        hipLaunchKernelGGL(p2p_exchange<true>, dim3(world), dim3(256), 0, (hipStream_t)stream, pr, world, rank, channel, seq, seq_counters, payload, n, out, status, ticks);
    else
        hipLaunchKernelGGL(p2p_exchange<false>, dim3(world), dim3(256), 0, (hipStream_t)stream, pr, world, rank, channel, seq, seq_counters, payload, n, out, status, ticks);
    return hipGetLastError() == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

}  // extern "C"

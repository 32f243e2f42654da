// plan.hip -- recorder and replayer of a launch plan: one training step as a table of kernel launches with by-value arguments,
// event records / waits between its streams and break points where the host has something of its own to do.
//
// Replaces, per training step, the Python / autograd / ctypes issue of the step's ~700 launches (the loop tools/train_net.py:232-234
// delegates to detectron2's trainer: forward -> losses -> backward -> optimizer) by ONE pass over the table from C.  The
// recording is made by the product's own C-ABI calls while they execute a real step (csrc/mgn_launch.h writes every launch down); the
// cross-stream dependencies are NOT recorded from the host's stream calls but derived by engine/plan.py from the memory each launch
// reads and writes (pointer arguments resolved against the allocator's blocks), which is also what makes the reuse of freed blocks
// across streams safe on replay.  What a step needs from the host (learning-rate tables) lives in device tables that are refreshed
// before the replay, exactly as for the hipGraph path it supersedes (engine/trainer.py).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "mgnet_hip.h"

namespace mgn_plan {

extern "C" int g_mgn_plan_recording = 0;

struct Node {
    int type;   // 0 launch, 1 prof mark
    const void* func;
    dim3 grid, block;
    size_t shmem;
    hipStream_t stream;
    int which;             // prof mark: 0 begin, 1 end
    int blob_off, nbytes;  // into Plan::blob
    int arg_off, nargs;    // into Plan::args
};

struct Op {
    int type, a;        // MGN_PLAN_OP_*; a = node / event index
    hipStream_t stream; // record / wait
};

struct Plan {
    std::vector<Node> nodes;
    std::vector<unsigned char> blob;
    std::vector<ArgDesc> args;
    std::vector<Op> ops;
    std::vector<hipEvent_t> events;
    std::vector<void*> argv;   // per launch op: pointers into blob (filled by compile)
    std::vector<int> argv_off; // node -> offset into argv
    std::vector<hipEvent_t> prof[2];
    // in-situ kernel trace of a replay (mgn_plan_trace): per launch node a hipEvent pair bound to the dispatch itself
    // (hipExtLaunchKernel: the events take the dispatch's own begin / end timestamps, no marker packets enter the queues)
    std::vector<hipEvent_t> tr[2];
    bool trace = false;
    std::vector<unsigned char> skip;   // what-if replays (mgn_plan_set_skip): node not launched
    // race hunting (mgn_plan_set_jitter): random delay kernels in front of launches shake the relative timing of the streams -- a replay
    // whose cross-stream edges are complete gives the same bits under any timing
    // data probes (mgn_plan_probe): after node i, a checksum of a memory range is written to a caller-owned slot -- two replays of
    // the same step can then be compared range by range to find the first buffer that differs
    struct Probe { int node; const void* ptr; size_t nwords; unsigned long long* out; int copy; };
    std::vector<Probe> probes;
    std::vector<unsigned char> fence_after;   // debugging (mgn_plan_set_skip mode 3): an event record + self-wait behind the node
    hipEvent_t fence_ev = nullptr;
    unsigned long long jitter_state = 0;
    int jitter_permille = 0, jitter_max_us = 0;
    bool overflow = false;
};

// order-independent checksum (64-bit integer sums: exact whatever the block order) ADDED to *out -- the caller zeroes the slot
__global__ void plan_checksum_kernel(const unsigned* __restrict__ p, size_t nwords, unsigned long long* out) {
    __shared__ unsigned long long sh[256];
    unsigned long long a = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) a += (unsigned long long)p[i] * (2 * (i % 8191) + 1);
    sh[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int k = 0; k < 256; ++k) t += sh[k];
        atomicAdd(out, t);
    }
}

__global__ void plan_delay_kernel(long long ticks) {   // wall_clock64: 100 MHz on gfx950
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

static Plan* g_rec = nullptr;
static std::mutex g_rec_mutex;   // launches are recorded from the caller's thread AND from the autograd thread of the backward pass

void record_launch(const void* func, dim3 grid, dim3 block, size_t shmem, hipStream_t stream, const unsigned char* blob, int nbytes,
                   const ArgDesc* args, int nargs) {
    std::lock_guard<std::mutex> lock(g_rec_mutex);
    Plan* p = g_rec;
    if (!p) return;
    if (nargs > MAX_ARGS || nbytes > MAX_ARG_BYTES) { p->overflow = true; return; }
    Node n{};
    n.type = 0; n.func = func; n.grid = grid; n.block = block; n.shmem = shmem; n.stream = stream;
    n.blob_off = (int)((p->blob.size() + 15) / 16 * 16);
    n.nbytes = nbytes;
    p->blob.resize((size_t)n.blob_off + (size_t)nbytes);
    memcpy(p->blob.data() + n.blob_off, blob, (size_t)nbytes);
    n.arg_off = (int)p->args.size(); n.nargs = nargs;
    p->args.insert(p->args.end(), args, args + nargs);
    p->nodes.push_back(n);
}

void record_prof_mark(int which, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_rec_mutex);
    Plan* p = g_rec;
    if (!p) return;
    Node n{};
    n.type = 1; n.which = which; n.stream = stream;
    p->nodes.push_back(n);
}

}  // namespace mgn_plan

using namespace mgn_plan;

extern "C" {

int mgn_plan_begin(void) {
    std::lock_guard<std::mutex> lock(g_rec_mutex);
    if (g_rec) return MGN_EINVAL;   // one recording at a time
    g_rec = new Plan();
    g_mgn_plan_recording = 1;
    return MGN_OK;
}

int mgn_plan_recorded(void) { return g_rec ? (int)g_rec->nodes.size() : -1; }

/* the plan being recorded (for mgn_plan_node_info / _args while it grows), or NULL */
const void* mgn_plan_current(void) { return g_rec; }

int mgn_plan_end(void** plan) {
    std::lock_guard<std::mutex> lock(g_rec_mutex);
    if (!g_rec || !plan) return MGN_EINVAL;
    g_mgn_plan_recording = 0;
    Plan* p = g_rec;
    g_rec = nullptr;
    if (p->overflow) { delete p; return MGN_ENOSPC; }
    *plan = p;
    return MGN_OK;
}

int mgn_plan_abort(void) {
    std::lock_guard<std::mutex> lock(g_rec_mutex);
    g_mgn_plan_recording = 0;
    delete g_rec;
    g_rec = nullptr;
    return MGN_OK;
}

int mgn_plan_node_count(const void* plan) { return plan ? (int)((const Plan*)plan)->nodes.size() : -1; }

int mgn_plan_node_info(const void* plan, int i, mgn_plan_node_info_t* out) {
    const Plan* p = (const Plan*)plan;
    if (!p || !out || i < 0 || i >= (int)p->nodes.size()) return MGN_EINVAL;
    const Node& n = p->nodes[i];
    out->type = n.type; out->stream = (void*)n.stream; out->func = n.func; out->nargs = n.nargs; out->nbytes = n.nbytes;
    out->grid[0] = n.grid.x; out->grid[1] = n.grid.y; out->grid[2] = n.grid.z;
    out->block[0] = n.block.x; out->block[1] = n.block.y; out->block[2] = n.block.z;
    out->shmem = n.shmem; out->which = n.which;
    out->blob = n.type == 0 ? p->blob.data() + n.blob_off : nullptr;
    out->name = n.type == 0 ? hipKernelNameRefByPtr(n.func, n.stream) : (n.which ? "prof_end" : "prof_begin");
    return MGN_OK;
}

/* per argument of node i: offset into the blob, size, kind (0 opaque, 1 pointer to const, 2 pointer) */
int mgn_plan_node_args(const void* plan, int i, int max_args, int* offsets, int* sizes, int* kinds) {
    const Plan* p = (const Plan*)plan;
    if (!p || i < 0 || i >= (int)p->nodes.size() || !offsets || !sizes || !kinds) return MGN_EINVAL;
    const Node& n = p->nodes[i];
    if (n.nargs > max_args) return MGN_ENOSPC;
    for (int k = 0; k < n.nargs; ++k) {
        const ArgDesc& d = p->args[n.arg_off + k];
        offsets[k] = d.offset; sizes[k] = d.size; kinds[k] = d.kind;
    }
    return n.nargs;
}

/* per argument of node i: the read-only pointer words of a by-value struct (two 64-bit masks per argument: bit w = the struct's
 * 8-byte word w is a pointer the kernel only reads through; csrc/mgn_launch.h MGN_PLAN_RO) */
int mgn_plan_node_ro(const void* plan, int i, int max_args, unsigned long long* ro) {
    const Plan* p = (const Plan*)plan;
    if (!p || i < 0 || i >= (int)p->nodes.size() || !ro) return MGN_EINVAL;
    const Node& n = p->nodes[i];
    if (n.nargs > max_args) return MGN_ENOSPC;
    for (int k = 0; k < n.nargs; ++k) {
        ro[2 * k] = p->args[n.arg_off + k].ro.w[0];
        ro[2 * k + 1] = p->args[n.arg_off + k].ro.w[1];
    }
    return n.nargs;
}

/* per argument of node i: the family of its struct's read-only declaration (csrc/mgn_launch.h: 0 none, MGN_RO_FAMILY_CONV 1 = argument
 * structs of the convolution kernels, MGN_RO_FAMILY_OTHER 2): which declarations a replay honours is decided per family */
int mgn_plan_node_ro_family(const void* plan, int i, int max_args, int* family) {
    const Plan* p = (const Plan*)plan;
    if (!p || i < 0 || i >= (int)p->nodes.size() || !family) return MGN_EINVAL;
    const Node& n = p->nodes[i];
    if (n.nargs > max_args) return MGN_ENOSPC;
    for (int k = 0; k < n.nargs; ++k) family[k] = p->args[n.arg_off + k].family;
    return n.nargs;
}

/* the replay schedule: ops[k] = (type, a, stream).  LAUNCH a = node; RECORD / WAIT a = event index (n_events are created here);
 * BREAK = return to the host (mgn_plan_run stops in front of it).  prof_slots > 0 creates that many hipEvent pairs for the prof marks. */
int mgn_plan_compile(void* plan, int n_ops, const int* types, const int* a, void* const* streams, int n_events, int prof_slots) {
    Plan* p = (Plan*)plan;
    if (!p || n_ops < 0 || (n_ops && (!types || !a || !streams)) || n_events < 0 || prof_slots < 0) return MGN_EINVAL;
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    for (int w = 0; w < 2; ++w) { for (hipEvent_t e : p->prof[w]) (void)hipEventDestroy(e); p->prof[w].clear(); }
    p->events.clear(); p->ops.clear();
    for (int k = 0; k < n_events; ++k) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return MGN_ELAUNCH;
        p->events.push_back(e);
    }
    for (int w = 0; w < 2; ++w)
        for (int k = 0; k < prof_slots; ++k) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return MGN_ELAUNCH;
            p->prof[w].push_back(e);
        }
    // argument pointer vectors of the launches
    p->argv.clear(); p->argv_off.assign(p->nodes.size(), 0);
    for (size_t i = 0; i < p->nodes.size(); ++i) {
        const Node& n = p->nodes[i];
        p->argv_off[i] = (int)p->argv.size();
        for (int k = 0; k < n.nargs; ++k) p->argv.push_back(p->blob.data() + n.blob_off + p->args[n.arg_off + k].offset);
    }
    for (int k = 0; k < n_ops; ++k) {
        const int t = types[k];
        if (t == MGN_PLAN_OP_LAUNCH) { if (a[k] < 0 || a[k] >= (int)p->nodes.size()) return MGN_EINVAL; }
        else if (t == MGN_PLAN_OP_RECORD || t == MGN_PLAN_OP_WAIT) { if (a[k] < 0 || a[k] >= n_events) return MGN_EINVAL; }
        else if (t != MGN_PLAN_OP_BREAK) return MGN_EINVAL;
        p->ops.push_back(Op{t, a[k], (hipStream_t)streams[k]});
    }
    return MGN_OK;
}

/* runs ops [from, ...) up to the next BREAK or the end; returns the index of the op it stopped in front of (== op count at the end),
 * or a negative error.  A BREAK at `from` itself is stepped over.  prof_slot: which event pair the prof marks of this pass record. */
int mgn_plan_run(void* plan, int from, int prof_slot) {
    Plan* p = (Plan*)plan;
    if (!p || from < 0 || from > (int)p->ops.size()) return MGN_EINVAL;
    const int n = (int)p->ops.size();
    int k = from;
    if (k < n && p->ops[k].type == MGN_PLAN_OP_BREAK) ++k;
    for (; k < n; ++k) {
        const Op& o = p->ops[k];
        if (o.type == MGN_PLAN_OP_LAUNCH) {
            const Node& nd = p->nodes[o.a];
            if (nd.type == 0) {
                if (!p->skip.empty() && p->skip[o.a] == 1) continue;
                if (p->jitter_permille > 0) {
                    p->jitter_state = p->jitter_state * 6364136223846793005ull + 1442695040888963407ull;
                    const unsigned r = (unsigned)(p->jitter_state >> 33);
                    if ((int)(r % 1000u) < p->jitter_permille) {
                        const long long us = 1 + (long long)((r / 1000u) % (unsigned)p->jitter_max_us);
                        plan_delay_kernel<<<1, 64, 0, nd.stream>>>(us * 100);
                    }
                }
                if (!p->skip.empty() && p->skip[o.a] == 2 &&   // what-if "twice": the extra launch first, the traced / ordinary one after it
                    hipLaunchKernel(nd.func, nd.grid, nd.block, p->argv.data() + p->argv_off[o.a], nd.shmem, nd.stream) != hipSuccess) {
                    (void)hipGetLastError();
                    return MGN_ELAUNCH;
                }
                if (p->trace) {
                    if (hipExtLaunchKernel(nd.func, nd.grid, nd.block, p->argv.data() + p->argv_off[o.a], nd.shmem, nd.stream,
                                           p->tr[0][o.a], p->tr[1][o.a], 0) != hipSuccess) {
                        (void)hipGetLastError();
                        return MGN_ELAUNCH;
                    }
                } else if (hipLaunchKernel(nd.func, nd.grid, nd.block, p->argv.data() + p->argv_off[o.a], nd.shmem, nd.stream) != hipSuccess) {
                    (void)hipGetLastError();
                    return MGN_ELAUNCH;
                }
                if (!p->fence_after.empty() && p->fence_after[o.a]) {
                    if (!p->fence_ev) (void)hipEventCreateWithFlags(&p->fence_ev, hipEventDisableTiming);
                    (void)hipEventRecord(p->fence_ev, nd.stream);
                    (void)hipStreamWaitEvent(nd.stream, p->fence_ev, 0);
                }
                for (const Plan::Probe& pr : p->probes)
                    if (pr.node == o.a) {
                        if (pr.copy) (void)hipMemcpyAsync(pr.out, pr.ptr, pr.nwords * 4, hipMemcpyDeviceToDevice, nd.stream);
                        else plan_checksum_kernel<<<(unsigned)(pr.nwords / 4096 < 1 ? 1 : (pr.nwords / 4096 > 1024 ? 1024 : pr.nwords / 4096)), 256, 0, nd.stream>>>(
                                (const unsigned*)pr.ptr, pr.nwords, pr.out);
                    }
            } else if (prof_slot >= 0 && prof_slot < (int)p->prof[nd.which].size()) {
                (void)hipEventRecord(p->prof[nd.which][prof_slot], nd.stream);
            }
        } else if (o.type == MGN_PLAN_OP_RECORD) {
            if (hipEventRecord(p->events[o.a], o.stream) != hipSuccess) return MGN_ELAUNCH;
        } else if (o.type == MGN_PLAN_OP_WAIT) {
            if (hipStreamWaitEvent(o.stream, p->events[o.a], 0) != hipSuccess) return MGN_ELAUNCH;
        } else {
            return k;   // BREAK
        }
    }
    return n;
}

/* replay node i on another stream than it was recorded on (the caller's schedule must order it accordingly) */
int mgn_plan_set_stream(void* plan, int i, void* stream) {
    Plan* p = (Plan*)plan;
    if (!p || i < 0 || i >= (int)p->nodes.size()) return MGN_EINVAL;
    p->nodes[i].stream = (hipStream_t)stream;
    return MGN_OK;
}

/* In-situ kernel trace: with on != 0 every launch of the following replays carries a hipEvent pair bound to the dispatch
 * (hipExtLaunchKernel), so that one un-profiled replay yields begin / end of each of its kernels on the device clock. */
int mgn_plan_trace(void* plan, int on) {
    Plan* p = (Plan*)plan;
    if (!p) return MGN_EINVAL;
    if (on && p->tr[0].empty()) {
        for (int w = 0; w < 2; ++w) {
            p->tr[w].assign(p->nodes.size(), nullptr);
            for (size_t i = 0; i < p->nodes.size(); ++i)
                if (p->nodes[i].type == 0 && hipEventCreate(&p->tr[w][i]) != hipSuccess) return MGN_ELAUNCH;
        }
    }
    p->trace = on != 0;
    return MGN_OK;
}

/* After a traced replay has completed: for every node i (n = node count) times in ms relative to the BEGIN of node `ref`:
 * t_a[i] = elapsed(begin event of ref, begin event of i), t_b[i] = elapsed(begin event of ref, end event of i),
 * dur[i] = elapsed(begin event of i, end event of i); NaN for marks, skipped and never-launched nodes.  (For events bound to
 * dispatches the runtime evaluates elapsed(x, y) as END(y's dispatch) - BEGIN(x's dispatch): t_a == t_b and begin = t_b - dur;
 * the caller checks which of the two conventions the installed runtime follows.) */
int mgn_plan_trace_read(void* plan, int ref, int n, float* t_a, float* t_b, float* dur) {
    Plan* p = (Plan*)plan;
    if (!p || !t_a || !t_b || !dur || n != (int)p->nodes.size() || p->tr[0].empty() || ref < 0 || ref >= n || !p->tr[0][ref])
        return MGN_EINVAL;
    const float nan = __builtin_nanf("");
    for (int i = 0; i < n; ++i) {
        t_a[i] = t_b[i] = dur[i] = nan;
        if (!p->tr[0][i] || (!p->skip.empty() && p->skip[i] == 1)) continue;
        float a, b, d;
        if (hipEventElapsedTime(&a, p->tr[0][ref], p->tr[0][i]) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (hipEventElapsedTime(&b, p->tr[0][ref], p->tr[1][i]) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (hipEventElapsedTime(&d, p->tr[0][i], p->tr[1][i]) != hipSuccess) { (void)hipGetLastError(); continue; }
        t_a[i] = a; t_b[i] = b; dur[i] = d;
    }
    return MGN_OK;
}

/* What-if replays.  skip = 1: node i is not launched (its outputs keep whatever they held -- the step's RESULTS are then
 * meaningless, its timing is that of the step without the node); skip = 2: node i is launched TWICE back to back (what the node
 * costs, measured as an increase, with the step's data left intact wherever the kernel is a pure function of its inputs);
 * skip = 0: as recorded. */
int mgn_plan_set_skip(void* plan, int i, int skip) {
    Plan* p = (Plan*)plan;
    if (!p || i < 0 || i >= (int)p->nodes.size() || skip < 0 || skip > 3) return MGN_EINVAL;
    if (skip == 3) {   // (debugging: an event record + wait on the node's own stream behind it: a barrier packet with fences)
        if (p->fence_after.empty()) p->fence_after.assign(p->nodes.size(), 0);
        p->fence_after[i] = 1;
        return MGN_OK;
    }
    if (p->skip.empty()) p->skip.assign(p->nodes.size(), 0);
    p->skip[i] = (unsigned char)skip;
    return MGN_OK;
}

/* Race hunting: from now on every launch of a replay is preceded, with probability permille / 1000, by a kernel that idles its stream for
 * 1 .. max_us microseconds (pseudo-random from `seed`).  permille = 0 switches it off. */
int mgn_plan_set_jitter(void* plan, unsigned long long seed, int permille, int max_us) {
    Plan* p = (Plan*)plan;
    if (!p || permille < 0 || permille > 1000 || (permille > 0 && max_us < 1)) return MGN_EINVAL;
    p->jitter_state = seed * 2654435761ull + 1;
    p->jitter_permille = permille;
    p->jitter_max_us = max_us;
    return MGN_OK;
}

/* debugging: after node i of every following replay an order-independent checksum of [ptr, ptr + nbytes) (device memory, nbytes % 4 == 0)
 * is ADDED to *out (device memory, 8 bytes, zeroed by the caller between replays) on the node's stream; node < 0 clears all probes */
int mgn_plan_probe(void* plan, int node, const void* ptr, size_t nbytes, void* out) {
    Plan* p = (Plan*)plan;
    if (!p) return MGN_EINVAL;
    if (node < 0) { p->probes.clear(); return MGN_OK; }
    const bool copy = node >= (1 << 24);   // node | 1 << 24: copy the range to `out` (nbytes of device memory) instead of a checksum
    node &= (1 << 24) - 1;
    if (node >= (int)p->nodes.size() || !ptr || !out || nbytes % 4) return MGN_EINVAL;
    p->probes.push_back(Plan::Probe{node, ptr, nbytes / 4, (unsigned long long*)out, copy ? 1 : 0});
    return MGN_OK;
}

int mgn_plan_prof_elapsed(void* plan, int slot, float* ms) {
    Plan* p = (Plan*)plan;
    if (!p || !ms || slot < 0 || slot >= (int)p->prof[0].size()) return MGN_EINVAL;
    return hipEventElapsedTime(ms, p->prof[0][slot], p->prof[1][slot]) == hipSuccess ? MGN_OK : MGN_ELAUNCH;
}

int mgn_plan_free(void* plan) {
    Plan* p = (Plan*)plan;
    if (!p) return MGN_OK;
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    for (int w = 0; w < 2; ++w) {
        for (hipEvent_t e : p->prof[w]) (void)hipEventDestroy(e);
        for (hipEvent_t e : p->tr[w]) if (e) (void)hipEventDestroy(e);
    }
    delete p;
    return MGN_OK;
}

}  // extern "C"

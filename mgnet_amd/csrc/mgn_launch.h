// mgn_launch.h -- every kernel launch of libmgnet_hip.so goes through mgn_plan::launch (this header is force-included into each
// csrc/*.hip by mgnet_amd/build.py and redefines hipLaunchKernelGGL).  Outside a recording it IS the ordinary launch.  While
// mgn_plan_begin() .. mgn_plan_end() is active (csrc/plan.hip, engine/plan.py) every launch is ALSO written down -- kernel, grid,
// block, dynamic LDS, stream and a by-value copy of its arguments with their types classified (pointer to const = read, pointer =
// written, anything else = opaque bytes that are scanned for pointers) -- so that a whole training step, recorded once, can be
// replayed from C without the Python / autograd / ctypes work of issuing its ~700 launches (tools/train_net.py:232-234 delegates the
// same loop to detectron2's trainer).  hipGraph cannot do this on this ROCm: hipGraphLaunch of the step's graph costs
// more host time than the eager issue and capturing the side-stream branches crashes.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstring>
#include <tuple>
#include <type_traits>
#include <utility>

namespace mgn_plan {

constexpr int MAX_ARGS = 48;
constexpr int MAX_ARG_BYTES = 1024;

// which 8-byte words of a by-value struct argument hold pointers the kernel only READS through (structs up to 1 KiB).  A struct
// says so with MGN_PLAN_RO right behind its definition; without it every pointer found in the struct counts as written -- which
// orders, e.g., the three decoders' first convolutions (all readers of one backbone feature map) one after the other on replay.
struct RoBits {
    unsigned long long w[2] = {0, 0};
    constexpr RoBits operator|(RoBits o) const { RoBits r; r.w[0] = w[0] | o.w[0]; r.w[1] = w[1] | o.w[1]; return r; }
};
constexpr RoBits ro_field(size_t offset, size_t size) {
    RoBits r;
    for (size_t b = offset / 8; b < (offset + size + 7) / 8 && b < 128; ++b) r.w[b / 64] |= 1ull << (b % 64);
    return r;
}
constexpr RoBits mgn_plan_ro(const void*) { return RoBits{}; }   // (argument types that declare nothing)
constexpr int mgn_plan_ro_family(const void*) { return 0; }
// A declaration belongs to a FAMILY, fixed here in the source next to the struct -- not derived from kernel names at run time:
//   MGN_RO_FAMILY_CONV  (1)  argument structs of the convolution kernels (forward, data gradient, weight gradient, stems): the
//                            declarations the replay honours by default (engine/plan.py MGN_PLAN_RO=conv)
//   MGN_RO_FAMILY_OTHER (2)  every other declared struct: honoured with MGN_PLAN_RO=all only (profiles/r05_plan_determinism.txt)
// A new kernel joins the default set only by declaring its struct with MGN_PLAN_RO_CONV.
#define MGN_RO_FAMILY_CONV 1
#define MGN_RO_FAMILY_OTHER 2
#define MGN_RO(f) | mgn_plan::ro_field(offsetof(S_, f), sizeof(S_::f))
#define MGN_PLAN_RO_IN(T, family, fields) \
    constexpr mgn_plan::RoBits mgn_plan_ro(const T*) { using S_ = T; return mgn_plan::RoBits{} fields; } \
    constexpr int mgn_plan_ro_family(const T*) { return family; }
#define MGN_PLAN_RO(T, fields) MGN_PLAN_RO_IN(T, MGN_RO_FAMILY_OTHER, fields)
#define MGN_PLAN_RO_CONV(T, fields) MGN_PLAN_RO_IN(T, MGN_RO_FAMILY_CONV, fields)

struct ArgDesc {
    unsigned short offset, size;
    unsigned char kind;   // 0 opaque bytes, 1 pointer to const (read), 2 pointer (read / written)
    unsigned char family; // kind 0: MGN_RO_FAMILY_* of the struct's declaration, 0 = none
    RoBits ro;            // kind 0: the struct's read-only pointer words
};

extern "C" int g_mgn_plan_recording;   // csrc/plan.hip
void record_launch(const void* func, dim3 grid, dim3 block, size_t shmem, hipStream_t stream, const unsigned char* blob, int nbytes,
                   const ArgDesc* args, int nargs);
void record_prof_mark(int which, hipStream_t stream);

template <class T>
constexpr unsigned char arg_kind() {
    if constexpr (std::is_pointer<T>::value) return std::is_const<typename std::remove_pointer<T>::type>::value ? 1 : 2;
    else return 0;
}

template <class... P>
struct Packer {
    alignas(16) unsigned char blob[MAX_ARG_BYTES];
    ArgDesc desc[MAX_ARGS];
    int n = 0, bytes = 0;
    template <class T>
    void put(const T& v) {
        static_assert(std::is_trivially_copyable<T>::value, "kernel arguments are plain data");
        constexpr int al = alignof(T) > 16 ? 16 : (int)alignof(T);
        bytes = (bytes + al - 1) / al * al;
        if (n < MAX_ARGS && bytes + (int)sizeof(T) <= MAX_ARG_BYTES) {
            std::memcpy(blob + bytes, &v, sizeof(T));
            RoBits ro;
            int family = 0;
            if constexpr (!std::is_pointer<T>::value && std::is_class<T>::value) {
                ro = mgn_plan_ro(static_cast<const T*>(nullptr));
                family = mgn_plan_ro_family(static_cast<const T*>(nullptr));
            }
            desc[n] = ArgDesc{(unsigned short)bytes, (unsigned short)sizeof(T), arg_kind<T>(), (unsigned char)family, ro};
        }
        bytes += (int)sizeof(T);
        ++n;
    }
};

// argument I of the kernel: the caller's I-th value converted to the parameter type, or (trailing parameters the call site leaves to
// their declared default -- every default in csrc/ is a null pointer) a value-initialised one
template <size_t I, class T, class Tup>
inline T pick_arg(Tup& t) {
    if constexpr (I < std::tuple_size<Tup>::value) return static_cast<T>(std::get<I>(t));
    else { static_assert(std::is_pointer<T>::value, "only null-pointer defaults are supported"); return T{}; }
}

template <class... P, size_t... I, class Tup>
inline void launch_values(void (*kernel)(P...), dim3 grid, dim3 block, size_t shmem, hipStream_t stream, std::index_sequence<I...>, Tup t) {
    std::tuple<P...> v{pick_arg<I, P>(t)...};
    if (__builtin_expect(g_mgn_plan_recording, 0)) {
        Packer<P...> pk;
        (pk.template put<P>(std::get<I>(v)), ...);
        record_launch(reinterpret_cast<const void*>(kernel), grid, block, shmem, stream, pk.blob, pk.bytes, pk.desc, pk.n);
    }
    kernel<<<grid, block, shmem, stream>>>(std::get<I>(v)...);
}

template <class... P, class... A>
inline void launch(void (*kernel)(P...), dim3 grid, dim3 block, size_t shmem, hipStream_t stream, A&&... a) {
    static_assert(sizeof...(A) <= sizeof...(P), "too many kernel arguments");
    launch_values(kernel, grid, block, shmem, stream, std::index_sequence_for<P...>{}, std::forward_as_tuple(a...));
}

// hipEvent pair around a kernel (bench.py's roofline leg): recorded as a plan node too, so that a replayed step times the same kernel
inline void prof_mark(int which, void* event, hipStream_t stream) {
    if (__builtin_expect(g_mgn_plan_recording, 0)) record_prof_mark(which, stream);
    if (event) (void)hipEventRecord((hipEvent_t)event, stream);
}

}  // namespace mgn_plan

#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) mgn_plan::launch(kernel, grid, block, shmem, stream, ##__VA_ARGS__)

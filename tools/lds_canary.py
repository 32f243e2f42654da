"""LDS canary: does any kernel of the library write LDS outside its own allocation?  A canary kernel (compiled here with hipcc, loaded through
ctypes) fills its block's LDS with a pattern and keeps re-checking it for a few hundred microseconds while, on another stream, one
kernel family of the training step after the other runs at the C4 shapes (tools/race_screen.py's cases).  A canary block that shares a
CU with a block writing out of bounds sees its pattern change.  Canary LDS per block is small enough to co-reside with everything
(8 / 16 KB variants).  Usage: lds_canary.py [rounds]"""
import ctypes, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import race_screen as rs

SRC = r'''
#include <hip/hip_runtime.h>
extern "C" __global__ void canary(unsigned* bad, int words, long long ticks, unsigned seed) {
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
extern "C" int launch_canary(unsigned* bad, int blocks, int lds_bytes, long long ticks, unsigned seed, void* stream) {
    hipLaunchKernelGGL(canary, dim3(blocks), dim3(64), lds_bytes, (hipStream_t)stream, bad, lds_bytes / 4, ticks, seed);
    return (int)hipGetLastError();
}
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "canary.hip"), "w").write(SRC)
so = os.path.join(ROOT, "tools", "probe", "libcanary.so")
if not os.path.exists(so) or os.environ.get("REBUILD"):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(d, "canary.hip"), "-o", so], check=True)
if not torch.cuda.is_available():
    print("built", so); sys.exit(0)
lib = ctypes.CDLL(so)
lib.launch_canary.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_uint, ctypes.c_void_p]
dev = torch.device("cuda:0")
bad = torch.zeros(1, dtype=torch.int32, device=dev)
cs = torch.cuda.Stream()
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
found = []
for name, fn, _ref in rs.cases(8):
    fn(); torch.cuda.synchronize()
    bad.zero_()
    for r in range(ROUNDS):
        for lds_bytes in (8192, 16384):
            # 2048 one-wave blocks: up to 8 per CU, resident for ~400 us (wall_clock64: 100 MHz)
            assert lib.launch_canary(bad.data_ptr(), 2048, lds_bytes, 40000, 1234 + r, ctypes.c_void_p(cs.cuda_stream)) == 0
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
    n = int(bad.item())
    print(f"{name:62s} {'ok' if n == 0 else f'{n} canary blocks saw their LDS change'}", flush=True)
    if n:
        found.append((name, n))
print("kernels next to which a canary's LDS changed:", found)

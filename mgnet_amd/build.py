"""Build libmgnet_hip.so (all csrc/*.hip, gfx950) in-tree with hipcc.  `python -m mgnet_amd.build [--force]`.

Every source is compiled to its own object (in parallel, re-done only when the source or a header is newer) and the
objects are linked into one shared library; the kernels of different files never reference each other."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "lib", "libmgnet_hip.so")
OBJ = os.path.join(HERE, "lib", "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
          "-I" + os.path.join(HERE, "csrc"),
          "-include", os.path.join(HERE, "csrc", "mgn_launch.h")]   # every launch through mgn_plan::launch (launch-plan recording)


def sources():
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))


def headers():
    return glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(HERE, "csrc", "*.h"))


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


def is_stale():
    return _stale(LIB, sources() + headers())


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hdr = headers()
    base = lambda s: [s[:-8] + ".hip"] if s.endswith("_f16.hip") else []   # x_f16.hip = `#define MGN_F16` + `#include "x.hip"`
    todo = [s for s in sources() if force or _stale(_obj(s), [s] + base(s) + hdr)]

    def compile_one(src):
        # per-file flags: a line `// hipcc-flags: <flags>` in the first 40 lines of the source
        extra = []
        with open(src) as f:
            for _, line in zip(range(40), f):
                if line.startswith("// hipcc-flags:"):
                    extra += line.split(":", 1)[1].split()
        cmd = [HIPCC] + CFLAGS + extra + ["-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(todo)))) as ex:
        list(ex.map(compile_one, todo))
    for o in glob.glob(os.path.join(OBJ, "*.o")):          # objects of deleted sources
        if o not in {_obj(s) for s in sources()}:
            os.remove(o)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in sources()] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

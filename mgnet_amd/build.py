"""Build libmgnet_hip.so (all csrc/*.hip, gfx950) in-tree with hipcc.  `python -m mgnet_amd.build [--force]`."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "lib", "libmgnet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include")]


def sources():
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))


def is_stale():
    if not os.path.exists(LIB):
        return True
    deps = sources() + glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(HERE, "csrc", "*.h"))
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [HIPCC] + FLAGS + sources() + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

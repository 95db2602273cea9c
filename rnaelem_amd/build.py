"""Builds rnaelem_amd/libelemdp.so (HIP kernels for gfx950 + host engine + C ABI) in-tree.

    python -m rnaelem_amd.build        # or  __graft_entry__.build()

hipcc cross-compiles without a GPU.  The library has no CPU implementation of the DP.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libelemdp.so")
SOURCES = ["kernels.hip", "train_kernels.hip", "lin_kernels.hip", "bpp_kernels.hip", "engine.cpp", "automaton.cpp", "energy_tables.cpp"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function", "-Wno-unused-variable"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", "elemdp.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    extra = os.environ.get("ELEMDP_CXXFLAGS", "").split()
    cmd = [HIPCC] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

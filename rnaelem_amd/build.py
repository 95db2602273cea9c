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


# per-file flags.  lin_kernels.hip: machine LICM hoists the constants of exp() (two dozen vector registers) out of the inner loops
# of the band kernels to the top of their block loop, where they stay live across every phase -- a workgroup per CU of k4_out.
EXTRA = {"lin_kernels.hip": ["-mllvm", "-disable-machine-licm"]}
OBJ_DIR = os.path.join(HERE, "_build")


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    extra = os.environ.get("ELEMDP_CXXFLAGS", "").split()
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = [f for f in FLAGS if f != "-shared"]

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, src + ".o")
        cmd = [HIPCC] + flags + EXTRA.get(src, []) + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

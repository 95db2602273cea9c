"""Builds the test-only CPU emulation library (see emul.cpp).  Not part of the product."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, "rnaelem_amd", "csrc")
LIB = os.path.join(HERE, "libelemdp_emul.so")
SRCS = [os.path.join(HERE, "emul.cpp"), os.path.join(CSRC, "automaton.cpp"), os.path.join(CSRC, "energy_tables.cpp")]
DEPS = SRCS + [os.path.join(CSRC, f) for f in ("dp_rules.h", "plan_rules.h", "energy_rules.h", "scan_rules.h",
                                                "device_layout.h", "automaton.h", "energy_tables.h", "host_prep.h", "lin_rules.h", "lin_params.h", "lin_fast.h")]


def build(force=False):
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS):
        return LIB
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-o", LIB] + SRCS)
    return LIB


if __name__ == "__main__":
    print(build(force=True))

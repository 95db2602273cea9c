"""Reads the gfx950 ISA of one kernel (hipcc -S --cuda-device-only) and prints what decided the last optimisations of round 1:

  * the sequence of vector-memory loads, waits and barriers (L<n> = n loads in a row, W<k> = s_waitcnt vmcnt(k), |B| = barrier,
    S = store / atomic, b = branch): a copy loop that waits for its loads before it stores shows up as L1 W0 L1 W0 ...;
  * instruction classes per barrier-delimited segment (VALU / f64 / SALU / LDS / VMEM) and the SGPR spill traffic
    (v_writelane / v_readlane).

    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -S --cuda-device-only \\
        rnaelem_amd/csrc/lin_kernels.hip -o /tmp/lk.s
    python tools/isa_report.py /tmp/lk.s k4_outILi0ELb1          # substring of the mangled kernel name
"""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r"^(_Z\w*%s\w*):" % re.escape(pat), txt, re.M)
if not m:
    raise SystemExit("no kernel matching %r" % pat)
name = m.group(1)
body = txt[m.end():txt.index("s_endpgm", m.end())]
ins = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", ".")) and not re.match(r"^\.?\w+:$", l.strip())]
print(name, len(ins), "instructions; v_readlane", sum(i.startswith("v_readlane") for i in ins), "v_writelane", sum(i.startswith("v_writelane") for i in ins))

seq, n = [], 0
for l in ins:
    if l.startswith(("global_load", "buffer_load", "flat_load")):
        n += 1
        continue
    tok = None
    if l.startswith("s_waitcnt") and "vmcnt" in l:
        tok = "W" + re.search(r"vmcnt\((\d+)\)", l).group(1)
    elif l.startswith("s_barrier"):
        tok = "|B|"
    elif l.startswith(("global_store", "global_atomic")):
        tok = "S"
    elif l.startswith("s_cbranch"):
        tok = "b"
    if tok:
        if n:
            seq.append("L%d" % n)
            n = 0
        if not (seq and tok in ("b", "S") and seq[-1] == tok):
            seq.append(tok)
print(" ".join(seq))


def cls(op):
    if re.match(r"v_(fma|mul|add|max|min|cmp\w*|div\w*|rcp|ldexp|frexp\w*|cvt)_f64|v_fmac_f64|v_cmp_\w+_f64", op):
        return "f64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


seg, cur = [], []
for l in ins:
    cur.append(l.split()[0])
    if l.startswith("s_barrier"):
        seg.append(cur)
        cur = []
seg.append(cur)
for k, sg in enumerate(seg):
    print("segment %d: %d" % (k, len(sg)), dict(Counter(cls(o) for o in sg)))

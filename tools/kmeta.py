"""Register / scratch / LDS metadata of the kernels of one .hip file (compiled to gfx950 assembly).  args: file.hip [filter [extra flags...]]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-S", "--cuda-device-only", src, "-o", "/tmp/kmeta.s"] + sys.argv[3:],
                      stderr=subprocess.DEVNULL)
t = open("/tmp/kmeta.s").read()
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", t, re.S):
    b = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    if flt not in name:
        continue
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, b).group(1)
    short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("elemdp::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-32s vgpr %3s vspill %3s sspill %3s scratch %3s" % (short, g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size")))

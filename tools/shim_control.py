"""Control for tests/test_round3_gpu.py: do the scan differences away from the motif come from the seventh digit of the trained
parameters?  Scans (reference binary, CPU) the model file the shim wrote and a model file written from the golden parameters."""
import os, subprocess, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rnaelem_amd import io
from tests.util import gload, gpath
from tests.golden.make_golden import parse_scan
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref")
t = gload("train_final.json")[1]
subprocess.run([os.path.join(R, "RNAelem_gpu"), "--fastq", gpath(t["fq"]), "--motif-pattern", t["pattern"], "--out1", "/tmp/s.model", "--out2", "/tmp/s.raw",
                "--max-iter", str(t["max_iter"]), "--no-shuffle", "--batch-size", "-1", "-t", "4", "--lambda-init", "0", "--epsilon", "1e-5"], capture_output=True)
m = io.read_model("/tmp/s.model")
print("max |x_shim - x_ref| as printed:", max(abs(a - b) for a, b in zip(m["x"], t["x"])))
io.write_model("/tmp/g.model", m, x=t["x"])
for name in ("s", "g"):
    subprocess.run([os.path.join(R, "RNAelem"), "scan", "--fastq", gpath(t["fq"]), "--motif-model", "/tmp/%s.model" % name, "--out1", "/tmp/%s2.raw" % name, "-t", "1"], capture_output=True)
a = sorted(parse_scan(open("/tmp/s2.raw").read()), key=lambda r: r["id"])
b = sorted(parse_scan(open("/tmp/g2.raw").read()), key=lambda r: r["id"])
c = sorted(parse_scan(open("/tmp/s.raw").read()), key=lambda r: r["id"])
g = sorted(t["records"], key=lambda r: r["id"])
print("files scanned by the reference binary: shim model vs golden-x model:", sum(x["rss"] != y["rss"] for x, y in zip(a, b)), "of", len(a), "rss differ")
print("shim in-process scan vs scan of its own model file:", sum(x["rss"] != y["rss"] for x, y in zip(c, a)), "differ")
print("golden in-process scan vs scan of golden-x model file:", sum(x["rss"] != y["rss"] for x, y in zip(g, b)), "differ")

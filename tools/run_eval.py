"""One load + a few train evaluations (target for rocprofv3).  args: n L reps pipeline group [option=value ...]"""
import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rnaelem_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
pipeline = int(sys.argv[4]) if len(sys.argv) > 4 else 4
group = int(sys.argv[5]) if len(sys.argv) > 5 else 0
eng = api.Engine(os.environ.get("ELEMDP_PATTERN", "((.*.))"), "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
eng.set_option("pipeline", pipeline)
if group:
    eng.set_option("group", group)
for kv in sys.argv[6:]:
    k, val = kv.split("=")
    eng.set_option(k, int(val))
seqs, quals = synth.synth_batch(n, L)
t0 = time.time()
eng.load_batch(seqs, quals)
print("load %.2fs" % (time.time() - t0))
x = eng.initial_params(1.0)
for _ in range(reps):
    fn, gr, eff, nsk = eng.train_eval(x)
    ms = eng.last_timing()
    print("pipeline %d: ms %s -> %.0f seq/s  fn %.12g |gr| %.9g nsk %d flagged %d" % (pipeline, ms[:2], n / ms[1] * 1e3, fn, abs(gr).sum(), nsk, ms[2]))

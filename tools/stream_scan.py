"""Streamed scan of a batch larger than what one load keeps resident (config E's shape).  args: n L [max_resident]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rnaelem_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
eng = api.Engine("(.....)", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
if len(sys.argv) > 3:
    eng.set_option("max_resident", int(sys.argv[3]))
seqs, quals = synth.synth_batch(n, L)
x = eng.initial_params(1.0)
t0 = time.time()
eng.load_batch(seqs, quals)
t1 = time.time()
recs, en = eng.scan(x)
t2 = time.time()
print("n=%d L=%d load %.2f s scan %.2f s -> %.0f seq/s with load; first record Ys=%d Ye=%d exist=%.4g" % (
    n, L, t1 - t0, t2 - t1, n / (t2 - t0), recs[0]["Ys"], recs[0]["Ye"], recs[0]["exist_prob"]), flush=True)

"""Scan throughput (config B / E shapes).  args: n L"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rnaelem_amd import api, synth
n, L = int(sys.argv[1]), int(sys.argv[2])
eng = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
seqs, quals = synth.synth_batch(n, L)
t0 = time.time(); eng.load_batch(seqs, quals); t1 = time.time()
x = eng.initial_params(1.0)
eng.scan(x)
t2 = time.time(); recs, en = eng.scan(x); t3 = time.time()
print("n=%d L=%d load %.2fs scan %.2fs -> %.0f seq/s ; first: Ys=%d Ye=%d exist=%.4g rss=%s" % (
    n, L, t1 - t0, t3 - t2, n / (t3 - t2), recs[0]["ys"] if "ys" in recs[0] else -1, recs[0].get("ye", -1), recs[0].get("exist_prob", 0), recs[0].get("rss", "")[:40]))

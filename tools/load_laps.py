"""Where does load_batch spend its time?  (ELEMDP_TIME=1 prints the laps)  args: n L pattern"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rnaelem_amd import api, synth
n, L = int(sys.argv[1]), int(sys.argv[2])
pat = sys.argv[3] if len(sys.argv) > 3 else "(.....)"
eng = api.Engine(pat, "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
for seed in (77 + L, None, 5):
    seqs, quals = synth.synth_batch(n, L, seed=seed)
    t0 = time.perf_counter()
    eng.load_batch(seqs, quals)
    print("== load_batch %d x %d: %.3f s" % (n, L, time.perf_counter() - t0), flush=True)

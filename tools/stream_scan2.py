"""Streamed scan twice (the second pass warm: buffers allocated); ELEMDP_TIME laps of the second pass only.  args: n L max_resident"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rnaelem_amd import api, synth
n, L, mr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
eng = api.Engine("(.....)", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
eng.set_option("max_resident", mr)
seqs, quals = synth.synth_batch(n, L)
x = eng.initial_params(1.0)
for rep in range(2):
    sys.stderr.write("==== pass %d\n" % rep); sys.stderr.flush()
    t0 = time.time()
    eng.load_batch(seqs, quals)
    t1 = time.time()
    recs, en = eng.scan(x)
    t2 = time.time()
    print("pass %d: n=%d L=%d load %.2f s scan %.2f s -> %.0f seq/s with load (device ms %s)" % (rep, n, L, t1 - t0, t2 - t1, n / (t2 - t0), eng.last_timing()[:2]), flush=True)

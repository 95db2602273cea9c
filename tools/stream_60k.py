"""Streaming train evaluation of a training set that does not fit the device at once: n x L synthetic RNAs,
pattern ((.*.)); the handle decides by itself to stream (or option max_resident forces a chunk size).  args: n L [max_resident]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from rnaelem_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
eng = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
if len(sys.argv) > 3:
    eng.set_option("max_resident", int(sys.argv[3]))
seqs, quals = synth.synth_batch(n, L)
t0 = time.time()
eng.load_batch(seqs, quals)
print("load_batch %.2f s" % (time.time() - t0), flush=True)
x = eng.initial_params(1.0)
for rep in range(int(os.environ.get("STREAM_REPS", "2"))):
    t0 = time.time()
    fn, gr, eff, nsk = eng.train_eval(x)
    dt = time.time() - t0
    print("eval %d: %.2f s -> %.0f seq/s  fn %.10g |gr| %.8g eff %.6g skipped %d  (kernel ms %s)" % (rep, dt, n / dt, fn, np.abs(gr).sum(), eff, nsk, eng.last_timing()[:2]), flush=True)
st = eng.seq_stats()
print("f sum %.10g  (== fn: %s)" % (st[:, 3].sum(), abs(st[:, 3].sum() - fn) < 1e-8 * abs(fn)))

"""Scan throughput (config B / E shapes).  args: n L [pattern [dbg [group]]]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rnaelem_amd import api, synth
n, L = int(sys.argv[1]), int(sys.argv[2])
pattern = sys.argv[3] if len(sys.argv) > 3 else "((.*.))"
eng = api.Engine(pattern, "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
if len(sys.argv) > 5:
    eng.set_option("group", int(sys.argv[5]))
if len(sys.argv) > 6:
    eng.set_option("group_streams", int(sys.argv[6]))
seqs, quals = synth.synth_batch(n, L)
t0 = time.time(); eng.load_batch(seqs, quals); t1 = time.time()
x = eng.initial_params(1.0)
tf = time.time(); eng.scan(x); tf = time.time() - tf
t2 = time.time(); recs, en = eng.scan(x); t3 = time.time()
if len(sys.argv) > 4:
    eng.set_option("dbg", int(sys.argv[4]))
    t2 = time.time(); eng.scan(x); t3 = time.time()
ms = eng.last_timing()
print("pattern %s S=%d n=%d L=%d load %.2fs first scan %.2fs, scan %.2fs (device %.0f ms, %d sequences re-run in log space) -> %.0f seq/s" % (
    pattern, eng.n_state, n, L, t1 - t0, tf, t3 - t2, ms[1], ms[2], n / (t3 - t2)))

"""One load + two train evaluations with option dbg (phases switched off: timing / counter experiments).  args: n L dbg"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rnaelem_amd import api, synth
n, L, dbg = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
eng = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
seqs, quals = synth.synth_batch(n, L)
eng.load_batch(seqs, quals)
x = eng.initial_params(1.0)
eng.set_option("dbg", dbg)
for _ in range(2):
    try:
        eng.train_eval(x)
    except Exception as e:
        print("dbg", dbg, "error", e)
print("dbg %d: %.1f ms" % (dbg, eng.last_timing()[1]), flush=True)

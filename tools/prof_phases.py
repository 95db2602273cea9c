"""In-kernel phase profile of one train evaluation of the scaled-linear pipeline (option "profile").  args: n L"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from rnaelem_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
eng = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
seqs, quals = synth.synth_batch(n, L)
eng.load_batch(seqs, quals)
x = eng.initial_params(1.0)
eng.train_eval(x)
eng.set_option("profile", 1)
fn, gr, eff, nsk = eng.train_eval(x)
ms = eng.last_timing()
c = eng.profile()
names = ["in setup", "in stage", "in products", "in items", "in unary", "out setup", "out stage", "out products", "out item records",
         "out item terms", "out (unused)", "out unary", "out flush"]
tot = c[:13].sum()
print("n=%d L=%d pipeline %.1f ms -> %.0f seq/s ; fn=%.6f" % (n, L, ms[1], n / ms[1] * 1e3, fn))
for k, nm in enumerate(names):
    print("  %-16s %6.2f %%  (%.3g cycles)" % (nm, 100 * c[k] / tot, c[k]))

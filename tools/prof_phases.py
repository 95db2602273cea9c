"""In-kernel phase profile of one train evaluation (option "profile")."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from rnaelem_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 0
eng = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
if slots:
    eng.set_option("slots", slots)
seqs, quals = synth.synth_batch(n, L)
t0 = time.time(); eng.load_batch(seqs, quals); print("load_batch %.2fs" % (time.time() - t0))
x = eng.initial_params(1.0)
eng.train_eval(x)
eng.set_option("profile", 1)
fn, gr, eff, nsk = eng.train_eval(x)
ms = eng.last_timing()
c = eng.profile()
names = ["stage", "in-U", "in-ext", "out-ext", "out-U", "other", "in-heavy", "out-heavy"]
tot = c[:8].sum()
print("n=%d L=%d kernel %.1f ms -> %.0f seq/s ; fn=%.6f" % (n, L, ms[1], n / ms[1] * 1e3, fn))
for k, nm in enumerate(names):
    print("  %-9s %6.2f %%  (%.3g cycles)" % (nm, 100 * c[k] / tot, c[k]))

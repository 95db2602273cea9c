"""One load + a few train evaluations (target for rocprofv3)."""
import sys, time
sys.path.insert(0, ".")
from rnaelem_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
slots = int(sys.argv[4]) if len(sys.argv) > 4 else 0
eng = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
if slots:
    eng.set_option("slots", slots)
seqs, quals = synth.synth_batch(n, L)
eng.load_batch(seqs, quals)
x = eng.initial_params(1.0)
for _ in range(reps):
    fn, gr, eff, nsk = eng.train_eval(x)
    print("kernel ms", eng.last_timing(), "fn", fn)

"""Train-evaluation time against the lockstep group size.  args: n L pipeline groups..."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rnaelem_amd import api, synth
n, L, pipeline = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
groups = [int(g) for g in sys.argv[4:]]
eng = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
eng.set_option("pipeline", pipeline)
seqs, quals = synth.synth_batch(n, L)
eng.load_batch(seqs, quals)
x = eng.initial_params(1.0)
for g in groups:
    eng.set_option("group", g)
    eng.train_eval(x)
    fn, gr, eff, nsk = eng.train_eval(x)
    ms = eng.last_timing()
    print("pipeline %d group %5d: %.1f ms -> %.0f seq/s (fn %.10g)" % (pipeline, g, ms[1], n / ms[1] * 1e3, fn), flush=True)

import sys, time
sys.path.insert(0, ".")
from rnaelem_amd import api, synth
n = int(sys.argv[1]); L = int(sys.argv[2])
eng = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
seqs, quals = synth.synth_batch(n, L)
t0 = time.time(); eng.load_batch(seqs, quals); print("load %.2fs" % (time.time() - t0))
x = eng.initial_params(1.0)
for g in [int(v) for v in sys.argv[3:]]:
    eng.set_option("group", g)
    eng.train_eval(x)
    fn, gr, eff, nsk = eng.train_eval(x)
    ms = eng.last_timing()
    print("group %5d: %.1f ms  -> %.0f seq/s  fn=%.9f" % (g, ms[1], n / ms[1] * 1e3, fn))

"""End-to-end check: L-BFGS-B training on synthetic sequences through the C ABI (prints objective per evaluation)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from rnaelem_amd import api, synth, train
n, L, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pattern = sys.argv[4] if len(sys.argv) > 4 else "((.*.))"
eng = api.Engine(pattern, "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
seqs, quals = synth.synth_batch(n, L)
eng.load_batch(seqs, quals)
x0 = eng.initial_params(0.0)
fl = []
def ev(x):
    r = eng.train_eval(x)
    fl.append(int(eng.last_timing()[2]))
    return r
t0 = time.time()
res = train.train(ev, x0, 0.1, 0.1, max_iter=iters, epsilon=1e-5, log=lambda m: print(m, flush=True))
dt = time.time() - t0
print("%s: %d iterations, %d evaluations in %.1f s (%.0f seq/s incl. host loop); f %.6g; fallback sequences per eval: max %d" % (
    res["message"], res["n_iter"], res["n_eval"], dt, n * res["n_eval"] / dt, res["f"], max(fl)))
print("lambda", res["x"][-2:], "finite", bool(np.all(np.isfinite(res["x"]))))

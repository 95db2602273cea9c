"""Wall time per iteration of the default training mode (mini-batch of 64 + shuffled negatives, Adam).  args: n L iters"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from rnaelem_amd import api, synth, train
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100
iters = int(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith('-') else 30
eng = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
seqs, quals = synth.synth_batch(n, L)
t_load = [0.0]

def ev_batch(s2, q2, x):
    t0 = time.perf_counter()
    eng.load_batch(s2, q2)
    t_load[0] += time.perf_counter() - t0
    return eng.train_eval(x) + (eng.seq_stats()[:, 4] != 0,)

def ev_joint(s2, q2, x, n_rec):
    t0 = time.perf_counter()
    eng.load_batch(s2, q2)
    t_load[0] += time.perf_counter() - t0
    fn, gr, _, nsk = eng.train_eval(x)
    skipped = eng.seq_stats()[:, 4] != 0
    return fn, gr, float(eng.bpp_eff()[:n_rec][~skipped[:n_rec]].sum()), nsk, skipped

if "--two-step" in sys.argv or "--joint" in sys.argv:
    ev = train.MiniBatches(seqs, quals, 64, ev_batch, kmer_shuf=2, evaluate_joint=None if "--two-step" in sys.argv else ev_joint)
else:   # the command line's way: two engines, the next batch loads while the current one is evaluated
    eng2 = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
    ev = train.MiniBatches(seqs, quals, 64, None, kmer_shuf=2, engines=[eng, eng2])
x0 = eng.initial_params(0.0)
rho = train.regularisation(len(x0), 0.1, 0.1)
train.minimize_adam(ev, x0, rho, max_iter=18)   # (untimed: both engines have evaluated a batch of theirs)
t_load[0] = 0.0
t0 = time.perf_counter()
train.minimize_adam(ev, x0, rho, max_iter=iters)
dt = time.perf_counter() - t0
if hasattr(ev, "finish"):
    ev.finish()
print("n=%d L=%d batch 64 + negatives: %.1f ms / iteration (%.1f ms in the two load_batch calls) -> %.0f seq/s" % (
    n, L, dt / iters * 1e3, t_load[0] / iters * 1e3, 128 * iters / dt))

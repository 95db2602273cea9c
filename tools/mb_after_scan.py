"""Debug: the default-mode (mini-batch) iteration time after a large scan in the same process (the order bench.py runs them in).
args: [--no-scan]"""
import gc, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from rnaelem_amd import api, synth, train
if "--no-scan" not in sys.argv:
    sec = bench.ScanSecondary(api, synth, 0)
    print("scan secondary:", sec.measure()["value"], flush=True)
    del sec
    gc.collect()
eng = api.Engine(bench.PATTERN, "~T2004~", bench.MAX_SPAN, bench.MAX_ILOOP, 1e-4, 0.1, 0, 0)
eng2 = api.Engine(bench.PATTERN, "~T2004~", bench.MAX_SPAN, bench.MAX_ILOOP, 1e-4, 0.1, 0, 0)
seqs, quals = synth.synth_batch(2000, 200)
ev = train.MiniBatches(seqs, quals, 64, None, kmer_shuf=2, engines=[eng, eng2])
x0 = eng.initial_params(0.0)
rho = train.regularisation(len(x0), 0.1, 0.1)
orig = ev.__call__
times = []
class Timed:
    def __init__(self, ev): self.ev = ev
    def __call__(self, x):
        t0 = time.perf_counter(); r = self.ev(x); times.append((time.perf_counter() - t0) * 1e3); return r
    def __getattr__(self, k): return getattr(self.ev, k)
tev = Timed(ev)
train.minimize_adam(tev, x0, rho, max_iter=6)
del times[:]
t0 = time.perf_counter()
train.minimize_adam(tev, x0, rho, max_iter=40)
dt = time.perf_counter() - t0
ev.finish()
print("%.1f ms / iteration; per call ms: %s" % (dt / 40 * 1e3, " ".join("%.0f" % t for t in times)))

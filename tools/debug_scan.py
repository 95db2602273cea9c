import sys
sys.path.insert(0, ".")
import numpy as np
from oracle import pyoracle as po
from rnaelem_amd import api, io, synth
from tests.util import gpath

m = io.read_model(gpath(sys.argv[1] if len(sys.argv) > 1 else "tiny_a.model"))
recs = io.read_fastq(gpath(sys.argv[2] if len(sys.argv) > 2 else "tiny.fq"))
eng = io.engine_from_model(m)
o, x = po.oracle_from_model(gpath(sys.argv[1] if len(sys.argv) > 1 else "tiny_a.model"))
eng.load_batch([s for _, s, _ in recs], [q for _, _, q in recs])
import ctypes as C
n, off, qoff = eng.n_seq, eng._off, eng._qoff
tot = int(off[-1])
start, inner, end = np.zeros(tot), np.zeros(tot), np.zeros(int(qoff[-1]))
psi = np.zeros(tot, dtype=np.int32)
rss = C.create_string_buffer(tot + 1)
ys, ye = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
ex, en = np.zeros(n), np.zeros(eng.n_param - 2)
so = api.ScanOut(api._dp(start), api._dp(end), api._dp(inner), api._i32(psi), C.cast(rss, C.c_char_p), api._i32(ys), api._i32(ye), api._dp(ex), api._dp(en))
eng._check(eng._lib.elemdp_scan(eng._h, api._dp(m["x"]), eng.n_param, C.byref(so)))
raw = rss.raw[:tot]
for k, (rid, seq, qual) in enumerate(recs):
    a = o.scan_seq(seq, qual)
    lo, hi = int(off[k]), int(off[k + 1])
    print(rid, "Ys/Ye gpu", ys[k], ye[k], "oracle", a["Ys"], a["Ye"])
    print(" rss gpu   ", repr(raw[lo:hi]))
    print(" rss oracle", repr(a["rss"]))
    print(" psi gpu   ", psi[lo:hi].tolist())
    print(" psi oracle", a["psihat"].tolist())
    print(" start maxdiff", np.nanmax(np.abs(np.where(np.isinf(a["start"]), 0, start[lo:hi] - a["start"]))))
    print(" end   maxdiff", np.nanmax(np.abs(np.where(np.isinf(a["end"]), 0, end[int(qoff[k]):int(qoff[k + 1])] - a["end"]))))
    print(" inner maxdiff", np.nanmax(np.abs(np.where(np.isinf(a["inner"]), 0, inner[lo:hi] - a["inner"]))))

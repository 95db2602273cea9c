"""ctypes wrapper around oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

May be imported from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from
rnaelem_amd/ (the product).  See oracle/elem_oracle.h for the C interface.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
LIB = os.path.join(HERE, "liboracle.so")
DEFAULT_PAR = os.path.join(REPO, "rnaelem_amd", "data", "turner2004.elempar")

NO_RSS, NO_PRF, NO_ENE, THETA_SOFTMAX, LIK_RATIO = 1, 2, 4, 8, 16
DBG_NO_THETA, DBG_FIX_RSS, DBG_NO_TURN = 1 << 8, 1 << 9, 1 << 10

_CODE = np.zeros(256, dtype=np.uint8)
for _c, _v in (("A", 1), ("a", 1), ("C", 2), ("c", 2), ("G", 3), ("g", 3), ("U", 4), ("u", 4), ("T", 4), ("t", 4)):
    _CODE[ord(_c)] = _v


def encode_seq(s):
    return _CODE[np.frombuffer(s.encode(), dtype=np.uint8)].copy()


def encode_qual(q):
    return (np.frombuffer(q.encode(), dtype=np.uint8) - 33).astype(np.uint8)


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


class SeqResult(C.Structure):
    _fields_ = [("Zo", C.c_double), ("Zari", C.c_double), ("Znasi", C.c_double), ("f", C.c_double),
                ("bpp_eff", C.c_double), ("skipped", C.c_int), ("L", C.c_int), ("W", C.c_int)]


class ScanResult(C.Structure):
    _fields_ = [("Ys", C.c_int), ("Ye", C.c_int), ("exist_prob", C.c_double), ("ZL", C.c_double),
                ("ZeL", C.c_double), ("PyNL", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        dp = C.POINTER(C.c_double)
        u8 = C.POINTER(C.c_uint8)
        i32 = C.POINTER(C.c_int32)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        for f in ("orc_n_param", "orc_n_state", "orc_n_node"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_get_params.argtypes = [C.c_void_p, dp]
        L.orc_set_params.argtypes = [C.c_void_p, dp]
        L.orc_hmm_json.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.orc_energy_table.argtypes = [C.c_void_p, C.c_char_p, dp, C.c_int]
        L.orc_hairpin_energy.restype = C.c_double
        L.orc_hairpin_energy.argtypes = [C.c_void_p, u8, C.c_int, C.c_int, C.c_int]
        L.orc_loop_energy.restype = C.c_double
        L.orc_loop_energy.argtypes = [C.c_void_p, u8, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_sum_ext_m.restype = C.c_double
        L.orc_sum_ext_m.argtypes = [C.c_void_p, u8, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_bpp.argtypes = [C.c_void_p, u8, C.c_int, dp, u8, dp, dp]
        L.orc_train_seq.argtypes = [C.c_void_p, u8, C.c_int, u8, C.c_char_p, C.POINTER(SeqResult)] + [dp] * 8
        L.orc_train_eval.argtypes = [C.c_void_p, dp, u8, i32, u8, i32, C.c_int, C.c_int, dp, dp, dp, i32]
        L.orc_scan_seq.argtypes = [C.c_void_p, u8, C.c_int, u8, C.POINTER(ScanResult), dp, dp, dp, i32, C.c_char_p, dp]
        _lib = L
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Oracle:
    """One reference-equivalent model (pattern + energy tables + hyper-parameters)."""

    def __init__(self, pattern, max_span=50, max_iloop=30, min_bpp=1e-4, tau=0.1, flags=0, par_text=None,
                 lam=(0.0, 0.0)):
        if par_text is None:
            par_text = open(DEFAULT_PAR).read()
        self.h = lib().orc_create(pattern.encode(), par_text.encode(), max_span, max_iloop, min_bpp, tau, flags)
        if not self.h:
            raise RuntimeError(lib().orc_last_error().decode())
        self.flags = flags
        self.n_param = lib().orc_n_param(self.h)
        self.S = lib().orc_n_state(self.h)
        self.M = lib().orc_n_node(self.h)
        x = self.get_params()
        x[-2:] = lam
        self.set_params(x)

    def __del__(self):
        try:
            if self.h:
                lib().orc_destroy(self.h)
        except Exception:
            pass

    def get_params(self):
        x = np.zeros(self.n_param)
        lib().orc_get_params(self.h, _dp(x))
        return x

    def set_params(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.size == self.n_param
        lib().orc_set_params(self.h, _dp(x))

    def hmm(self):
        buf = C.create_string_buffer(1 << 20)
        n = lib().orc_hmm_json(self.h, buf, len(buf))
        assert n > 0
        return json.loads(buf.value.decode())

    def energy_table(self, name):
        out = np.zeros(40000)
        n = lib().orc_energy_table(self.h, name.encode(), _dp(out), out.size)
        assert n > 0, name
        return out[:n].copy()

    def hairpin_energy(self, seq, i, j):
        return lib().orc_hairpin_energy(self.h, _u8(seq), len(seq), i, j)

    def loop_energy(self, seq, i, j, p, q):
        return lib().orc_loop_energy(self.h, _u8(seq), len(seq), i, j, p, q)

    def sum_ext_m(self, seq, i, j, ext):
        return lib().orc_sum_ext_m(self.h, _u8(seq), len(seq), i, j, int(ext))

    def bpp(self, seq):
        """-> (lnbpp[(L+1),(W+1)], kept[(L+1),(W+1)] uint8, bpp_eff, lnZ)"""
        L = len(seq)
        W = min(L, self._max_span())
        ln = np.full((L + 1, W + 1), -np.inf)
        kept = np.zeros((L + 1, W + 1), dtype=np.uint8)
        eff = C.c_double()
        lnz = C.c_double()
        rc = lib().orc_bpp(self.h, _u8(seq), L, _dp(ln), _u8(kept), C.byref(eff), C.byref(lnz))
        if rc:
            raise RuntimeError(lib().orc_last_error().decode())
        return ln, kept, eff.value, lnz.value

    def _max_span(self):
        return getattr(self, "max_span_", 1 << 30)

    def train_seq(self, seq, qual, fix_rss=None, tables=False):
        """One sequence through the reference's train schedule.  Returns a dict."""
        L = len(seq)
        res = SeqResult()
        nt = self.n_param - 2
        ENo, ENx = np.zeros(nt), np.zeros(nt)
        EHo, EHx = np.zeros(2), np.zeros(2)
        io = np.zeros((L + 1) * self.S)
        oo = np.zeros((L + 1) * self.S)
        ins = outs = None
        if tables:
            W = min(L, self._max_span())
            ins = np.zeros((L + 1) * (W + 1) * 7 * self.S)
            outs = np.zeros((L + 1) * (W + 1) * 7 * self.S)
        rc = lib().orc_train_seq(self.h, _u8(seq), L, _u8(qual), fix_rss.encode() if fix_rss else None, C.byref(res),
                                 _dp(ENo), _dp(EHo), _dp(ENx), _dp(EHx), _dp(io), _dp(ins), _dp(outs), _dp(oo))
        if rc:
            raise RuntimeError(lib().orc_last_error().decode())
        out = dict(Zo=res.Zo, Zari=res.Zari, Znasi=res.Znasi, f=res.f, bpp_eff=res.bpp_eff, skipped=res.skipped,
                   L=res.L, W=res.W, ENo=ENo, EHo=EHo, ENx=ENx, EHx=EHx, inside_o=io.reshape(L + 1, self.S),
                   outside_o=oo.reshape(L + 1, self.S))
        if tables:
            out["inside"] = ins.reshape(L + 1, res.W + 1, 7, self.S)
            out["outside"] = outs.reshape(L + 1, res.W + 1, 7, self.S)
        return out

    def train_eval(self, x, seqs, quals, n_threads=1):
        """fn, gr, sum_eff, n_skipped over a batch (lists of uint8 arrays)."""
        off = np.zeros(len(seqs) + 1, dtype=np.int32)
        qoff = np.zeros(len(seqs) + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(s) for s in seqs])
        qoff[1:] = np.cumsum([len(q) for q in quals])
        sc = np.concatenate(seqs).astype(np.uint8)
        qc = np.concatenate(quals).astype(np.uint8)
        x = np.ascontiguousarray(x, dtype=np.float64)
        gr = np.zeros(self.n_param)
        fn, eff = C.c_double(), C.c_double()
        nsk = C.c_int32()
        rc = lib().orc_train_eval(self.h, _dp(x), _u8(sc), _i32(off), _u8(qc), _i32(qoff), len(seqs), n_threads,
                                  C.byref(fn), _dp(gr), C.byref(eff), C.byref(nsk))
        if rc:
            raise RuntimeError(lib().orc_last_error().decode())
        return fn.value, gr, eff.value, nsk.value

    def scan_seq(self, seq, qual):
        L = len(seq)
        res = ScanResult()
        start, end, inner = np.zeros(L), np.zeros(L + 1), np.zeros(L)
        psi = np.zeros(L, dtype=np.int32)
        rss = C.create_string_buffer(L + 1)
        EN = np.zeros(self.n_param - 2)
        rc = lib().orc_scan_seq(self.h, _u8(seq), L, _u8(qual), C.byref(res), _dp(start), _dp(end), _dp(inner),
                                _i32(psi), rss, _dp(EN))
        if rc:
            raise RuntimeError(lib().orc_last_error().decode())
        return dict(Ys=res.Ys, Ye=res.Ye, exist_prob=res.exist_prob, ZL=res.ZL, ZeL=res.ZeL, PyNL=res.PyNL,
                    start=start, end=end, inner=inner, psihat=psi, rss=rss.raw[:L].decode(), EN=EN)


def make_oracle(pattern, max_span=50, max_iloop=30, **kw):
    o = Oracle(pattern, max_span=max_span, max_iloop=max_iloop, **kw)
    o.max_span_ = max_span
    return o


def read_fastq(path):
    """4-line FASTQ with L+1 quality chars (fastq_io.hpp:64-108).  -> list of (id, seq_codes, qual)"""
    out = []
    with open(path) as f:
        lines = f.read().split("\n")
    for k in range(0, len(lines) - 3, 4):
        rid, s, _, q = lines[k:k + 4]
        out.append((rid, encode_seq(s), encode_qual(q)))
    return out


def read_model(path):
    """Parse the reference's model text format (motif_io.hpp:118-262) -> dict."""
    d = {}
    for line in open(path):
        if ": " not in line:
            continue
        k, v = line.split(": ", 1)
        d[k.strip()] = v.strip()
    out = dict(pattern=d["pattern"], max_span=int(d["max-span"]), max_iloop=int(d["max-internal-loop"]),
               tau=float(d["tau"]), min_bpp=float(d["min-bpp"]), lam=json.loads(d["lambda"]),
               softmax=bool(int(d.get("theta-softmax", "0"))), no_rss=bool(int(d.get("no-rss", "0"))),
               no_prf=bool(int(d.get("no-profile", "0"))), no_ene=bool(int(d.get("no-energy", "0"))),
               ene_param=d.get("ene-param", "~T2004~"))
    out["w"] = json.loads(d["s"] if "s" in d else d["theta"])
    return out


def energy_param_text(name):
    """parameter text of `ene-param` (energy_model.hpp:150-160): the two built-in sets, or a parameter file"""
    if name == "~T2004~":
        return open(DEFAULT_PAR).read()
    if name == "~A2007~":
        return open(os.path.join(REPO, "rnaelem_amd", "data", "andronescu2007.elempar")).read()
    return open(name).read()


def oracle_from_model(path, extra_flags=0):
    md = read_model(path)
    flags = (NO_RSS if md["no_rss"] else 0) | (NO_PRF if md["no_prf"] else 0) | (NO_ENE if md["no_ene"] else 0) | \
        (THETA_SOFTMAX if md["softmax"] else 0) | extra_flags
    o = make_oracle(md["pattern"], md["max_span"], md["max_iloop"], min_bpp=md["min_bpp"], tau=md["tau"], flags=flags,
                    par_text=energy_param_text(md["ene_param"]))
    x = np.array([v for row in md["w"] for v in row] + list(md["lam"]), dtype=np.float64)
    o.set_params(x)
    return o, x

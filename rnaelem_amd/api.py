"""Host-side mirror of the reference's operator interface, over the C ABI of libelemdp.so.

    RNAelemTrainer::operator()(x, fn, gr)   RNAelem/motif_trainer.hpp:595   ->  Engine.train_eval(x)
    RNAelemScanner::scan(model)             RNAelem/motif_scanner.hpp:938   ->  Engine.scan(x)

Only ctypes and numpy are used here; torch enters in rnaelem_amd/distributed.py for the RCCL
all-reduce.  There is no CPU fallback: if the library cannot be loaded or no GPU is present the
calls raise.
"""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libelemdp.so")

NO_RSS, NO_PROFILE, NO_ENERGY, THETA_SOFTMAX, LIK_RATIO = 1, 2, 4, 8, 16
DBG_FIX_RSS, DBG_NO_TURN = 1 << 9, 1 << 10

# every symbol include/elemdp.h declares
SYMBOLS = ["elemdp_last_error", "elemdp_abi_version", "elemdp_set_data_dir", "elemdp_create", "elemdp_destroy",
           "elemdp_n_param", "elemdp_n_state", "elemdp_n_node", "elemdp_initial_params", "elemdp_describe",
           "elemdp_set_option", "elemdp_load_batch", "elemdp_batch_bpp_eff", "elemdp_batch_pairs", "elemdp_train_eval",
           "elemdp_partial_len", "elemdp_train_partial", "elemdp_train_finish", "elemdp_set_finish_params", "elemdp_train_seq_stats",
           "elemdp_debug_tables", "elemdp_scan", "elemdp_last_timing", "elemdp_debug_profile", "elemdp_kernel_name", "elemdp_kmer_shuffle", "elemdp_epoch_permutation",
           "elemdp_comm_unique_id", "elemdp_comm_init", "elemdp_comm_destroy"]


class ModelDesc(C.Structure):
    _fields_ = [("pattern", C.c_char_p), ("energy_param", C.c_char_p), ("max_span", C.c_int32), ("max_iloop", C.c_int32),
                ("min_bpp", C.c_double), ("tau", C.c_double), ("flags", C.c_int32), ("device", C.c_int32)]


class ScanOut(C.Structure):
    _fields_ = [("start", C.POINTER(C.c_double)), ("end", C.POINTER(C.c_double)), ("inner", C.POINTER(C.c_double)),
                ("psihat", C.POINTER(C.c_int32)), ("rss", C.c_char_p), ("ys", C.POINTER(C.c_int32)),
                ("ye", C.POINTER(C.c_int32)), ("exist_prob", C.POINTER(C.c_double)), ("en", C.POINTER(C.c_double))]


class ElemdpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libelemdp error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load_library():
    """dlopen libelemdp.so (fails loudly if it has not been built: there is no fallback)."""
    global _lib
    if _lib is None:
        # (ELEMDP_LIBRARY: another build of the same sources -- a timing variant, or the host-sanitizer build of tools/sanitize_cpu.sh)
        path = os.environ.get("ELEMDP_LIBRARY") or LIB_PATH
        if not os.path.exists(path):
            raise ElemdpError(-100, "%s is not built (run `python -m rnaelem_amd.build`)" % path)
        L = C.CDLL(path)
        dp, u8, i32 = C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
        hp = C.c_void_p
        L.elemdp_last_error.restype = C.c_char_p
        L.elemdp_kernel_name.restype = C.c_char_p
        L.elemdp_set_data_dir.argtypes = [C.c_char_p]
        if path != LIB_PATH:      # (the library looks for its energy parameter files next to itself)
            L.elemdp_set_data_dir(os.path.join(os.path.dirname(LIB_PATH), "data").encode())
        L.elemdp_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(hp)]
        L.elemdp_destroy.argtypes = [hp]
        for f in ("elemdp_n_param", "elemdp_n_state", "elemdp_n_node", "elemdp_partial_len"):
            getattr(L, f).argtypes = [hp]
        L.elemdp_initial_params.argtypes = [hp, C.c_double, dp, C.c_int32]
        L.elemdp_describe.argtypes = [hp, C.c_char_p, C.c_int32]
        L.elemdp_set_option.argtypes = [hp, C.c_char_p, C.c_double]
        L.elemdp_load_batch.argtypes = [hp, u8, i32, u8, i32, C.c_char_p, C.c_int32]
        L.elemdp_batch_bpp_eff.argtypes = [hp, dp, C.c_int32]
        L.elemdp_batch_pairs.argtypes = [hp, C.c_int32, u8, dp, C.c_int32]
        L.elemdp_train_eval.argtypes = [hp, dp, C.c_int32, dp, dp, dp, i32]
        L.elemdp_train_partial.argtypes = [hp, dp, C.c_int32, C.c_void_p, C.c_int32]
        L.elemdp_train_finish.argtypes = [hp, dp, dp, dp, dp, i32]
        L.elemdp_set_finish_params.argtypes = [hp, dp, C.c_int32]
        L.elemdp_train_seq_stats.argtypes = [hp, dp, C.c_int32]
        L.elemdp_debug_tables.argtypes = [hp] + [dp] * 7
        L.elemdp_scan.argtypes = [hp, dp, C.c_int32, C.POINTER(ScanOut)]
        L.elemdp_last_timing.argtypes = [hp, dp, C.c_int32]
        L.elemdp_epoch_permutation.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
        L.elemdp_kmer_shuffle.argtypes = [C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]
        L.elemdp_debug_profile.argtypes = [hp, dp, C.c_int32]
        L.elemdp_comm_unique_id.argtypes = [C.c_char_p]
        L.elemdp_comm_init.argtypes = [hp, C.c_int32, C.c_int32, C.c_char_p]
        L.elemdp_comm_destroy.argtypes = [hp]
        _lib = L
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def epoch_permutation(n, seed):
    """perm of std::shuffle(.., std::mt19937(seed)) on n elements (host; elemdp_epoch_permutation)."""
    perm = np.zeros(n, dtype=np.int32)
    rc = load_library().elemdp_epoch_permutation(int(n), int(seed), _i32(perm))
    if rc:
        raise ElemdpError(rc, "epoch_permutation")
    return perm


def kmer_shuffle(codes, k, iter_cnt):
    """Shuffled negative of one sequence (host; elemdp_kmer_shuffle)."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    out = np.zeros_like(codes)
    rc = load_library().elemdp_kmer_shuffle(_u8(codes), len(codes), int(k), int(iter_cnt), _u8(out))
    if rc:
        raise ElemdpError(rc, "kmer_shuffle")
    return out


class Engine:
    """One model on one GPU (== one `RNAelem` object plus the trainer / scanner workers around it)."""

    def __init__(self, pattern, energy_param=None, max_span=50, max_iloop=30, min_bpp=1e-4, tau=0.1, flags=0, device=-1):
        self._lib = load_library()
        self._h = C.c_void_p()
        self._keep = (pattern.encode(), None if energy_param is None else energy_param.encode())
        d = ModelDesc(self._keep[0], self._keep[1], max_span, max_iloop, min_bpp, tau, flags, device)
        self._check(self._lib.elemdp_create(C.byref(d), C.byref(self._h)))
        self.flags = flags
        self.max_span = max_span
        self.n_param = self._lib.elemdp_n_param(self._h)
        self.n_state = self._lib.elemdp_n_state(self._h)
        self.n_node = self._lib.elemdp_n_node(self._h)
        self.n_seq = 0
        self._off = self._qoff = None

    def _check(self, rc):
        if rc < 0:
            raise ElemdpError(rc, self._lib.elemdp_last_error().decode())
        return rc

    def close(self):
        if getattr(self, "_h", None):
            self._lib.elemdp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- model
    def initial_params(self, lambda_init=0.0):
        x = np.zeros(self.n_param)
        self._check(self._lib.elemdp_initial_params(self._h, lambda_init, _dp(x), self.n_param))
        return x

    def describe(self):
        buf = C.create_string_buffer(1 << 20)
        self._check(self._lib.elemdp_describe(self._h, buf, len(buf)))
        return json.loads(buf.value.decode())

    def set_option(self, key, value):
        self._check(self._lib.elemdp_set_option(self._h, key.encode(), float(value)))

    # ---- batch
    def load_batch(self, seqs, quals, fix_rss=None):
        """seqs: list of uint8 code arrays; quals: list of uint8 arrays with len(seq)+1 entries."""
        n = len(seqs)
        off = np.zeros(n + 1, dtype=np.int32)
        qoff = np.zeros(n + 1, dtype=np.int32)
        if n:
            off[1:] = np.cumsum([len(s) for s in seqs])
            qoff[1:] = np.cumsum([len(q) for q in quals])
        sc = np.ascontiguousarray(np.concatenate(seqs) if n else np.zeros(1), dtype=np.uint8)
        qc = np.ascontiguousarray(np.concatenate(quals) if n else np.zeros(1), dtype=np.uint8)
        fx = None if fix_rss is None else "".join(fix_rss).encode()
        self._check(self._lib.elemdp_load_batch(self._h, _u8(sc), _i32(off), _u8(qc), _i32(qoff), fx, n))
        self.n_seq, self._off, self._qoff = n, off, qoff

    def bpp_eff(self):
        out = np.zeros(self.n_seq)
        self._check(self._lib.elemdp_batch_bpp_eff(self._h, _dp(out), self.n_seq))
        return out

    def pairs(self, index, with_lnbpp=False):
        L = int(self._off[index + 1] - self._off[index])
        W = min(L, self.max_span)
        kept = np.zeros((L + 1, W + 1), dtype=np.uint8)
        ln = np.full((L + 1, W + 1), -np.inf) if with_lnbpp else None
        self._check(self._lib.elemdp_batch_pairs(self._h, index, _u8(kept), _dp(ln), kept.size))
        return kept, ln

    # ---- training: == RNAelemTrainer::operator()
    def train_eval(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        gr = np.zeros(self.n_param)
        fn, eff, nsk = C.c_double(), C.c_double(), C.c_int32()
        self._check(self._lib.elemdp_train_eval(self._h, _dp(x), self.n_param, C.byref(fn), _dp(gr), C.byref(eff),
                                                C.byref(nsk)))
        return fn.value, gr, eff.value, nsk.value

    def partial_len(self):
        return self._lib.elemdp_partial_len(self._h)

    def train_partial(self, x, out=None, device_ptr=None):
        """Local sums before the cross-rank reduction.  `device_ptr`: raw device address to write to."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        if device_ptr is not None:
            self._check(self._lib.elemdp_train_partial(self._h, _dp(x), self.n_param, C.c_void_p(device_ptr), 1))
            return None
        if out is None:
            out = np.zeros(self.partial_len())
        self._check(self._lib.elemdp_train_partial(self._h, _dp(x), self.n_param, out.ctypes.data_as(C.c_void_p), 0))
        return out

    def train_finish(self, reduced, x=None):
        reduced = np.ascontiguousarray(reduced, dtype=np.float64)
        if x is not None:
            x = np.ascontiguousarray(x, dtype=np.float64)
            self._check(self._lib.elemdp_set_finish_params(self._h, _dp(x), self.n_param))
        gr = np.zeros(self.n_param)
        fn, eff, nsk = C.c_double(), C.c_double(), C.c_int32()
        self._check(self._lib.elemdp_train_finish(self._h, _dp(reduced), C.byref(fn), _dp(gr), C.byref(eff), C.byref(nsk)))
        return fn.value, gr, eff.value, nsk.value

    # ---- in-library collective (RCCL) for hosts without torch.distributed
    @staticmethod
    def comm_unique_id():
        """128-byte id (ncclUniqueId) that rank 0 creates and hands to the other ranks"""
        buf = C.create_string_buffer(128)
        rc = load_library().elemdp_comm_unique_id(buf)
        if rc < 0:
            raise ElemdpError(rc, load_library().elemdp_last_error().decode())
        return buf.raw

    def comm_init(self, rank, world, uid):
        self._check(self._lib.elemdp_comm_init(self._h, rank, world, C.c_char_p(uid)))

    def comm_destroy(self):
        self._check(self._lib.elemdp_comm_destroy(self._h))

    def seq_stats(self):
        out = np.zeros((self.n_seq, 5))
        self._check(self._lib.elemdp_train_seq_stats(self._h, _dp(out), self.n_seq))
        return out

    def debug_tables(self):
        assert self.n_seq == 1
        L = int(self._off[1])
        W = min(L, self.max_span)
        S, nt = self.n_state, self.n_param - 2
        ins = np.zeros((L + 1, W + 1, 7, S))
        outs = np.zeros((L + 1, W + 1, 7, S))
        io, oo = np.zeros((L + 1, S)), np.zeros((L + 1, S))
        ENo, ENx, EH = np.zeros(nt), np.zeros(nt), np.zeros(4)
        self._check(self._lib.elemdp_debug_tables(self._h, _dp(ins), _dp(outs), _dp(io), _dp(oo), _dp(ENo), _dp(ENx), _dp(EH)))
        return dict(inside=ins, outside=outs, inside_o=io, outside_o=oo, ENo=ENo, ENx=ENx, EHo=EH[:2], EHx=EH[2:])

    # ---- scanning: == RNAelemScanner::scan
    def scan(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        n, off, qoff = self.n_seq, self._off, self._qoff
        tot = int(off[-1])
        start, inner, end = np.zeros(tot), np.zeros(tot), np.zeros(int(qoff[-1]))
        psi = np.zeros(tot, dtype=np.int32)
        rss = C.create_string_buffer(tot + 1)
        ys, ye = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        ex, en = np.zeros(n), np.zeros(self.n_param - 2)
        so = ScanOut(_dp(start), _dp(end), _dp(inner), _i32(psi), C.cast(rss, C.c_char_p), _i32(ys), _i32(ye), _dp(ex), _dp(en))
        self._check(self._lib.elemdp_scan(self._h, _dp(x), self.n_param, C.byref(so)))
        raw = rss.raw[:tot].decode()
        recs = []
        for k in range(n):
            a, b = int(off[k]), int(off[k + 1])
            recs.append(dict(start=start[a:b], inner=inner[a:b], end=end[int(qoff[k]):int(qoff[k + 1])], psihat=psi[a:b],
                             rss=raw[a:b], Ys=int(ys[k]), Ye=int(ye[k]), exist_prob=float(ex[k])))
        return recs, en

    def last_timing(self):
        """[ms whole evaluation, ms DP pipeline, sequences re-evaluated in log space]"""
        ms = np.zeros(3)
        self._lib.elemdp_last_timing(self._h, _dp(ms), 3)
        return ms

    def profile(self):
        c = np.zeros(16)
        self._lib.elemdp_debug_profile(self._h, _dp(c), 16)
        return c

    def kernel_name(self):
        return self._lib.elemdp_kernel_name().decode()

"""ctypes wrapper of the test-only CPU emulation (tests/emul/emul.cpp)."""
import ctypes as C
import os
import json

import numpy as np

from tests.emul import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        # (ELEMDP_EMUL_LIBRARY: another build of the driver, e.g. the sanitizer build of tools/sanitize_cpu.sh)
        L = C.CDLL(os.environ.get("ELEMDP_EMUL_LIBRARY") or _build.build())
        dp, u8, i32 = C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
        L.emu_create.restype = C.c_void_p
        L.emu_create.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
        L.emu_destroy.argtypes = [C.c_void_p]
        L.emu_last_error.restype = C.c_char_p
        L.emu_n_param.argtypes = [C.c_void_p]
        L.emu_n_state.argtypes = [C.c_void_p]
        L.emu_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.emu_set_prune.argtypes = [C.c_void_p, C.c_int]
        L.emu_set_fast.argtypes = [C.c_void_p, C.c_int]
        L.emu_set_fast.restype = C.c_int
        L.emu_fast_cells.restype = C.c_long
        L.emu_check_det_lists.argtypes = [C.c_void_p]
        L.emu_energy_table.argtypes = [C.c_void_p, C.c_char_p, dp, C.c_int]
        L.emu_hairpin_energy.restype = C.c_double
        L.emu_hairpin_energy.argtypes = [C.c_void_p, u8, C.c_int, C.c_int, C.c_int]
        L.emu_loop_energy.restype = C.c_double
        L.emu_loop_energy.argtypes = [C.c_void_p, u8, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.emu_loop_weight.restype = C.c_double
        L.emu_loop_weight.argtypes = [C.c_void_p, u8, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.emu_scan_flags.argtypes = [C.c_void_p, i32, C.c_int]
        L.emu_pattern_nodes.argtypes = [C.c_void_p]
        L.emu_sum_ext_m.restype = C.c_double
        L.emu_sum_ext_m.argtypes = [C.c_void_p, u8, C.c_int, C.c_int, C.c_int, C.c_int]
        L.emu_bpp.argtypes = [C.c_void_p, u8, C.c_int, dp, u8, dp, dp]
        L.emu_train_seq.argtypes = [C.c_void_p, dp, u8, C.c_int, u8, C.c_char_p] + [dp] * 9
        L.emu_train_seq_lin.argtypes = [C.c_void_p, dp, u8, C.c_int, u8, C.c_char_p, C.c_int] + [dp] * 9
        L.emu_scan_seq.argtypes = [C.c_void_p, dp, u8, C.c_int, u8, dp, dp, dp, dp, i32, C.c_char_p, dp]
        L.emu_scan_seq_lin.argtypes = L.emu_scan_seq.argtypes
        _lib = L
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


class Emul:
    def __init__(self, pattern, par_text, max_span=50, max_iloop=30, min_bpp=1e-4, tau=0.1, flags=0):
        self.h = lib().emu_create(pattern.encode(), par_text.encode(), max_span, max_iloop, min_bpp, tau, flags)
        if not self.h:
            raise RuntimeError(lib().emu_last_error().decode())
        self.n_param = lib().emu_n_param(self.h)
        self.S = lib().emu_n_state(self.h)
        self.max_span = max_span

    def __del__(self):
        try:
            lib().emu_destroy(self.h)
        except Exception:
            pass

    def set_prune(self, on):
        lib().emu_set_prune(self.h, int(bool(on)))

    def set_fast(self, on):
        """band targets through the table-driven forms (lin_fast.h and the records of the fast blobs: what k4_in / k4_out run by
        default); returns a bit mask: 1 the automaton's lists fit the programs, 2 the one-state automaton's, 4 the shadow
        automaton's, 8 there is a shadow state"""
        return int(lib().emu_set_fast(self.h, int(bool(on))))

    def describe(self):
        buf = C.create_string_buffer(1 << 20)
        assert lib().emu_describe(self.h, buf, len(buf)) > 0
        return json.loads(buf.value.decode())

    def energy_table(self, name):
        out = np.zeros(40000)
        n = lib().emu_energy_table(self.h, name.encode(), _dp(out), out.size)
        assert n > 0, name
        return out[:n].copy()

    def hairpin_energy(self, seq, i, j):
        return lib().emu_hairpin_energy(self.h, _u8(seq), len(seq), i, j)

    def loop_energy(self, seq, i, j, p, q):
        return lib().emu_loop_energy(self.h, _u8(seq), len(seq), i, j, p, q)

    def loop_weight(self, seq, i, j, p, q):
        return lib().emu_loop_weight(self.h, _u8(seq), len(seq), i, j, p, q)

    def scan_flags(self):
        """rows {kind 0 right / 1 left / 2 pair, pl, pr, cl, cr, ScanFlag word} of every forward transition, and M"""
        out = np.zeros((4096, 6), dtype=np.int32)
        n = lib().emu_scan_flags(self.h, out.ctypes.data_as(C.POINTER(C.c_int32)), 4096)
        assert n >= 0, n
        return out[:n].copy(), lib().emu_pattern_nodes(self.h)

    def sum_ext_m(self, seq, i, j, ext):
        return lib().emu_sum_ext_m(self.h, _u8(seq), len(seq), i, j, int(ext))

    def bpp(self, seq):
        L = len(seq)
        W = min(L, self.max_span)
        ln = np.full((L + 1, W + 1), -np.inf)
        kept = np.zeros((L + 1, W + 1), dtype=np.uint8)
        eff, lnz = C.c_double(), C.c_double()
        if lib().emu_bpp(self.h, _u8(seq), L, _dp(ln), _u8(kept), C.byref(eff), C.byref(lnz)):
            raise RuntimeError(lib().emu_last_error().decode())
        return ln, kept, eff.value, lnz.value

    def train_seq(self, x, seq, qual, fix_rss=None, tables=False, linear=None):
        """linear=None: the log-space rules (dp_rules.h); 0 / 1: the scaled-linear rules (lin_rules.h) with the
        reference schedule / the ari-only + one-state schedule."""
        L = len(seq)
        W = min(L, self.max_span)
        x = np.ascontiguousarray(x, dtype=np.float64)
        nt = self.n_param - 2
        out9 = np.zeros(9)
        ENo, ENx, EHo, EHx = np.zeros(nt), np.zeros(nt), np.zeros(2), np.zeros(2)
        io, oo = np.zeros((L + 1) * self.S), np.zeros((L + 1) * self.S)
        ins = outs = None
        if tables:
            ins = np.zeros((L + 1) * (W + 1) * 7 * self.S)
            outs = np.zeros((L + 1) * (W + 1) * 7 * self.S)
        fx = fix_rss.encode() if fix_rss else None
        if linear is None:
            rc = lib().emu_train_seq(self.h, _dp(x), _u8(seq), L, _u8(qual), fx, _dp(out9),
                                     _dp(ENo), _dp(EHo), _dp(ENx), _dp(EHx), _dp(io), _dp(ins), _dp(outs), _dp(oo))
        else:
            rc = lib().emu_train_seq_lin(self.h, _dp(x), _u8(seq), L, _u8(qual), fx, int(linear), _dp(out9),
                                         _dp(ENo), _dp(EHo), _dp(ENx), _dp(EHx), _dp(io), _dp(ins), _dp(outs), _dp(oo))
        if rc:
            raise RuntimeError(lib().emu_last_error().decode())
        r = dict(Zo=out9[0], Zari=out9[1], Znasi=out9[2], f=out9[3], bpp_eff=out9[4], skipped=int(out9[5]), L=L, W=W,
                 n_items=int(out9[8]), ENo=ENo, EHo=EHo, ENx=ENx, EHx=EHx, inside_o=io.reshape(L + 1, self.S),
                 outside_o=oo.reshape(L + 1, self.S))
        if tables:
            r["inside"] = ins.reshape(L + 1, W + 1, 7, self.S)
            r["outside"] = outs.reshape(L + 1, W + 1, 7, self.S)
        return r

    def scan_seq(self, x, seq, qual, linear=False):
        L = len(seq)
        x = np.ascontiguousarray(x, dtype=np.float64)
        out6 = np.zeros(6)
        start, end, inner = np.zeros(L), np.zeros(L + 1), np.zeros(L)
        psi = np.zeros(L, dtype=np.int32)
        rss = C.create_string_buffer(L + 1)
        EN = np.zeros(self.n_param - 2)
        rc = (lib().emu_scan_seq_lin if linear else lib().emu_scan_seq)(self.h, _dp(x), _u8(seq), L, _u8(qual), _dp(out6), _dp(start), _dp(end), _dp(inner),
                                psi.ctypes.data_as(C.POINTER(C.c_int32)), rss, _dp(EN))
        if rc:
            raise RuntimeError(lib().emu_last_error().decode())
        return dict(Ys=int(out6[0]), Ye=int(out6[1]), exist_prob=out6[2], ZL=out6[3], ZeL=out6[4], PyNL=out6[5],
                    start=start, end=end, inner=inner, psihat=psi, rss=rss.raw[:L].decode(), EN=EN)

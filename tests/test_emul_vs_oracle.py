"""Host logic + rule headers of the PRODUCT checked on CPU (no GPU needed).

The per-target recurrences the HIP kernels execute (rnaelem_amd/csrc/dp_rules.h, scan_rules.h), the
plan builder (plan_rules.h), the energy evaluation (energy_rules.h), the parameter-text parser and the
automaton flattening are compiled into a test-only serial driver (tests/emul) and compared with the
oracle and the golden vectors.  The GPU orchestration around them is covered by tests marked `gpu`.
"""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.emul.pyemul import Emul
from tests.util import arr, assert_log_close, gload, gpath

PAR = open(po.DEFAULT_PAR).read()
HMM = gload("hmm.json")


def model_pair(name):
    md = po.read_model(gpath(name))
    o, x = po.oracle_from_model(gpath(name))
    flags = (2 if md["no_prf"] else 0) | (4 if md["no_ene"] else 0) | (8 if md["softmax"] else 0)
    e = Emul(md["pattern"], PAR, md["max_span"], md["max_iloop"], md["min_bpp"], md["tau"], flags)
    return o, e, x


@pytest.mark.parametrize("pattern", sorted(HMM))
def test_automaton_matches_reference(pattern):
    got = Emul(pattern, PAR).describe()
    ref = HMM[pattern]
    for k in ("reg_pattern", "M", "S", "node", "theta_id", "theta_sizes", "state", "loop_state", "reachable", "right",
              "left", "pair", "loop_loop"):
        assert got[k] == ref[k], (pattern, k)


def test_automaton_rejects_malformed_patterns():
    for bad in ("", "(.", ".)", "(.]"):
        with pytest.raises(RuntimeError):
            Emul(bad, PAR)


@pytest.mark.parametrize("fname", ["turner2004.elempar", "andronescu2007.elempar"])
def test_energy_parser_matches_oracle_bitwise(fname):
    par = open(os.path.join(os.path.dirname(po.DEFAULT_PAR), fname)).read()
    o = po.make_oracle("(.)", par_text=par)
    e = Emul("(.)", par)
    for name in ("stack", "hairpin", "bulge", "internal", "ninio", "mismatch_h", "mismatch_i", "mismatch_m", "mismatch_1ni",
                 "mismatch_23i", "mismatch_ext", "dangle5", "dangle3", "int_11", "int_21", "int_22", "triloop", "tetraloop",
                 "hexaloop", "term_au", "mlintern", "mlclosing", "ml_base", "lxc37"):
        assert np.array_equal(o.energy_table(name), e.energy_table(name)), name


def test_energy_parser_accepts_vienna_layout_with_comments():
    """A user-supplied ViennaRNA file carries comments, a 7th (NN) stack column and enthalpy sections."""
    lines = PAR.split("\n")
    out, i = [], 0
    while i < len(lines):
        out.append(lines[i])
        if lines[i].startswith("# stack"):
            out.append("/*  CG    GC    GU    UG    AU    UA    NN          */")
            for r in range(6):
                out.append(lines[i + 1 + r] + "   999    /* row */")
            out.append("  1 2 3 4 5 6 7 /* NN */")
            out.append("")
            out.append("# stack_enthalpies")
            out.append("  -1 -2 -3 -4 -5 -6 -7")
            i += 7
            continue
        i += 1
    a, b = Emul("(.)", PAR), Emul("(.)", "\n".join(out))
    for name in ("stack", "hairpin", "int_11", "mismatch_h"):
        assert np.array_equal(a.energy_table(name), b.energy_table(name)), name


def test_energy_functions_match_oracle():
    rng = np.random.RandomState(7)
    o = po.make_oracle("(.)")
    e = Emul("(.)", PAR)
    n = 0
    for _ in range(300):
        L = rng.randint(12, 60)
        s = rng.randint(1, 5, size=L).astype(np.uint8)
        if rng.rand() < 0.2:
            s[rng.randint(L)] = 0
        i = rng.randint(0, L - 8)
        j = rng.randint(i + 4, L)
        assert o.hairpin_energy(s, i, j) == e.hairpin_energy(s, i, j)
        for ext in (0, 1):
            assert o.sum_ext_m(s, i, j, ext) == e.sum_ext_m(s, i, j, ext)
            assert o.sum_ext_m(s, j, i, ext) == e.sum_ext_m(s, j, i, ext)
        if j - i >= 6:
            p = rng.randint(i + 1, min(j - 3, i + 8))
            q = rng.randint(max(p + 2, j - 8), j)
            a, b = o.loop_energy(s, i, j, p, q), e.loop_energy(s, i, j, p, q)
            if 0 in (s[i + 1], s[j - 1], s[p - 1], s[q + 1]) and p - i - 1 == 2 and j - q - 1 == 2:
                continue   # int22 with an N base: undefined in the reference
            assert a == b
            n += 1
    assert n > 150
    # special hairpins by exact lookup
    for loop, L in (("CAACG", 3), ("CUUCGG", 4), ("ACAGUACU", 6)):
        s = po.encode_seq("AA" + loop + "AA")
        assert o.hairpin_energy(s, 2, 2 + L + 1) == e.hairpin_energy(s, 2, 2 + L + 1)


@pytest.mark.parametrize("fq,W", [("1.fq", 50), ("syn_L100_n3.fq", 50), ("0.fq", 20)])
def test_bpp_filter_matches_oracle(fq, W):
    o = po.make_oracle("(.)", W, 30, min_bpp=1e-4)
    e = Emul("(.)", PAR, W, 30, 1e-4)
    for rid, seq, qual in po.read_fastq(gpath(fq)):
        ln_o, kept_o, eff_o, lnz_o = o.bpp(seq)
        ln_e, kept_e, eff_e, lnz_e = e.bpp(seq)
        assert np.array_equal(kept_o, kept_e)
        assert eff_o == eff_e
        assert lnz_e == pytest.approx(lnz_o, rel=1e-12)
        assert_log_close(ln_e, ln_o, rtol=1e-10, what="lnbpp")


CASES = [("tiny_a.model", "tiny.fq"), ("tiny_ne.model", "tiny.fq"), ("0.model", "0.fq"), ("1.model", "0.fq"), ("3.model", "0.fq"),
         ("syn_b.model", "syn_L40_n3.fq"), ("syn_sm.model", "syn_L40_n3.fq"), ("syn_c12.model", "syn_L40_n3.fq"),
         ("syn_b.model", "syn_L100_n3.fq"), ("syn_c12.model", "syn_L100_n3.fq"), ("trna_a.model", "positive_head6.fq")]


@pytest.mark.parametrize("model,fq", CASES)
def test_train_sequence_matches_oracle(model, fq):
    o, e, x = model_pair(model)
    full = fq in ("tiny.fq", "0.fq", "syn_L40_n3.fq")
    for rid, seq, qual in po.read_fastq(gpath(fq)):
        a = o.train_seq(seq, qual, tables=full)
        b = e.train_seq(x, seq, qual, tables=full)
        assert a["skipped"] == b["skipped"]
        for k in ("Zo", "Zari", "Znasi"):
            assert_log_close(b[k], a[k], rtol=1e-12, what=k)
        assert b["bpp_eff"] == a["bpp_eff"]
        assert_log_close(b["inside_o"], a["inside_o"], rtol=1e-11, what="inside_o")
        if a["skipped"]:
            continue
        assert b["f"] == pytest.approx(a["f"], rel=1e-11, abs=1e-12)
        assert_log_close(b["outside_o"], a["outside_o"], rtol=1e-11, what="outside_o")
        for k in ("ENo", "ENx", "EHo", "EHx"):
            np.testing.assert_allclose(b[k], a[k], rtol=1e-9, atol=1e-11, err_msg=k)
        if full:
            assert_log_close(b["inside"], a["inside"], rtol=1e-11, what="inside table")
            assert_log_close(b["outside"], a["outside"], rtol=1e-11, what="outside table")


# reference known-answer cases (RNAelem-test/test.cpp:88-203) through the product's rules
from tests.test_oracle_golden import EMISSION_COUNTS, PATH_COUNTS  # noqa: E402

BIG = 2 ** 31 - 1
DBG_FLAGS = 4 | (1 << 9) | (1 << 10)   # no-energy, FIX_RSS, NO_TURN


@pytest.mark.parametrize("pattern,seq,rss,count", PATH_COUNTS)
def test_reference_path_counts(pattern, seq, rss, count):
    e = Emul(pattern, PAR, BIG, BIG, 0.0, 1.0, DBG_FLAGS)
    x = np.zeros(e.n_param)          # theta = 0 == DBG_NO_THETA
    x[-2:] = 1.0
    r = e.train_seq(x, po.encode_seq(seq), np.ones(len(seq) + 1, dtype=np.uint8), fix_rss=rss)
    assert np.exp(r["Zo"]) == pytest.approx(count, rel=1e-13)
    assert np.exp(r["outside_o"][0, 0]) == pytest.approx(count, rel=1e-13)


@pytest.mark.parametrize("seq,rss,counts", EMISSION_COUNTS)
def test_reference_emission_counts(seq, rss, counts):
    e = Emul(".", PAR, BIG, BIG, 0.0, 1.0, DBG_FLAGS)
    x = np.zeros(e.n_param)
    x[-2:] = 1.0
    r = e.train_seq(x, po.encode_seq(seq), np.ones(len(seq) + 1, dtype=np.uint8), fix_rss=rss)
    np.testing.assert_allclose(r["ENo"] * np.exp(r["Zo"]), [v for row in counts for v in row], rtol=1e-12, atol=1e-12)


# ---- the scaled-linear train rules (lin_rules.h: what the k4_* kernels run) ------------------------------------
@pytest.mark.parametrize("model,fq", CASES)
@pytest.mark.parametrize("schedule", [0, 1])
def test_linear_train_rules_match_oracle(model, fq, schedule):
    o, e, x = model_pair(model)
    full = fq in ("tiny.fq", "0.fq", "syn_L40_n3.fq")
    for rid, seq, qual in po.read_fastq(gpath(fq)):
        a = o.train_seq(seq, qual, tables=full)
        b = e.train_seq(x, seq, qual, tables=full, linear=schedule)
        for k in ("Zo", "Zari", "Znasi"):
            assert_log_close(b[k], a[k], rtol=1e-11, what=k)
        assert_log_close(b["inside_o"], a["inside_o"], rtol=1e-10, what="inside_o")
        if full:
            assert_log_close(b["inside"], a["inside"], rtol=1e-10, what="inside table")
        if a["skipped"] or (schedule == 1 and not np.isfinite(a["Znasi"])):
            assert b["skipped"] == 2          # flagged: the log-space pipeline decides
            continue
        assert b["skipped"] == 0
        assert b["f"] == pytest.approx(a["f"], rel=1e-10, abs=1e-12)
        for k in ("ENo", "ENx", "EHo", "EHx"):
            np.testing.assert_allclose(b[k], a[k], rtol=1e-9, atol=1e-11, err_msg=k)
        if schedule == 0:
            assert_log_close(b["outside_o"], a["outside_o"], rtol=1e-10, what="outside_o")
            if full:
                assert_log_close(b["outside"], a["outside"], rtol=1e-10, what="outside table")


@pytest.mark.parametrize("pattern,seq,rss,count", PATH_COUNTS)
def test_reference_path_counts_linear(pattern, seq, rss, count):
    e = Emul(pattern, PAR, BIG, BIG, 0.0, 1.0, DBG_FLAGS)
    x = np.zeros(e.n_param)
    x[-2:] = 1.0
    r = e.train_seq(x, po.encode_seq(seq), np.ones(len(seq) + 1, dtype=np.uint8), fix_rss=rss, linear=0)
    assert np.exp(r["Zo"]) == pytest.approx(count, rel=1e-13)


@pytest.mark.parametrize("seq,rss,counts", EMISSION_COUNTS)
def test_reference_emission_counts_linear(seq, rss, counts):
    e = Emul(".", PAR, BIG, BIG, 0.0, 1.0, DBG_FLAGS)
    x = np.zeros(e.n_param)
    x[-2:] = 1.0
    r = e.train_seq(x, po.encode_seq(seq), np.ones(len(seq) + 1, dtype=np.uint8), fix_rss=rss, linear=0)
    np.testing.assert_allclose(r["ENo"] * np.exp(r["Zo"]), [v for row in counts for v in row], rtol=1e-12, atol=1e-12)


SCAN = [("0.model", "0.fq"), ("1.model", "0.fq"), ("3.model", "0.fq"), ("tiny_a.model", "tiny.fq"),
        ("syn_b.model", "syn_L100_n3.fq"), ("syn_sm.model", "syn_L40_n3.fq"), ("syn_c12.model", "syn_L100_n3.fq"),
        ("trna_a.model", "positive_head6.fq")]


@pytest.mark.parametrize("model,fq", SCAN)
def test_scan_sequence_matches_oracle(model, fq):
    o, e, x = model_pair(model)
    for rid, seq, qual in po.read_fastq(gpath(fq)):
        a = o.scan_seq(seq, qual)
        b = e.scan_seq(x, seq, qual)
        assert (a["Ys"], a["Ye"]) == (b["Ys"], b["Ye"])
        for k in ("ZL", "ZeL", "PyNL"):
            assert_log_close(b[k], a[k], rtol=1e-11, what=k)
        for k in ("start", "end", "inner"):
            assert_log_close(b[k], a[k], rtol=1e-9, atol=1e-9, what=k)
        assert b["exist_prob"] == pytest.approx(a["exist_prob"], rel=1e-10)
        assert list(a["psihat"]) == list(b["psihat"])
        assert a["rss"] == b["rss"]
        np.testing.assert_allclose(b["EN"], a["EN"], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("model,fq", SCAN)
def test_scan_with_linear_sum_passes_matches_oracle(model, fq):
    """K4 / K5 of the scan in the scaled-linear semiring (lin_rules.h with the start constraint and the SCAN / END
    statistics), K6 in log space: what elemdp_scan's batch pipeline runs."""
    o, e, x = model_pair(model)
    for rid, seq, qual in po.read_fastq(gpath(fq)):
        a = o.scan_seq(seq, qual)
        b = e.scan_seq(x, seq, qual, linear=True)
        assert (a["Ys"], a["Ye"]) == (b["Ys"], b["Ye"])
        for k in ("ZL", "ZeL", "PyNL"):
            assert_log_close(b[k], a[k], rtol=1e-10, atol=1e-10, what=k)
        for k in ("start", "end", "inner"):
            assert_log_close(b[k], a[k], rtol=1e-9, atol=1e-9, what=k)
        assert b["exist_prob"] == pytest.approx(a["exist_prob"], rel=1e-10)
        assert list(a["psihat"]) == list(b["psihat"])
        assert a["rss"] == b["rss"]
        np.testing.assert_allclose(b["EN"], a["EN"], rtol=1e-9, atol=1e-11)


def test_plan_enumeration_from_the_end_major_mask_is_identical():
    """The plan builder (k_plan_cells / k_plan_fill) enumerates the interior-loop items with enum_interior_by_end (bit runs
    of the pair mask indexed by end position) instead of the reference-shaped double loop enum_interior: every plan the
    emulation builds is enumerated both ways; items, order, tsc and inside-set flags must be identical -- with and without
    the BPP filter, with a small C, and with C larger than a mask word."""
    import ctypes
    from rnaelem_amd import api, synth
    from tests.emul import pyemul
    L = pyemul.lib()
    L.emu_enum_checked.restype = L.emu_enum_mismatches.restype = ctypes.c_long
    before = L.emu_enum_checked()
    seqs, quals = synth.synth_batch(3, 120)
    x = api.Engine("((.*.))").initial_params(1.0)
    for kw in (dict(), dict(min_bpp=0.0), dict(max_iloop=7), dict(max_span=45, max_iloop=40)):
        e = Emul("((.*.))", PAR, **kw)
        for s, q in zip(seqs, quals):
            e.train_seq(x, s, q)
    assert L.emu_enum_checked() >= before + 12
    assert L.emu_enum_mismatches() == 0


# ---- static pruning of the transition lists (Automaton::flatten(prune), the engine's default) -------------------
PLANES = "PEMB12LO"


def useful_mask(e):
    """[8][S] booleans: (plane, state) pairs that can occur in a complete parse (Automaton::liveness)."""
    d = e.describe()
    m = np.zeros((8, e.S), dtype=bool)
    for k, ids in enumerate(d["useful"]):
        m[k, ids] = True
    return m


def assert_pruned_tables(got, want, mask7, what):
    """pruned tables: equal to the oracle's on the useful (plane, state) pairs (the other entries never reach a terminal: the
    pruned rules leave some of them at log 0, and nothing reads them)"""
    assert_log_close(got[:, :, mask7], want[:, :, mask7], rtol=1e-10, what=what)


@pytest.mark.parametrize("pattern", sorted(HMM))
def test_liveness_is_consistent(pattern):
    e = Emul(pattern, PAR)
    d = e.describe()
    for k in range(8):
        assert set(d["useful"][k]) <= set(d["inside_live"][k]), (pattern, PLANES[k])
    s00 = d["state"].index([0, 0])
    assert s00 in d["useful"][7] and s00 in d["useful"][6]   # background: O and L of (0,0) are always in a parse


@pytest.mark.parametrize("model,fq", CASES)
@pytest.mark.parametrize("linear", [None, 0, 1])
def test_pruned_train_rules_match_oracle(model, fq, linear):
    o, e, x = model_pair(model)
    e.set_prune(True)
    um = useful_mask(e)
    full = fq in ("tiny.fq", "0.fq", "syn_L40_n3.fq")
    for rid, seq, qual in po.read_fastq(gpath(fq)):
        a = o.train_seq(seq, qual, tables=full)
        b = e.train_seq(x, seq, qual, tables=full, linear=linear)
        for k in ("Zo", "Zari", "Znasi"):
            assert_log_close(b[k], a[k], rtol=1e-11, what=k)
        assert_log_close(b["inside_o"][:, um[7]], a["inside_o"][:, um[7]], rtol=1e-10, what="inside_o")
        if full:
            assert_pruned_tables(b["inside"], a["inside"], um[:7], "inside table")
        if a["skipped"] or (linear == 1 and not np.isfinite(a["Znasi"])):
            continue
        assert b["skipped"] == 0
        assert b["f"] == pytest.approx(a["f"], rel=1e-10, abs=1e-12)
        for k in ("ENo", "ENx", "EHo", "EHx"):
            np.testing.assert_allclose(b[k], a[k], rtol=1e-9, atol=1e-11, err_msg=k)
        if linear != 1:
            assert_log_close(b["outside_o"], a["outside_o"], rtol=1e-10, what="outside_o")
            if full:
                assert_log_close(b["outside"], a["outside"], rtol=1e-10, what="outside table")


@pytest.mark.parametrize("model,fq", SCAN)
@pytest.mark.parametrize("linear", [False, True])
def test_pruned_scan_matches_oracle(model, fq, linear):
    o, e, x = model_pair(model)
    e.set_prune(True)
    for rid, seq, qual in po.read_fastq(gpath(fq)):
        a = o.scan_seq(seq, qual)
        b = e.scan_seq(x, seq, qual, linear=linear)
        assert (a["Ys"], a["Ye"]) == (b["Ys"], b["Ye"])
        for k in ("ZL", "ZeL", "PyNL"):
            assert_log_close(b[k], a[k], rtol=1e-10, atol=1e-10, what=k)
        for k in ("start", "end", "inner"):
            assert_log_close(b[k], a[k], rtol=1e-9, atol=1e-9, what=k)
        assert b["exist_prob"] == pytest.approx(a["exist_prob"], rel=1e-10)
        assert list(a["psihat"]) == list(b["psihat"])
        assert a["rss"] == b["rss"]
        np.testing.assert_allclose(b["EN"], a["EN"], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("pattern,seq,rss,count", PATH_COUNTS)
def test_reference_path_counts_pruned(pattern, seq, rss, count):
    e = Emul(pattern, PAR, BIG, BIG, 0.0, 1.0, DBG_FLAGS)
    e.set_prune(True)
    x = np.zeros(e.n_param)
    x[-2:] = 1.0
    for linear in (None, 0):
        r = e.train_seq(x, po.encode_seq(seq), np.ones(len(seq) + 1, dtype=np.uint8), fix_rss=rss, linear=linear)
        assert np.exp(r["Zo"]) == pytest.approx(count, rel=1e-13)

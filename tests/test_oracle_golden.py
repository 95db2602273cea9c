"""Pins the CPU oracle (oracle/elem_oracle.cpp) against the reference.

Sources of truth, all committed under tests/golden/ (generator: tests/golden/make_golden.py):
  * the reference's own known-answer tests: PATH_COUNT / EMISSION_COUNT (RNAelem-test/test.cpp:88-203),
    BPP_RNAFOLD (test-exact.cpp:86-138, RNAfold 2.3.1 dot plot), FastqIO (test-exact.cpp:38-52);
  * outputs of the compiled reference (oracle/_ref) on stated inputs: automata, energy tables,
    fn/gr at 17 digits, per-sequence partition functions and expected counts, scan records.
These tests run on CPU only (-m "not gpu").
"""
import hashlib
import struct

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.util import arr, assert_log_close, gload, gpath

HMM = gload("hmm.json")


@pytest.mark.parametrize("pattern", sorted(HMM))
def test_pattern_automaton_matches_reference(pattern):
    ref = HMM[pattern]
    o = po.make_oracle(pattern)
    got = o.hmm()
    for k in ("reg_pattern", "M", "S", "node", "theta_id", "theta_sizes", "state", "loop_state", "reachable", "right",
              "left", "pair", "loop_loop"):
        assert got[k] == ref[k], (pattern, k)
    # initial theta = uniform log-probabilities (profile_hmm.hpp:286-313)
    x = o.get_params()
    flat = [v for row in ref["theta"] for v in row]
    np.testing.assert_allclose(x[:-2], flat, rtol=0, atol=1e-15)


def test_automaton_sizes_for_all_preset_patterns():
    for pattern, ref in gload("hmm_sizes.json").items():
        h = po.make_oracle(pattern).hmm()
        got = [h["M"], h["S"], len(h["loop_state"]), len(h["loop_loop"]), sum(map(len, h["right"])),
               sum(map(len, h["left"])), sum(map(len, h["pair"]))]
        assert got == ref, pattern


@pytest.mark.parametrize("tag,fname", [("T2004", "turner2004.elempar"), ("A2007", "andronescu2007.elempar")])
def test_energy_tables_bit_exact(tag, fname):
    ref = gload("energy_%s.json" % tag)
    import os
    par = open(os.path.join(os.path.dirname(po.DEFAULT_PAR), fname)).read()
    o = po.make_oracle("(.)", par_text=par)
    for name, val in ref.items():
        if name in ("triloops", "tetraloops", "hexaloops"):
            continue
        if name == "int_22_acgu":
            t = o.energy_table("int_22").reshape(8, 8, 5, 5, 5, 5)[1:7, 1:7, 1:5, 1:5, 1:5, 1:5].ravel()
            full = o.energy_table("int_22").reshape(8, 8, 5, 5, 5, 5)
            # every entry the reference never reads from the file is -inf here (documented deviation:
            # the reference leaves part of it uninitialised, energy_param.hpp:604-608)
            mask = np.ones(full.shape, bool)
            mask[1:7, 1:7, 1:5, 1:5, 1:5, 1:5] = False
            assert np.all(np.isneginf(full[mask]))
        elif isinstance(val, (dict, list)):
            t = o.energy_table(name)
        else:
            t = o.energy_table(name)
            assert t[0] == val, name
            continue
        if isinstance(val, dict):
            raw = b"".join(struct.pack("<d", v) for v in t)
            assert len(t) == val["n"], name
            assert hashlib.sha256(raw).hexdigest() == val["sha256"], name
        elif name.startswith("mismatch_"):
            # pair-type row 0 is never indexed by a real pair; in the reference it holds the spill-over of the
            # neighbouring table's out-of-bounds 7th block (energy_param.hpp:556-566 read 8 blocks into
            # [7][5][5] arrays), so only rows 1..6 are compared.
            assert np.array_equal(t.reshape(7, 25)[1:], arr(val).reshape(7, 25)[1:]), name
        else:
            assert np.array_equal(t, arr(val)), name


# (pattern, seq, rss) -> exp(Z) as asserted by RNAelem-test/test.cpp:101-176
PATH_COUNTS = [
    (".", "A", ".", 2), (".", "AA", "..", 4), (".", "CAAAG", "(...)", 7), (".", "ACAAAGA", ".(...).", 9),
    (".", "ACACAAAGGA", ".(.(...)).", 10), (".", "ACACAGACAGAAGA", ".(.(.).(.)..).", 10), (".", "CACAGAG", "(.(.).)", 4),
    ("(.)", "CAAAG", "(...)", 2), ("(.)", "CCAAAGG", "((...))", 3), ("(.*)", "CAAAG", "(...)", 4),
    ("(.*)", "CCAAAGG", "((...))", 7), (".*.", "AA", "..", 2), (".*.", "CAAAG", "(...)", 6),
    ("(.).(.)", "CAGACAG", "(.).(.)", 2), ("(.).(.)", "CCAGACAGG", "((.).(.))", 2), ("(.)*(.)", "CAGCAG", "(.)(.)", 2),
    ("(.)*(.)", "CCAGCAGG", "((.)(.))", 2),
]
# RNAelem-test/test.cpp:191-202
EMISSION_COUNTS = [
    ("A", ".", [[1, 0, 0, 0], [1, 0, 0, 0]]), ("CAG", "(.)", [[1, 2, 2, 0], [1, 0, 0, 0]]),
    ("CACGG", "(...)", [[4, 10, 11, 0], [3, 4, 3, 0]]), ("CAGAU", "(.)..", [[7, 5, 5, 3], [3, 0, 0, 2]]),
]
DBG = po.NO_ENE | po.DBG_NO_THETA | po.DBG_FIX_RSS | po.DBG_NO_TURN
BIG = 2 ** 31 - 1


def _dbg_oracle(pattern):
    # RNAelemDPTest fixture (test.cpp:81-85): no_ene, tau=1, lambda={1,1}, min_bpp=0, W=C=large
    return po.make_oracle(pattern, BIG, BIG, min_bpp=0.0, tau=1.0, flags=DBG, lam=(1.0, 1.0))


@pytest.mark.parametrize("pattern,seq,rss,count", PATH_COUNTS)
def test_reference_path_count_cases(pattern, seq, rss, count):
    o = _dbg_oracle(pattern)
    r = o.train_seq(po.encode_seq(seq), np.ones(len(seq) + 1, dtype=np.uint8), fix_rss=rss)
    assert np.exp(r["Zo"]) == pytest.approx(count, rel=4e-16 * 4)       # EXPECT_DOUBLE_EQ = 4 ulp
    assert np.exp(r["outside_o"][0, 0]) == pytest.approx(count, rel=4e-16 * 4)


@pytest.mark.parametrize("seq,rss,counts", EMISSION_COUNTS)
def test_reference_emission_count_cases(seq, rss, counts):
    # motif_test.hpp:22-31 runs the outside pass with Z = oneL, i.e. un-normalised counts
    o = _dbg_oracle(".")
    r = o.train_seq(po.encode_seq(seq), np.ones(len(seq) + 1, dtype=np.uint8), fix_rss=rss)
    got = r["ENo"] * np.exp(r["Zo"])
    assert ["%g" % v for v in got] == ["%g" % v for row in counts for v in row]


def test_path_count_golden_from_reference_debug_build():
    for c in gload("pathcount.json"):
        o = _dbg_oracle(c["pattern"])
        r = o.train_seq(po.encode_seq(c["seq"]), np.ones(len(c["seq"]) + 1, dtype=np.uint8), fix_rss=c["rss"])
        assert np.exp(r["Zo"]) == pytest.approx(c["Z"], rel=1e-14)
        assert np.exp(r["outside_o"][0, 0]) == pytest.approx(c["Zout"], rel=1e-14)
        np.testing.assert_allclose(r["ENo"] * np.exp(r["Zo"]), [v for row in c["ENo"] for v in row], rtol=1e-13, atol=1e-13)


def test_fastq_reader_contract():
    recs = po.read_fastq(gpath("0.fq"))           # test-exact.cpp:38-52
    assert len(recs) == 2 and recs[1][0] == "@1"
    assert all(len(q) == len(s) + 1 for _, s, q in recs)
    assert recs[0][2][-1] == 0 and recs[1][2][-1] == 5   # '!' => has motif, '&' => no motif


def test_bpp_against_rnafold_dotplot():
    """BPP_RNAFOLD (test-exact.cpp:86-138): ln BPP vs RNAfold -p --maxBPspan=50, 1e-5 abs."""
    (rid, seq, qual), = po.read_fastq(gpath("1.fq"))
    o = po.make_oracle("(.)", 50, 30, min_bpp=0.0)
    ln, kept, eff, lnz = o.bpp(seq)
    n = 0
    for line in open(gpath("rnafold_1_0_ubox.txt")):
        if line.startswith("#"):
            continue
        i, j, sp = line.split()
        i, j, sp = int(i), int(j), float(sp)
        # the reference test stores the value at [i-1][j-i] and reads it back as lnBPP(i-1, j)
        assert ln[i - 1, j - (i - 1)] == pytest.approx(2 * np.log(sp), abs=1e-5), (i, j)
        n += 1
    assert n == 1146   # every `i j sqrt(p) ubox` data line of 1.0.ps


@pytest.mark.parametrize("name,fq", [("bpp_1fq.json", "1.fq"), ("bpp_syn_L100.json", "syn_L100_n3.fq")])
def test_bpp_filter_against_reference(name, fq):
    ref = gload(name)
    recs = po.read_fastq(gpath(fq))
    o = po.make_oracle("(.)", ref["W"], ref["C"], min_bpp=ref["min_bpp"])
    for (rid, seq, qual), r in zip(recs, ref["seqs"]):
        ln, kept, eff, lnz = o.bpp(seq)
        assert eff == r["bpp_eff"]
        assert lnz == pytest.approx(r["lnZ"], rel=1e-13)
        got = sorted((int(i), int(i + d)) for i, d in np.argwhere(kept))
        assert got == sorted((a, b) for a, b in r["kept"])
        for i, j, v in r["lnbpp"]:
            assert ln[i, j - i] == pytest.approx(v, rel=1e-11, abs=1e-11)


EVAL = gload("eval.json")


@pytest.mark.parametrize("case", EVAL, ids=["%s-%s" % (c["model"], c["fq"]) for c in EVAL])
def test_fn_gr_against_reference(case):
    if case["fq"] == "positive.fq":
        pytest.skip("76-sequence case is covered by test_fn_gr_config_a (slow)")
    o, x = po.oracle_from_model(gpath(case["model"]))
    np.testing.assert_allclose(x, case["x"], rtol=0, atol=0)
    recs = po.read_fastq(gpath(case["fq"]))
    fn, gr, eff, nsk = o.train_eval(x, [s for _, s, _ in recs], [q for _, _, q in recs])
    assert fn == pytest.approx(case["fn"], rel=1e-12, abs=1e-12)
    np.testing.assert_allclose(gr, arr(case["gr"]), rtol=1e-10, atol=1e-11)
    assert eff == pytest.approx(case["sum_eff"], rel=1e-14)


EVAL_LIK = gload("eval_lik.json")


@pytest.mark.parametrize("case", EVAL_LIK, ids=["%s-%s" % (c["model"], c["fq"]) for c in EVAL_LIK])
def test_fn_gr_lik_ratio_against_reference(case):
    """--lik-ratio objective (motif_trainer.hpp:156-202): sequences without motif contribute Z(ari) - Z(ari,nasi)."""
    o, x = po.oracle_from_model(gpath(case["model"]), extra_flags=po.LIK_RATIO)
    assert np.array_equal(x, arr(case["x"]))
    recs = po.read_fastq(gpath(case["fq"]))
    fn, gr, eff, nsk = o.train_eval(x, [s for _, s, _ in recs], [q for _, _, q in recs])
    assert fn == pytest.approx(case["fn"], rel=1e-12, abs=1e-14)
    np.testing.assert_allclose(gr, arr(case["gr"]), rtol=1e-9, atol=1e-12)


def test_fn_gr_config_a():
    """BASELINE config A: material/positive.fa x '(.....)' (76 tRNA), x0 and a perturbed point."""
    recs = po.read_fastq(gpath("positive.fq"))
    assert len(recs) == 76
    for case in EVAL:
        if case["fq"] != "positive.fq":
            continue
        o, x = po.oracle_from_model(gpath(case["model"]))
        fn, gr, eff, nsk = o.train_eval(x, [s for _, s, _ in recs], [q for _, _, q in recs], n_threads=8)
        assert fn == pytest.approx(case["fn"], rel=1e-11)
        np.testing.assert_allclose(gr, arr(case["gr"]), rtol=1e-9, atol=1e-10)
        assert eff == pytest.approx(case["sum_eff"], rel=1e-12)
        if case["model"] == "trna_x0.model":   # y = 36.2538, |gr|^2 = 13139.3, BP = 0.299584
            assert fn == pytest.approx(36.2538, abs=1e-4)
            assert np.sum(gr ** 2) == pytest.approx(13139.3, rel=1e-5)
            assert eff / 76 == pytest.approx(0.299584, abs=1e-6)


DP = gload("dp.json")


@pytest.mark.parametrize("case", DP, ids=["%s-%s" % (c["model"], c["fq"]) for c in DP])
def test_per_sequence_dp_against_reference(case):
    o, x = po.oracle_from_model(gpath(case["model"]))
    assert o.S == case["S"] and o.M == case["M"]
    recs = po.read_fastq(gpath(case["fq"]))
    for (rid, seq, qual), r in zip(recs, case["seqs"]):
        full = "inside" in r
        g = o.train_seq(seq, qual, tables=full)
        for k in ("Zo", "Zari", "Znasi"):
            assert_log_close(g[k], r[k], rtol=1e-13, what=k)
        assert_log_close(g["inside_o"], arr(r["inside_o"]), rtol=1e-12, what="inside_o")
        if "f" not in r:
            assert g["skipped"]
            continue
        assert g["f"] == pytest.approx(r["f"], rel=1e-12, abs=1e-13)
        assert_log_close(g["outside_o"], arr(r["outside_o_full"]), rtol=1e-12, what="outside_o")
        np.testing.assert_allclose(g["ENo"], [v for row in r["ENo"] for v in row], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(g["ENx"], [v for row in r["ENx"] for v in row], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(g["EHo"], r["EHo"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(g["EHx"], r["EHx"], rtol=1e-10, atol=1e-12)
        if full:
            for key in ("inside", "outside"):
                dense = np.full(g[key].shape, -np.inf)
                for i, d, e, s, v in r[key]:
                    dense[i, d, e, s] = v
                assert_log_close(g[key], dense, rtol=1e-12, what=key)


SCAN = gload("scan.json")


@pytest.mark.parametrize("case", SCAN, ids=["%s-%s" % (c["model"], c["fq"]) for c in SCAN])
def test_scan_records_against_reference_binary(case):
    """`RNAelem scan` prints 6 significant digits (util.hpp:98-105) -> compare at that precision."""
    o, x = po.oracle_from_model(gpath(case["model"]))
    recs = po.read_fastq(gpath(case["fq"]))
    byid = {r["id"]: r for r in case["records"]}
    nodes = o.hmm()["node"]
    for rid, seq, qual in recs:
        r = byid[rid]
        g = o.scan_seq(seq, qual)
        assert (g["Ys"], g["Ye"]) == (r["Ys"], r["Ye"])
        for k in ("start", "end", "inner"):
            ref = arr(r[k])
            got = g[k]
            assert np.array_equal(np.isneginf(ref), np.isneginf(got)), k
            m = ~np.isneginf(ref)
            np.testing.assert_allclose(got[m], ref[m], rtol=2e-5, atol=1e-300)
        assert g["exist_prob"] == pytest.approx(r["exist_prob"], rel=2e-5)
        assert list(g["psihat"]) == r["psihat"]
        assert g["rss"] == r["rss"]
        mot = "".join(" " if (h == 0 or h == o.M - 1) else nodes[h] for h in g["psihat"])
        assert mot == r["mot"]


@pytest.mark.parametrize("model", ["0.model", "1.model", "2.model", "3.model", "tiny_a.model"])
def test_oracle_gradient_by_central_difference(model):
    """What the reference's MACHINE_DIFF_GR (test-exact.cpp:54-84) intends: d=1e-5, 1e-6 abs."""
    o, x = po.oracle_from_model(gpath(model))
    recs = po.read_fastq(gpath("tiny.fq" if model.startswith("tiny") else "0.fq"))
    S, Q = [s for _, s, _ in recs], [q for _, _, q in recs]
    fn, gr, _, _ = o.train_eval(x, S, Q)
    d = 1e-5
    for i in range(len(x)):
        xp, xm = x.copy(), x.copy()
        xp[i] += d / 2
        xm[i] -= d / 2
        fp = o.train_eval(xp, S, Q)[0]
        fm = o.train_eval(xm, S, Q)[0]
        assert gr[i] == pytest.approx((fp - fm) / d, abs=1e-6), (model, i)

"""Parity of the HIP path (through the C ABI of libelemdp.so) with the oracle and the golden vectors.

Every test here needs a real MI355X (`-m gpu`).  Tolerances: |d log Z| <= 1e-9 *
max(1,|log Z|); fn rel 1e-9; gr abs 1e-7 + rel 1e-7; log posteriors abs 1e-6 with identical -inf
pattern; argmax positions and Viterbi strings exact.
"""
import numpy as np
import pytest

from oracle import pyoracle as po
from rnaelem_amd import api, io, synth
from tests.util import arr, assert_log_close, gload, gpath

pytestmark = pytest.mark.gpu

PAR = "~T2004~"


def load(model, fq):
    m = io.read_model(gpath(model))
    eng = io.engine_from_model(m)
    recs = io.read_fastq(gpath(fq))
    return m, eng, recs


def oracle_for(model):
    return po.oracle_from_model(gpath(model))


def test_library_reports_native_kernel():
    eng = api.Engine("(.)")
    assert eng.kernel_name().startswith("k")
    assert eng.describe()["S"] == eng.n_state


def test_initial_params_match_reference_x0():
    for c in gload("eval.json"):
        if c["model"] in ("trna_x0.model", "syn_x0.model"):
            m = io.read_model(gpath(c["model"]))
            eng = io.engine_from_model(m)
            np.testing.assert_allclose(eng.initial_params(0.0), c["x"], rtol=0, atol=0)


@pytest.mark.parametrize("fq,W", [("1.fq", 50), ("syn_L100_n3.fq", 50), ("0.fq", 20)])
def test_bpp_filter(fq, W):
    """K1 on the GPU vs the oracle: ln BPP of every candidate, kept set, bpp_eff."""
    eng = api.Engine("(.)", PAR, W, 30, 1e-4)
    eng.set_option("keep_lnbpp", 1)
    recs = io.read_fastq(gpath(fq))
    eng.load_batch([s for _, s, _ in recs], [q for _, _, q in recs])
    o = po.make_oracle("(.)", W, 30, min_bpp=1e-4)
    eff = eng.bpp_eff()
    for k, (rid, seq, qual) in enumerate(recs):
        ln_o, kept_o, eff_o, lnz_o = o.bpp(seq)
        kept, ln = eng.pairs(k, with_lnbpp=True)
        assert np.array_equal(kept, kept_o)
        assert eff[k] == eff_o
        assert_log_close(ln, ln_o, rtol=1e-9, atol=1e-9, what="lnbpp")


def test_bpp_against_rnafold_dotplot():
    """The reference's BPP_RNAFOLD known answer (test-exact.cpp:86-138), on the GPU, 1e-5 abs."""
    (rid, seq, qual), = io.read_fastq(gpath("1.fq"))
    eng = api.Engine("(.)", PAR, 50, 30, 1e-300)   # threshold so low that nothing is filtered
    eng.set_option("keep_lnbpp", 1)
    eng.load_batch([seq], [qual])
    kept, ln = eng.pairs(0, with_lnbpp=True)
    n = 0
    for line in open(gpath("rnafold_1_0_ubox.txt")):
        if line.startswith("#"):
            continue
        i, j, sp = line.split()
        i, j, sp = int(i), int(j), float(sp)
        assert ln[i - 1, j - (i - 1)] == pytest.approx(2 * np.log(sp), abs=1e-5), (i, j)
        n += 1
    assert n == 1146


TABLE_CASES = [("tiny_a.model", "tiny.fq"), ("tiny_ne.model", "tiny.fq"), ("0.model", "0.fq"), ("1.model", "0.fq"),
               ("3.model", "0.fq"), ("syn_b.model", "syn_L40_n3.fq"), ("syn_c12.model", "syn_L40_n3.fq")]


def useful_masks(eng):
    """[8][S] booleans: (plane, state) pairs that can occur in a complete parse (Automaton::liveness via elemdp_describe)"""
    um = np.zeros((8, eng.n_state), dtype=bool)
    for k, ids in enumerate(eng.describe()["useful"]):
        um[k, ids] = True
    return um


@pytest.mark.parametrize("model,fq", TABLE_CASES)
@pytest.mark.parametrize("prune", [0, 1])
def test_tables_of_single_sequences(model, fq, prune):
    """Full inside / outside tables, exterior chains and expected counts of one sequence at a time.  prune = 0: the
    complete transition lists, inside tables equal to the oracle's everywhere; prune = 1 (the default): lists pruned to the
    transitions of complete parses, inside tables equal on the useful (plane, state) pairs (the others are never read);
    outside tables and all statistics identical in both modes."""
    m, eng, recs = load(model, fq)
    eng.set_option("prune", prune)
    um = useful_masks(eng)
    o, x = oracle_for(model)
    for rid, seq, qual in recs:
        a = o.train_seq(seq, qual, tables=True)
        eng.set_option("first_pass_only", 1)
        eng.load_batch([seq], [qual])
        eng.train_eval(x)
        t = eng.debug_tables()
        if prune:
            assert_log_close(t["inside"][:, :, um[:7]], a["inside"][:, :, um[:7]], rtol=1e-10, what="inside")
            assert_log_close(t["inside_o"][:, um[7]], a["inside_o"][:, um[7]], rtol=1e-10, what="inside_o")
        else:
            assert_log_close(t["inside"], a["inside"], rtol=1e-10, what="inside")
            assert_log_close(t["inside_o"], a["inside_o"], rtol=1e-10, what="inside_o")
        if not a["skipped"]:
            assert_log_close(t["outside"], a["outside"], rtol=1e-10, what="outside")
            assert_log_close(t["outside_o"], a["outside_o"], rtol=1e-10, what="outside_o")
            np.testing.assert_allclose(t["ENo"], a["ENo"], rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(t["EHo"], a["EHo"], rtol=1e-8, atol=1e-10)
        eng.set_option("first_pass_only", 0)
        eng.train_eval(x)
        t = eng.debug_tables()
        st = eng.seq_stats()[0]
        for k, name in enumerate(("Zo", "Zari", "Znasi")):
            assert_log_close(st[k], a[name], rtol=1e-10, what=name)
        assert st[4] == a["skipped"]
        if not a["skipped"]:
            assert st[3] == pytest.approx(a["f"], rel=1e-9, abs=1e-10)
            np.testing.assert_allclose(t["ENx"], a["ENx"], rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(t["EHx"], a["EHx"], rtol=1e-8, atol=1e-10)


EVAL = gload("eval.json")


@pytest.mark.parametrize("case", EVAL, ids=["%s-%s" % (c["model"], c["fq"]) for c in EVAL])
def test_fn_gr_against_reference_golden(case):
    """elemdp_train_eval vs fn / gr printed by the compiled reference at 17 digits."""
    m, eng, recs = load(case["model"], case["fq"])
    np.testing.assert_allclose(m["x"], case["x"], rtol=0, atol=0)
    eng.load_batch([s for _, s, _ in recs], [q for _, _, q in recs])
    fn, gr, eff, nsk = eng.train_eval(m["x"])
    assert fn == pytest.approx(case["fn"], rel=1e-9, abs=1e-9)
    np.testing.assert_allclose(gr, arr(case["gr"]), rtol=1e-7, atol=1e-7)
    assert eff == pytest.approx(case["sum_eff"], rel=1e-12)
    assert nsk == 0
    # the partial / finish pair used by the multi-GPU path gives the same numbers
    part = eng.train_partial(m["x"])
    fn2, gr2, eff2, nsk2 = eng.train_finish(part)
    # (same numbers up to the order of the LDS atomics that accumulate the expected counts)
    assert fn2 == pytest.approx(fn, rel=1e-13) and eff2 == eff
    np.testing.assert_allclose(gr2, gr, rtol=1e-11, atol=1e-12)


from tests.test_oracle_golden import EMISSION_COUNTS, PATH_COUNTS  # noqa: E402

BIG = 2 ** 30


@pytest.mark.parametrize("pattern,seq,rss,count", PATH_COUNTS)
def test_reference_path_count_cases(pattern, seq, rss, count):
    """PATH_COUNT_CASES (RNAelem-test/test.cpp:88-177): number of motif alignments on a fixed structure."""
    eng = api.Engine(pattern, PAR, BIG, BIG, 0.0, 1.0, api.NO_ENERGY | api.DBG_FIX_RSS | api.DBG_NO_TURN)
    x = np.zeros(eng.n_param)
    x[-2:] = 1.0
    eng.set_option("first_pass_only", 1)
    eng.load_batch([io.encode_seq(seq)], [np.ones(len(seq) + 1, dtype=np.uint8)], fix_rss=[rss])
    eng.train_eval(x)
    t = eng.debug_tables()
    Z = eng.seq_stats()[0][0]
    assert np.exp(Z) == pytest.approx(count, rel=1e-12)
    assert np.exp(t["outside_o"][0, 0]) == pytest.approx(count, rel=1e-12)


@pytest.mark.parametrize("seq,rss,counts", EMISSION_COUNTS)
def test_reference_emission_count_cases(seq, rss, counts):
    eng = api.Engine(".", PAR, BIG, BIG, 0.0, 1.0, api.NO_ENERGY | api.DBG_FIX_RSS | api.DBG_NO_TURN)
    x = np.zeros(eng.n_param)
    x[-2:] = 1.0
    eng.set_option("first_pass_only", 1)
    eng.load_batch([io.encode_seq(seq)], [np.ones(len(seq) + 1, dtype=np.uint8)], fix_rss=[rss])
    eng.train_eval(x)
    t = eng.debug_tables()
    Z = eng.seq_stats()[0][0]
    np.testing.assert_allclose(t["ENo"] * np.exp(Z), [v for row in counts for v in row], rtol=1e-11, atol=1e-11)


def test_skipped_sequence_and_no_motif_record():
    """A pattern longer than the sequence makes Z(ari) = log 0 -> the sequence is skipped
    (motif_trainer.hpp:211-215); a record whose last quality is not '!' uses the nasi mask."""
    eng = api.Engine("(.........)", PAR, 50, 30, 1e-4)
    o = po.make_oracle("(.........)", 50, 30, min_bpp=1e-4)
    seqs = [io.encode_seq("GGGAAAUCCC"), io.encode_seq("GGGGAAAUCCCCAAGGGAAACCCAAAGGCAGCAAAAGCUGCC")]
    quals = [np.r_[np.full(10, 10), 0].astype(np.uint8), np.r_[np.full(42, 10), 5].astype(np.uint8)]
    x = eng.initial_params(0.5)
    o.set_params(x)
    eng.load_batch(seqs, quals)
    fn, gr, eff, nsk = eng.train_eval(x)
    fo, go, eo, no = o.train_eval(x, seqs, quals)
    assert nsk == no == 1
    assert fn == pytest.approx(fo, rel=1e-9)
    np.testing.assert_allclose(gr, go, rtol=1e-7, atol=1e-7)
    assert eff == pytest.approx(eo, rel=1e-12)


SCAN = gload("scan.json")


@pytest.mark.parametrize("case", SCAN, ids=["%s-%s" % (c["model"], c["fq"]) for c in SCAN])
def test_scan_against_oracle_and_reference_records(case):
    m, eng, recs = load(case["model"], case["fq"])
    o, x = oracle_for(case["model"])
    eng.load_batch([s for _, s, _ in recs], [q for _, _, q in recs])
    got, en = eng.scan(m["x"])
    byid = {r["id"]: r for r in case["records"]}
    nodes = eng.describe()["node"]
    en_o = np.zeros(eng.n_param - 2)
    for (rid, seq, qual), g in zip(recs, got):
        a = o.scan_seq(seq, qual)
        en_o += a["EN"]
        assert (g["Ys"], g["Ye"]) == (a["Ys"], a["Ye"])
        for k in ("start", "end", "inner"):
            assert_log_close(g[k], a[k], rtol=1e-8, atol=1e-6, what=k)
        assert g["exist_prob"] == pytest.approx(a["exist_prob"], rel=1e-9)
        assert list(g["psihat"]) == list(a["psihat"])
        assert g["rss"] == a["rss"]
        # and the record text equals what `RNAelem scan` printed (6 significant digits)
        r = byid[rid]
        assert (g["Ys"], g["Ye"]) == (r["Ys"], r["Ye"])
        assert g["rss"] == r["rss"] and list(g["psihat"]) == r["psihat"]
        txt = io.scan_record(rid, seq, g, nodes)
        assert ("mot: " + r["mot"]) in txt
    np.testing.assert_allclose(en, en_o, rtol=1e-8, atol=1e-10)


def test_gradient_by_central_difference():
    """The check the reference's MACHINE_DIFF_GR intends (test-exact.cpp:54-84): d = 1e-5, 1e-6 abs."""
    for model, fq in (("0.model", "0.fq"), ("1.model", "0.fq"), ("3.model", "0.fq")):
        m, eng, recs = load(model, fq)
        eng.load_batch([s for _, s, _ in recs], [q for _, _, q in recs])
        x = m["x"]
        fn, gr, _, _ = eng.train_eval(x)
        d = 1e-5
        for i in range(len(x)):
            xp, xm = x.copy(), x.copy()
            xp[i] += d / 2
            xm[i] -= d / 2
            fp = eng.train_eval(xp)[0]
            fm = eng.train_eval(xm)[0]
            assert gr[i] == pytest.approx((fp - fm) / d, abs=1e-6), (model, i)


def test_ragged_batch_and_batch_linearity():
    """Mixed lengths in one batch; fn / gr of a batch equal the sums over any split of it."""
    m = io.read_model(gpath("syn_b.model"))
    eng = io.engine_from_model(m)
    seqs, quals = [], []
    for L, n in ((30, 3), (200, 2), (75, 4), (8, 2), (120, 3)):
        s, q = synth.synth_batch(n, L, seed=1000 + L)
        seqs += s
        quals += q
    quals[1][-1] = 7
    eng.load_batch(seqs, quals)
    fn, gr, eff, nsk = eng.train_eval(m["x"])
    stats = eng.seq_stats()
    assert fn == pytest.approx(stats[:, 3].sum(), rel=1e-12)
    h = len(seqs) // 2
    eng.load_batch(seqs[:h], quals[:h])
    f1, g1, e1, n1 = eng.train_eval(m["x"])
    eng.load_batch(seqs[h:], quals[h:])
    f2, g2, e2, n2 = eng.train_eval(m["x"])
    assert fn == pytest.approx(f1 + f2, rel=1e-11)
    np.testing.assert_allclose(gr, g1 + g2, rtol=1e-9, atol=1e-11)
    assert eff == pytest.approx(e1 + e2, rel=1e-12) and nsk == n1 + n2
    # spot-check against the oracle
    o, x = oracle_for("syn_b.model")
    fo, go, eo, no = o.train_eval(x, seqs, quals, n_threads=8)
    assert fn == pytest.approx(fo, rel=1e-9)
    np.testing.assert_allclose(gr, go, rtol=1e-7, atol=1e-7)


def test_scan_config_b_size_properties_and_spot_checks():
    """BASELINE config B: 1 000 synthetic RNAs of L=150, pattern '((.*.))' -- scan on the batch pipeline.
    Size-independent properties of every record + the oracle on sampled sequences; the fused log-space scan kernel
    (the scan's range fallback; option pipeline = 3 runs the whole scan on it) must give the same records."""
    m = io.read_model(gpath("syn_l1.model"))
    seqs, quals = synth.synth_batch(1000, 150)
    eng = io.engine_from_model(m)
    eng.load_batch(seqs, quals)
    recs, en = eng.scan(m["x"])
    assert eng.last_timing()[2] == 0                       # no sequence needed the log-space fallback
    M = len(eng.describe()["node"])
    for r in recs:
        assert 0.0 <= r["exist_prob"] <= 1.0 + 1e-12
        assert np.logaddexp.reduce(r["start"]) == pytest.approx(np.log(r["exist_prob"]), abs=1e-9)
        assert r["Ys"] == len(r["start"]) - 1 - int(np.argmax(r["start"][::-1]))     # last maximum
        assert r["Ye"] == len(r["end"]) - 1 - int(np.argmax(r["end"][::-1]))
        assert np.all(r["inner"] <= 1e-9) and np.all(r["start"] <= 1e-9)
        assert set(r["rss"]) <= set("OLRHBIM ") and np.all((0 <= r["psihat"]) & (r["psihat"] < M))
        assert r["rss"].count("L") == r["rss"].count("R")
    assert np.all(en >= 0) and np.all(np.isfinite(en))
    o, xo = oracle_for("syn_l1.model")
    for k in (0, 333, 999):
        a = o.scan_seq(seqs[k], quals[k])
        b = recs[k]
        assert (a["Ys"], a["Ye"]) == (b["Ys"], b["Ye"])
        for key in ("start", "end", "inner"):
            assert_log_close(b[key], a[key], rtol=1e-8, atol=1e-6, what=key)
        assert list(a["psihat"]) == list(b["psihat"]) and a["rss"] == b["rss"]
    eng2 = io.engine_from_model(m)
    eng2.set_option("pipeline", 3)
    eng2.load_batch(seqs[:64], quals[:64])
    recs2, en2 = eng2.scan(m["x"])
    for a, b in zip(recs2, recs[:64]):
        assert (a["Ys"], a["Ye"], a["rss"]) == (b["Ys"], b["Ye"], b["rss"]) and list(a["psihat"]) == list(b["psihat"])
        assert_log_close(b["start"], a["start"], rtol=1e-8, atol=1e-6, what="start")


@pytest.mark.parametrize("pattern", ["((.*.))", "(.....)", "((((...))))", ".(.*.).(.*.)."])
def test_staged_cyk_keeps_the_reference_tie_order(pattern):
    """k5_cyk reduces (value, ordinal) pairs across lanes; with uniform theta (x0) ties are everywhere, so the parse must
    be the one of the serial lane-per-target evaluation in the reference's candidate order (option dbg = 8) -- on a
    ragged batch, for automata that take 2, 4 and 8 products per lane."""
    seqs, quals = [], []
    for n, L in ((40, 37), (40, 90), (24, 160)):
        a, b = synth.synth_batch(n, L)
        seqs += a
        quals += b
    eng = api.Engine(pattern)
    eng.load_batch(seqs, quals)
    x = eng.initial_params(1.0)
    recs, en = eng.scan(x)
    eng.set_option("dbg", 8)
    recs2, en2 = eng.scan(x)
    for a, b in zip(recs, recs2):
        assert (a["Ys"], a["Ye"], a["rss"]) == (b["Ys"], b["Ye"], b["rss"])
        assert list(a["psihat"]) == list(b["psihat"])
    np.testing.assert_allclose(en, en2, rtol=1e-12)       # (the sum passes accumulate with atomics)
    o = po.make_oracle(pattern, 50, 30, min_bpp=1e-4, tau=0.1)
    o.set_params(x)
    for k in (0, 41, 103):
        r = o.scan_seq(seqs[k], quals[k])
        assert r["rss"] == recs[k]["rss"] and list(r["psihat"]) == list(recs[k]["psihat"])


def test_pipelines_agree_and_linear_pipeline_is_the_one_measured():
    """The scaled-linear pipeline (default, what bench.py times) against the log-space batch pipeline (its range fallback) on a
    ragged batch: same fn / gr, and no sequence needed the log-space fallback.  The fused kernel of round 1 (pipeline 2) is
    retired for training."""
    m = io.read_model(gpath("syn_b.model"))
    recs = io.read_fastq(gpath("syn_L40_n3.fq")) + io.read_fastq(gpath("syn_L100_n3.fq")) + io.read_fastq(gpath("syn_L150_n8.fq"))
    seqs, quals = [s for _, s, _ in recs], [q for _, _, q in recs]
    res = {}
    for pipe in (4, 3):
        eng = io.engine_from_model(m)
        eng.set_option("pipeline", pipe)
        eng.load_batch(seqs, quals)
        res[pipe] = eng.train_eval(m["x"])
        if pipe == 4:
            assert eng.last_timing()[2] == 0
    for pipe in (3,):
        assert res[pipe][0] == pytest.approx(res[4][0], rel=1e-11)
        np.testing.assert_allclose(res[pipe][1], res[4][1], rtol=1e-9, atol=1e-10)
        assert res[pipe][2:] == res[4][2:]


def test_linear_pipeline_hands_out_of_range_sequences_to_the_log_pipeline():
    """lambda = 40 puts the partition functions (~e^1000 and beyond) outside the double range of the scaled-linear tables:
    those sequences are flagged and re-evaluated in log space, the result still matches the oracle."""
    eng = api.Engine("((.*.))", PAR, 50, 30, 1e-4, 0.1)
    o = po.make_oracle("((.*.))", 50, 30, min_bpp=1e-4, tau=0.1)
    recs = io.read_fastq(gpath("syn_L150_n8.fq"))
    seqs, quals = [s for _, s, _ in recs], [q for _, _, q in recs]
    x = eng.initial_params(40.0)
    o.set_params(x)
    eng.load_batch(seqs, quals)
    fn, gr, eff, nsk = eng.train_eval(x)
    assert eng.last_timing()[2] > 0          # some sequences took the fallback
    fo, go, eo, no = o.train_eval(x, seqs, quals)
    assert nsk == no
    assert fn == pytest.approx(fo, rel=1e-8, abs=1e-9)
    np.testing.assert_allclose(gr, go, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("pattern", ["(((((.*.)))))(((.*.)))", ".", "(.)*(.)"])
def test_other_automaton_sizes(pattern):
    """S = 59 (automaton blob larger than the LDS staging limit: tuple lists stay in global memory, 4 cells per workgroup),
    S = 6 and a pattern with two stems -- fn / gr of a ragged batch against the oracle."""
    eng = api.Engine(pattern, PAR, 50, 30, 1e-4, 0.1)
    o = po.make_oracle(pattern, 50, 30, min_bpp=1e-4, tau=0.1)
    recs = io.read_fastq(gpath("syn_L40_n3.fq")) + io.read_fastq(gpath("syn_L100_n3.fq"))
    seqs, quals = [s for _, s, _ in recs], [q for _, _, q in recs]
    rng = np.random.RandomState(11)
    x = eng.initial_params(0.7)
    x[:-2] += 0.3 * rng.randn(len(x) - 2)
    o.set_params(x)
    eng.load_batch(seqs, quals)
    fn, gr, eff, nsk = eng.train_eval(x)
    fo, go, eo, no = o.train_eval(x, seqs, quals)
    assert nsk == no
    assert fn == pytest.approx(fo, rel=1e-9, abs=1e-10)
    np.testing.assert_allclose(gr, go, rtol=1e-7, atol=1e-7)


def test_long_sequences_take_the_unstaged_exterior_path():
    """L = 2 300 > 2 048: the exterior-chain kernels cannot stage the per-sequence context in LDS (STAGE = false), the band
    kernels run 2 300 cells per diagonal; train evaluation and scan of a ragged batch (2 300, 2 100, 60) against the oracle."""
    pattern = "((.*.))"
    seqs, quals = [], []
    for L, seed in ((2300, 5), (2100, 6), (60, 7)):
        a, b = synth.synth_batch(1, L, seed=seed)
        seqs += a
        quals += b
    quals[1] = quals[1].copy()
    quals[1][-1] = 5                                          # one sequence without motif
    eng = api.Engine(pattern, PAR, 50, 30, 1e-4, 0.1)
    o = po.make_oracle(pattern, 50, 30, min_bpp=1e-4, tau=0.1)
    x = eng.initial_params(0.5)
    x[:-2] += 0.2 * np.random.RandomState(3).randn(len(x) - 2)
    o.set_params(x)
    eng.load_batch(seqs, quals)
    fn, gr, eff, nsk = eng.train_eval(x)
    fo, go, eo, no = o.train_eval(x, seqs, quals, n_threads=3)
    assert nsk == no
    assert fn == pytest.approx(fo, rel=1e-9, abs=1e-10)
    np.testing.assert_allclose(gr, go, rtol=1e-7, atol=1e-7)
    assert eff == pytest.approx(eo, rel=1e-12)
    recs, en = eng.scan(x)
    for k in (0, 2):
        a = o.scan_seq(seqs[k], quals[k])
        assert (a["Ys"], a["Ye"]) == (recs[k]["Ys"], recs[k]["Ye"])
        assert a["rss"] == recs[k]["rss"] and list(a["psihat"]) == list(recs[k]["psihat"])
        assert_log_close(recs[k]["start"], a["start"], rtol=1e-8, atol=1e-6, what="start")
        assert_log_close(recs[k]["end"], a["end"], rtol=1e-8, atol=1e-6, what="end")


def test_sharded_evaluation_with_device_buffers_equals_the_single_engine_one():
    """What bench.py / the command line do on N GPUs, on one: every "rank" keeps its contiguous share resident, writes its
    partial sums into a torch CUDA buffer (elemdp_train_partial with a device pointer), the buffers are summed (the
    all-reduce), and every rank finishes fn / gr from the sum."""
    import torch
    from rnaelem_amd.distributed import assigned_range
    m = io.read_model(gpath("syn_b.model"))
    recs = io.read_fastq(gpath("syn_L40_n3.fq")) + io.read_fastq(gpath("syn_L100_n3.fq")) + io.read_fastq(gpath("syn_L150_n8.fq"))
    seqs, quals = [s for _, s, _ in recs], [q for _, _, q in recs]
    x = m["x"]
    whole = io.engine_from_model(m)
    whole.load_batch(seqs, quals)
    ref = whole.train_eval(x)
    world, total, engines = 3, None, []
    for rank in range(world):
        a, b = assigned_range(len(seqs), world, rank)
        eng = io.engine_from_model(m)
        eng.load_batch(seqs[a:b], quals[a:b])
        buf = torch.zeros(eng.partial_len(), dtype=torch.float64, device="cuda")
        eng.train_partial(x, device_ptr=buf.data_ptr())
        torch.cuda.synchronize()
        host = eng.train_partial(x)                              # the same vector through the host path
        np.testing.assert_allclose(buf.cpu().numpy(), host, rtol=1e-12, atol=1e-14)
        total = buf if total is None else total + buf
        engines.append(eng)
    for eng in engines:
        fn, gr, eff, nsk = eng.train_finish(total.cpu().numpy())
        assert fn == pytest.approx(ref[0], rel=1e-11)
        np.testing.assert_allclose(gr, ref[1], rtol=1e-9, atol=1e-10)
        assert eff == pytest.approx(ref[2], rel=1e-12) and nsk == ref[3]


EVAL_LIK = gload("eval_lik.json")


@pytest.mark.parametrize("case", EVAL_LIK, ids=["%s-%s" % (c["model"], c["fq"]) for c in EVAL_LIK])
def test_lik_ratio_objective_against_reference_golden(case):
    """ELEMDP_LIK_RATIO (--lik-ratio, motif_trainer.hpp:156-202) vs fn / gr of the compiled reference, on batches with
    both labels; every pipeline / schedule gives the same numbers."""
    m = io.read_model(gpath(case["model"]))
    recs = io.read_fastq(gpath(case["fq"]))
    seqs, quals = [s for _, s, _ in recs], [q for _, _, q in recs]
    for pipe, sched in ((4, 1), (4, 0), (3, 1), (3, 0)):
        par = m["ene_param"] if m["ene_param"] in ("~T2004~", "~A2007~") else open(m["ene_param"]).read()
        eng = api.Engine(m["pattern"], par, m["max_span"], m["max_iloop"], m["min_bpp"], m["tau"], m["flags"] | api.LIK_RATIO)
        eng.set_option("pipeline", pipe)
        eng.set_option("schedule", sched)
        eng.load_batch(seqs, quals)
        fn, gr, eff, nsk = eng.train_eval(m["x"])
        assert fn == pytest.approx(case["fn"], rel=1e-9, abs=1e-12), (pipe, sched)
        np.testing.assert_allclose(gr, arr(case["gr"]), rtol=1e-7, atol=1e-9, err_msg=str((pipe, sched)))
        assert nsk == 0
    with pytest.raises(api.ElemdpError):
        eng.set_option("pipeline", 2)            # (retired)


def test_error_behaviour():
    with pytest.raises(api.ElemdpError):
        api.Engine("(.")
    eng = api.Engine("(.)")
    with pytest.raises(api.ElemdpError) as e:
        eng.train_eval(eng.initial_params())
    assert e.value.code == -4      # ELEMDP_ESTATE: no batch loaded
    with pytest.raises(api.ElemdpError):   # quality must have L+1 entries (motif_trainer.hpp:139)
        eng.load_batch([io.encode_seq("ACGU")], [np.zeros(4, dtype=np.uint8)])


def test_baseline_config_full_size():
    """BASELINE configs C/D at full size: 10 000 x L=200 x '((.*.))', x0 with lambda = (1,1).
    Size-independent properties + oracle spot checks on sampled sequences."""
    m = io.read_model(gpath("syn_l1.model"))
    eng = io.engine_from_model(m)
    seqs, quals = synth.synth_batch(10000, 200)
    eng.load_batch(seqs, quals)
    x = m["x"]
    fn, gr, eff, nsk = eng.train_eval(x)
    stats = eng.seq_stats()
    assert nsk == 0 and np.all(np.isfinite(stats[:, :4]))
    assert eng.last_timing()[2] == 0                 # the scaled-linear pipeline handled every sequence
    assert fn == pytest.approx(stats[:, 3].sum(), rel=1e-11)
    assert np.all(stats[:, 3] >= -1e-9)              # f_n = -log P(label | seq) >= 0
    assert np.all(stats[:, 1] <= stats[:, 0] + 1e-9) and np.all(stats[:, 2] <= stats[:, 0] + 1e-9)
    np.testing.assert_allclose(np.logaddexp(stats[:, 1], stats[:, 2]), stats[:, 0], rtol=1e-11)   # Z = Zari (+) Znasi
    assert eff == pytest.approx(eng.bpp_eff().sum(), rel=1e-12)
    fn2, gr2, _, _ = eng.train_eval(x)               # idempotent up to atomic summation order
    assert fn2 == pytest.approx(fn, rel=1e-12)
    np.testing.assert_allclose(gr2, gr, rtol=1e-9, atol=1e-9)
    o, xo = oracle_for("syn_l1.model")
    for k in (0, 1234, 5000, 9999):
        a = o.train_seq(seqs[k], quals[k])
        assert_log_close(stats[k, 0], a["Zo"], rtol=1e-10, what="Zo")
        assert_log_close(stats[k, 1], a["Zari"], rtol=1e-10, what="Zari")
        assert stats[k, 3] == pytest.approx(a["f"], rel=1e-8, abs=1e-10)
        assert eng.bpp_eff()[k] == a["bpp_eff"]

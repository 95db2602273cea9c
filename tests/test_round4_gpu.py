"""Round 4, GPU: band-kernel workgroups that own several blocks of cells (option "nblk"), ranged evaluations, streamed scans."""
import numpy as np
import pytest

from rnaelem_amd import api, synth

pytestmark = pytest.mark.gpu


def ragged(seed, shapes):
    seqs, quals = [], []
    for L, n in shapes:
        s_, q_ = synth.synth_batch(n, L, seed=seed + L)
        seqs += s_
        quals += q_
    return seqs, quals


@pytest.mark.parametrize("pattern", ["((.*.))", "(.....)", "(.(.).)"])
def test_workgroups_of_several_blocks_train(pattern):
    """k4_in / k4_out with nblk blocks of cells per workgroup (context staged once, phases block by block; the heavy sums of
    a block are cleared by the lanes that own them) against one block per workgroup and against the oracle: fn, gr, kept
    fractions on a ragged batch with negatives, both schedules.  Lengths are chosen so that diagonals end in partial blocks and
    in workgroups with fewer blocks than nblk (RNAelemTrainDP::operator(), motif_trainer.hpp:124-272)."""
    from oracle import pyoracle as po
    seqs, quals = ragged(1100, ((40, 5), (200, 7), (97, 6), (131, 6), (13, 3)))
    for k in range(0, len(quals), 2):
        quals[k][-1] = 5
    res = {}
    x = None
    for nblk in (1, 2, 3, 5):
        for sched in (0, 1):
            eng = api.Engine(pattern, "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
            eng.set_option("nblk", nblk)
            eng.set_option("schedule", sched)
            eng.load_batch(seqs, quals)
            if x is None:
                x = eng.initial_params(0.7)
                x[:-2] += np.linspace(-0.3, 0.3, len(x) - 2)
            res[(nblk, sched)] = eng.train_eval(x)
            again = eng.train_eval(x)      # (the same slots once more: nothing left behind in the tables)
            assert again[0] == pytest.approx(res[(nblk, sched)][0], rel=1e-12)
    ref = res[(1, 0)]
    for key, r in res.items():
        assert r[0] == pytest.approx(ref[0], rel=1e-10), key
        np.testing.assert_allclose(r[1], ref[1], rtol=1e-8, atol=1e-9, err_msg=str(key))
        assert r[2] == ref[2] and r[3] == ref[3]
    o = po.make_oracle(pattern, 50, 30, min_bpp=1e-4, tau=0.1)
    fo, go, eo, no = o.train_eval(x, seqs[:8], quals[:8])
    eng = api.Engine(pattern, "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
    eng.set_option("nblk", 3)
    eng.load_batch(seqs[:8], quals[:8])
    fn, gr, eff, nsk = eng.train_eval(x)
    assert fn == pytest.approx(fo, rel=1e-9)
    np.testing.assert_allclose(gr, go, rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize("pattern", ["((.*.))", "(.....)"])
def test_workgroups_of_several_blocks_scan(pattern):
    """The scan's sum passes (start / inner posteriors, the constrained passes with their skipped blocks, end posteriors) with
    several blocks per workgroup against one (motif_scanner.hpp:186-252)."""
    seqs, quals = ragged(1300, ((35, 4), (180, 6), (97, 5), (300, 3), (64, 5)))
    for k in range(0, len(quals), 3):
        quals[k][len(quals[k]) // 2] = 5
    out = {}
    x = None
    for nblk in (1, 2, 4):
        eng = api.Engine(pattern, "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
        eng.set_option("nblk", nblk)
        eng.load_batch(seqs, quals)
        if x is None:
            x = eng.initial_params(0.7)
            x[:-2] += np.linspace(-0.4, 0.4, len(x) - 2)
        out[nblk] = eng.scan(x)
    r0, en0 = out[1]
    for nblk in (2, 4):
        r1, en1 = out[nblk]
        np.testing.assert_allclose(en1, en0, rtol=1e-9, atol=1e-12)
        for a_, b_ in zip(r0, r1):
            assert (a_["Ys"], a_["Ye"]) == (b_["Ys"], b_["Ye"])
            assert a_["rss"] == b_["rss"] and np.array_equal(a_["psihat"], b_["psihat"])
            assert b_["exist_prob"] == pytest.approx(a_["exist_prob"], rel=1e-9)
            for key in ("start", "inner", "end"):
                fa, fb = np.isfinite(a_[key]), np.isfinite(b_[key])
                assert np.array_equal(fa, fb), key
                np.testing.assert_allclose(b_[key][fb], a_[key][fa], rtol=1e-8, atol=1e-9, err_msg=key)


def test_three_blocks_per_workgroup_at_size():
    """At a size where a launch has many more workgroups than the chip keeps resident: three blocks per workgroup against the
    default (one), results and rate of both (the rates are equal within the noise: DESIGN.md 4.2d)."""
    seqs, quals = synth.synth_batch(1024, 200, seed=77)
    eng = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
    eng.load_batch(seqs, quals)
    x = eng.initial_params(1.0)
    a = eng.train_eval(x)
    a = eng.train_eval(x)
    ms_one = eng.last_timing()[1]
    eng.set_option("nblk", 3)
    b = eng.train_eval(x)
    b = eng.train_eval(x)
    ms_three = eng.last_timing()[1]
    print("1024 x L=200: %.1f ms with one block per workgroup, %.1f ms with three" % (ms_one, ms_three))
    assert a[0] == pytest.approx(b[0], rel=1e-10)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-8, atol=1e-9)


def test_ranged_evaluation_equals_loading_the_range_alone():
    """Options eval_first / eval_count (the mini-batch trainer's look-ahead: eight coming batches resident as one, evaluated
    range by range) against loading that range as a batch of its own: fn, gr, per-sequence rows, skipped count -- default and
    deterministic mode; ranges that the engine cannot honour are refused instead of silently evaluating everything."""
    seqs, quals = ragged(1500, ((60, 10), (120, 12), (200, 10)))
    for k in range(1, len(quals), 2):
        quals[k][-1] = 5
    eng = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
    eng.load_batch(seqs, quals)
    x = eng.initial_params(0.5)
    x[:-2] += np.linspace(-0.2, 0.2, len(x) - 2)
    for det in (0, 1):
        for first, count in ((0, 7), (7, 12), (19, 13), (0, 32)):
            eng.set_option("deterministic", det)
            eng.set_option("eval_first", first)
            eng.set_option("eval_count", count)
            fn, gr, eff, nsk = eng.train_eval(x)
            rows = eng.seq_stats()[first:first + count].copy()
            one = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
            one.set_option("deterministic", det)
            one.load_batch(seqs[first:first + count], quals[first:first + count])
            f1, g1, e1, n1 = one.train_eval(x)
            assert fn == pytest.approx(f1, rel=1e-11), (det, first, count)
            np.testing.assert_allclose(gr, g1, rtol=1e-9, atol=1e-10)
            assert nsk == n1
            np.testing.assert_allclose(rows, one.seq_stats(), rtol=1e-10, atol=1e-12)
    eng.set_option("deterministic", 0)
    eng.set_option("eval_first", 25)
    eng.set_option("eval_count", 12)          # past the end of the batch
    with pytest.raises(Exception):
        eng.train_eval(x)
    # the log-space pipeline and a streamed batch evaluate whole batches only
    e3 = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
    e3.set_option("pipeline", 3)
    e3.load_batch(seqs, quals)
    e3.set_option("eval_first", 4)
    e3.set_option("eval_count", 8)
    with pytest.raises(Exception):
        e3.train_eval(x)
    e3.set_option("eval_first", 0)
    e3.set_option("eval_count", len(seqs))    # (the whole batch is no range)
    full = e3.train_eval(x)
    es = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
    es.set_option("max_resident", 10)
    es.load_batch(seqs, quals)
    es.set_option("eval_first", 4)
    es.set_option("eval_count", 8)
    with pytest.raises(Exception):
        es.train_eval(x)
    es.set_option("eval_count", 0)
    whole = es.train_eval(x)
    assert whole[0] == pytest.approx(full[0], rel=1e-9)


from rnaelem_amd import io, train    # noqa: E402
from tests.util import assert_log_close, gload, gpath  # noqa: E402

LIK_SHUFFLE = gload("train_trace_lik_shuffle.json")


@pytest.mark.parametrize("t", LIK_SHUFFLE, ids=["%s-%s-batch%d" % (c["fq"], c["pattern"], c["batch_size"]) for c in LIK_SHUFFLE])
def test_lik_ratio_with_shuffled_negatives_on_the_gpu_reproduces_the_reference_trace(t):
    """`elem train --lik-ratio` in the default mode (Adam, a shuffled negative per record and iteration: motif_trainer.hpp:156-202
    with :145-152) the way the command line runs it -- two engines, records and negatives loaded as one batch, look-ahead of
    several evaluations, ranged evaluations -- against the data term and |gr|^2 `RNAelem train --lik-ratio` prints."""
    recs = io.read_fastq(gpath(t["fq"]))
    seqs, quals = [s for _, s, _ in recs], [q for _, _, q in recs]
    engs = [api.Engine(t["pattern"], "~T2004~", 50, 30, 1e-4, t["tau"], api.LIK_RATIO) for _ in range(2)]
    ev = train.MiniBatches(seqs, quals, t["batch_size"], None, kmer_shuf=t["kmer_shuf"], engines=engs)
    x0 = engs[0].initial_params(t["lambda_init"])
    r = train.minimize_adam(ev, x0, train.regularisation(len(x0), t["rho_theta"], t["rho_lambda"]), max_iter=t["max_iter"])
    ev.finish()
    assert len(r["trace"]) == len(t["iter_fn"])
    for row, y, g2 in zip(r["trace"], t["iter_fn"], t["iter_gnorm"]):
        assert row[4] == pytest.approx(y, rel=2e-5, abs=1e-8)
        assert row[2] == pytest.approx(g2, rel=2e-5)


def test_streamed_scan_of_several_large_chunks_equals_the_resident_scan():
    """Config E's path at a size where chunks are real batches: a scan of 3 200 sequences streamed in chunks of 1 000 (three
    full chunks and a rest; filter + plan of the next chunk built on the second inner engine while the current one is scanned)
    against the resident scan of the same batch -- Ys / Ye / parse exact, posteriors and E[N] to 1e-10, records in input
    order -- and against the oracle on a few of its records (RNAelemScanner::scan, motif_scanner.hpp:938-949)."""
    from oracle import pyoracle as po
    seqs, quals = ragged(1700, ((120, 1500), (90, 1000), (150, 700)))
    order = np.random.RandomState(3).permutation(len(seqs))
    seqs, quals = [seqs[k] for k in order], [quals[k] for k in order]      # (mixed lengths inside every chunk)
    res = api.Engine("(.....)", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
    res.load_batch(seqs, quals)
    x = res.initial_params(1.0)
    x[:-2] += np.linspace(-0.3, 0.3, len(x) - 2)
    r0, en0 = res.scan(x)
    eng = api.Engine("(.....)", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
    eng.set_option("max_resident", 1000)
    eng.load_batch(seqs, quals)
    r1, en1 = eng.scan(x)
    assert len(r1) == len(seqs)
    np.testing.assert_allclose(en1, en0, rtol=1e-10, atol=1e-12)
    for a_, b_ in zip(r0, r1):
        assert (a_["Ys"], a_["Ye"]) == (b_["Ys"], b_["Ye"])
        assert a_["rss"] == b_["rss"] and np.array_equal(a_["psihat"], b_["psihat"])
        assert b_["exist_prob"] == pytest.approx(a_["exist_prob"], rel=1e-10)
        for key in ("start", "inner", "end"):
            fa, fb = np.isfinite(a_[key]), np.isfinite(b_[key])
            assert np.array_equal(fa, fb), key
            np.testing.assert_allclose(b_[key][fb], a_[key][fa], rtol=1e-10, atol=1e-10, err_msg=key)
    o = po.make_oracle("(.....)", 50, 30, min_bpp=1e-4, tau=0.1)
    o.set_params(x)
    for k in (0, 999, 1000, 2500, 3199):
        a = o.scan_seq(seqs[k], quals[k])
        assert (r1[k]["Ys"], r1[k]["Ye"]) == (a["Ys"], a["Ye"]) and r1[k]["rss"] == a["rss"]
        assert r1[k]["exist_prob"] == pytest.approx(a["exist_prob"], rel=1e-8)


@pytest.mark.parametrize("pattern,prune", [("((.*.))", 1), ("(.....)", 1), ("(.*.)(.*.)(.*.)", 1), ("(.*.)(.*.)(.*.)", 0), ("((.*.))", 0)])
def test_deterministic_mode_on_other_automata(pattern, prune):
    """Option "deterministic" (one copy of the heavy sums, every sum single-writer: pairs of a cell in one wave, tuples dealt to the
    waves by target, stem cells by wave) on automata of other shapes: the complete lists of `(.*.)(.*.)(.*.)` have 124 pairs -- more
    than a wave, so one wave alone walks the pair phases -- and run the generic kernels.  Repeats are bit-identical (also after an
    evaluation at another point through the same slots), and equal the default mode to its own reproducibility."""
    seqs, quals = ragged(1900, ((30, 4), (120, 6), (61, 5), (200, 4)))
    for k in range(0, len(quals), 3):
        quals[k][-1] = 5
    eng = api.Engine(pattern, "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
    eng.set_option("prune", prune)
    eng.load_batch(seqs, quals)
    x = eng.initial_params(0.6)
    x[:-2] += np.linspace(-0.25, 0.25, len(x) - 2)
    ref = eng.train_eval(x)
    eng.set_option("deterministic", 1)
    a = eng.train_eval(x)
    eng.train_eval(x + 0.01)
    b = eng.train_eval(x)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3]
    assert a[0] == pytest.approx(ref[0], rel=1e-11)
    np.testing.assert_allclose(a[1], ref[1], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("pattern", ["((.*.))", "(.....)", ".(.).", "(.(.).)"])
def test_viterbi_pass_on_the_compact_table_equals_the_dense_one(pattern, monkeypatch):
    """k5_cyk / k5_cyk_ext / the traceback on the compact table layout (TableView::ldm / stm: every cell stored, a missing column
    read as log 0) against the dense [e][d][i][s] table of round 3 (ELEMDP_CYK_DENSE, read at every launch): the same parse for
    every sequence of a ragged batch with masked positions, poly-A stretches (ties) included -- CYKFun / trace_back,
    motif_scanner.hpp:802-913, :262-362."""
    seqs, quals = ragged(2100, ((33, 5), (150, 6), (300, 3), (71, 6)))
    seqs.append(np.full(90, 1, dtype=np.uint8))                      # AAAA...: every parse ties with many others
    quals.append(np.r_[np.full(90, 10, dtype=np.uint8), np.uint8(0)])
    for k in range(0, len(quals), 4):
        quals[k][len(quals[k]) // 3] = 5
    eng = api.Engine(pattern, "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
    eng.load_batch(seqs, quals)
    x = eng.initial_params(0.8)
    x[:-2] += np.linspace(-0.35, 0.35, len(x) - 2)
    monkeypatch.delenv("ELEMDP_CYK_DENSE", raising=False)
    compact, _ = eng.scan(x)
    monkeypatch.setenv("ELEMDP_CYK_DENSE", "1")
    dense, _ = eng.scan(x)
    for a_, b_ in zip(compact, dense):
        assert (a_["Ys"], a_["Ye"]) == (b_["Ys"], b_["Ye"])
        assert a_["rss"] == b_["rss"] and np.array_equal(a_["psihat"], b_["psihat"])


def test_filter_from_the_candidate_table_equals_the_mask_walk(monkeypatch):
    """Rule 6c of the BPP filter from the candidate table (one load and one fma per candidate, the factor of the other pair
    folded into extra planes) -- as one workgroup per sequence sweeping all diagonals (k6_in_seq / k6_out_seq, the default) and as
    one launch per diagonal (k6_in_tab / k6_out_tab, ELEMDP_BPP_DIAG) -- against round 2's walk over the pair mask
    (ELEMDP_BPP_WALK): identical kept sets and kept fractions, ln BPP to 1e-10 -- ragged lengths (C = W - 7 < 30 for the short
    ones), N bases, max_iloop 0 .. 30, a band of 120."""
    rng = np.random.default_rng(4)
    for W, C in ((50, 30), (50, 7), (24, 30), (120, 30), (50, 0), (50, 2), (50, 5)):
        seqs, quals = [], []
        for L in (1, 5, 9, 20, 31, 33, 47, 64, 100, 150, 301, 600):
            s_, q_ = synth.synth_batch(3, L, seed=int(rng.integers(1 << 30)))
            if L > 30:
                s_[1] = s_[1].copy()
                s_[1][rng.integers(0, L, size=L // 12)] = 0
            seqs += s_
            quals += q_
        res = {}
        for mode in ("", "ELEMDP_BPP_DIAG", "ELEMDP_BPP_WALK"):
            for m in ("ELEMDP_BPP_DIAG", "ELEMDP_BPP_WALK"):
                monkeypatch.delenv(m, raising=False)
            if mode:
                monkeypatch.setenv(mode, "1")
            eng = api.Engine("(.)", "~T2004~", W, C, 1e-4)
            eng.set_option("keep_lnbpp", 1)
            eng.load_batch(seqs, quals)
            res[mode] = ([eng.pairs(k, with_lnbpp=True) for k in range(len(seqs))], eng.bpp_eff())
        for m in ("ELEMDP_BPP_DIAG", "ELEMDP_BPP_WALK"):
            monkeypatch.delenv(m, raising=False)
        for mode in ("", "ELEMDP_BPP_DIAG"):
            for k, ((ka, la), (kb, lb)) in enumerate(zip(res[mode][0], res["ELEMDP_BPP_WALK"][0])):
                assert np.array_equal(ka, kb), (mode, W, C, k)
                assert_log_close(la, lb, rtol=1e-10, atol=1e-10, what="lnbpp %s W=%d C=%d k=%d" % (mode, W, C, k))
            np.testing.assert_array_equal(res[mode][1], res["ELEMDP_BPP_WALK"][1])


def test_plan_builder_forms_agree(monkeypatch):
    """The plan of a batch built three ways gives the same evaluation: role lists by one workgroup per (sequence, role) with LDS
    counters + popcount count pass (default), role lists by global atomics in separate passes (ELEMDP_ROLE_GLOBAL; also what a
    sequence too long for the LDS counters forces: L = 900), count pass by enumeration (ELEMDP_PLAN_ENUM_COUNT; also what a
    sequence with N bases takes).  Default mode (atomic sums: 1e-11) and deterministic mode (sorted segments, item copies by
    k_permute_items: bit-identical)."""
    rng = np.random.default_rng(12)
    seqs, quals = ragged(300, ((60, 5), (200, 6), (131, 4), (33, 3)))
    seqs[2] = seqs[2].copy()
    seqs[2][rng.integers(0, len(seqs[2]), size=6)] = 0          # N bases: the enumerating count pass for this sequence
    long_s, long_q = synth.synth_batch(1, 900, seed=77)
    res = {}
    for det in (0, 1):
        for mode in ("", "ELEMDP_ROLE_GLOBAL", "ELEMDP_PLAN_ENUM_COUNT", "long"):
            for m in ("ELEMDP_ROLE_GLOBAL", "ELEMDP_PLAN_ENUM_COUNT"):
                monkeypatch.delenv(m, raising=False)
            if mode.startswith("ELEMDP"):
                monkeypatch.setenv(mode, "1")
            eng = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
            if det:
                eng.set_option("deterministic", 1)
            if mode == "long":     # the long sequence takes the whole plan set to the global passes; its own terms are subtracted below
                eng.load_batch(seqs + long_s, quals + long_q)
            else:
                eng.load_batch(seqs, quals)
            x = eng.initial_params(1.0)
            fn, gr, eff, nsk = eng.train_eval(x)
            res[(det, mode)] = (fn, gr, eng.seq_stats()[:len(seqs)].copy(), eng.bpp_eff()[:len(seqs)].copy())
        for m in ("ELEMDP_ROLE_GLOBAL", "ELEMDP_PLAN_ENUM_COUNT"):
            monkeypatch.delenv(m, raising=False)
        ref = res[(det, "")]
        for mode in ("ELEMDP_ROLE_GLOBAL", "ELEMDP_PLAN_ENUM_COUNT"):
            got = res[(det, mode)]
            if det:
                assert got[0] == ref[0] and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
            else:
                assert got[0] == pytest.approx(ref[0], rel=1e-11)
                np.testing.assert_allclose(got[1], ref[1], rtol=1e-10, atol=1e-11)
            np.testing.assert_array_equal(got[3], ref[3])
        got = res[(det, "long")]           # per-sequence rows of the shared sequences are the same whatever else is in the batch
        if det:
            assert np.array_equal(got[2], ref[2])
        else:
            np.testing.assert_allclose(got[2], ref[2], rtol=1e-10, atol=1e-11)
        np.testing.assert_array_equal(got[3], ref[3])


@pytest.mark.parametrize("pattern", ["((.*.))", "(.....)"])
def test_table_layouts_agree(pattern):
    """The compact tables without row padding (the default since round 4), with rows padded to 64-byte lines (row_pad = 8, the
    layout of round 3) and cell by cell (cell_major: the seven rows of a cell side by side) hold the same values: a deterministic
    train evaluation is bit-identical, a default one agrees to 1e-11, and the scan's records are the same."""
    seqs, quals = ragged(500, ((70, 4), (200, 5), (33, 3), (121, 4)))
    res = {}
    for name, opts in (("plain", {}), ("pad8", {"row_pad": 8}), ("cell", {"cell_major": 1}), ("cell8", {"cell_major": 1, "row_pad": 8})):
        for det in (0, 1):
            eng = api.Engine(pattern, "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
            for k, v in opts.items():
                eng.set_option(k, v)
            if det:
                eng.set_option("deterministic", 1)
            eng.load_batch(seqs, quals)
            x = eng.initial_params(1.0)
            res[(name, det)] = eng.train_eval(x)[:2] + (eng.seq_stats().copy(),)
            if det:
                res[(name, "scan")] = eng.scan(x)
    for name in ("pad8", "cell", "cell8"):
        a, b = res[(name, 1)], res[("plain", 1)]
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), name
        a, b = res[(name, 0)], res[("plain", 0)]
        assert a[0] == pytest.approx(b[0], rel=1e-11)
        np.testing.assert_allclose(a[1], b[1], rtol=1e-10, atol=1e-11)
        (ra, ea), (rb, eb) = res[(name, "scan")], res[("plain", "scan")]
        for p, q in zip(ra, rb):
            assert (p["Ys"], p["Ye"], p["rss"]) == (q["Ys"], q["Ye"], q["rss"]) and list(p["psihat"]) == list(q["psihat"])
            np.testing.assert_allclose(p["start"], q["start"], rtol=1e-10, atol=1e-300)
            np.testing.assert_allclose(p["end"], q["end"], rtol=1e-10, atol=1e-300)
            assert p["exist_prob"] == pytest.approx(q["exist_prob"], rel=1e-10)
        np.testing.assert_allclose(ea, eb, rtol=1e-10, atol=1e-12)


def test_filter_forms_on_random_lengths(monkeypatch):
    """The per-sequence filter kernels (default) against the mask walk on 150 sequences of random lengths 1 .. 600 in one batch
    (the LDS image of the per-sequence kernels is sized by the longest: 62 KB, three workgroups per CU -- and 88 KB with the 900 of
    the third round, one per CU; short sequences take C = W - 7 < 30; every tenth has N
    bases; a band of 50 and one of 33): identical kept sets and kept fractions, ln BPP to 1e-10."""
    rng = np.random.default_rng(77)
    for W, lmax in ((50, 600), (33, 600), (50, 900)):
        seqs, quals = [], []
        for k in range(150):
            L = int(rng.integers(1, lmax + 1)) if k % 3 else int(rng.integers(1, 70))
            (s_,), (q_,) = synth.synth_batch(1, L, seed=int(rng.integers(1 << 30)))
            if k % 10 == 0 and L > 10:
                s_ = s_.copy()
                s_[rng.integers(0, L, size=max(1, L // 20))] = 0
            seqs.append(s_)
            quals.append(q_)
        res = {}
        for mode in ("", "ELEMDP_BPP_WALK"):
            monkeypatch.delenv("ELEMDP_BPP_WALK", raising=False)
            if mode:
                monkeypatch.setenv(mode, "1")
            eng = api.Engine("(.)", "~T2004~", W, 30, 1e-4)
            eng.set_option("keep_lnbpp", 1)
            eng.load_batch(seqs, quals)
            res[mode] = ([eng.pairs(k, with_lnbpp=True) for k in range(len(seqs))], eng.bpp_eff())
        monkeypatch.delenv("ELEMDP_BPP_WALK", raising=False)
        for k, ((ka, la), (kb, lb)) in enumerate(zip(res[""][0], res["ELEMDP_BPP_WALK"][0])):
            assert np.array_equal(ka, kb), (W, k, len(seqs[k]))
            assert_log_close(la, lb, rtol=1e-10, atol=1e-10, what="lnbpp W=%d k=%d L=%d" % (W, k, len(seqs[k])))
        np.testing.assert_array_equal(res[""][1], res["ELEMDP_BPP_WALK"][1])

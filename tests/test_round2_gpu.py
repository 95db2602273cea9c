"""Round-2 GPU parity (through the C ABI): BASELINE config E's workload, scan in --no-rss mode, model files from either
writer, the `eval` sub-command, pruned vs complete transition lists, concurrent vs serial streams, full-size gradient
against the oracle, error paths of load_batch, device binding."""
import os
import threading

import numpy as np
import pytest

from oracle import pyoracle as po
from rnaelem_amd import api, cli, io, synth
from tests.util import arr, assert_log_close, gload, gpath

pytestmark = pytest.mark.gpu


def check_scan_case(case, eng, m):
    recs = io.read_fastq(gpath(case["fq"]))
    eng.load_batch([s for _, s, _ in recs], [q for _, _, q in recs])
    got, en = eng.scan(m["x"])
    byid = {r["id"]: r for r in case["records"]}
    for (rid, seq, qual), g in zip(recs, got):
        r = byid[rid]
        assert (g["Ys"], g["Ye"]) == (r["Ys"], r["Ye"])
        assert g["rss"] == r["rss"] and list(g["psihat"]) == r["psihat"]
        for k in ("start", "end", "inner"):       # `RNAelem scan` prints 6 significant digits
            ref = arr(r[k])
            assert np.array_equal(np.isneginf(ref), np.isneginf(g[k])), k
            mk = ~np.isneginf(ref)
            np.testing.assert_allclose(g[k][mk], ref[mk], rtol=2e-5, atol=1e-9)
        assert g["exist_prob"] == pytest.approx(r["exist_prob"], rel=2e-5)
    return recs, got


def test_config_e_records_against_the_reference_binary_and_the_oracle():
    """BASELINE config E's workload: L = 300 scanned with a '(.....)' model (S = 29).  Ys / Ye / psihat / rss exact against
    `RNAelem scan`, log posteriors against the oracle to 1e-6 with the same -inf pattern."""
    case = gload("scan_e.json")[0]
    m = io.read_model(gpath(case["model"]))
    eng = io.engine_from_model(m)
    assert eng.n_state == 29
    recs, got = check_scan_case(case, eng, m)
    assert eng.last_timing()[2] == 0
    o, x = po.oracle_from_model(gpath(case["model"]))
    for (rid, seq, qual), g in zip(recs, got):
        a = o.scan_seq(seq, qual)
        for k in ("start", "end", "inner"):
            assert_log_close(g[k], a[k], rtol=1e-8, atol=1e-6, what=k)
        assert g["exist_prob"] == pytest.approx(a["exist_prob"], rel=1e-9)


def test_config_e_shape_at_scale_properties_and_spot_checks():
    """2 048 x L=300 x '(.....)': size-independent properties of every record + the oracle on sampled sequences."""
    m = io.read_model(gpath("trna_a.model"))
    eng = io.engine_from_model(m)
    seqs, quals = synth.synth_batch(2048, 300)
    eng.load_batch(seqs, quals)
    recs, en = eng.scan(m["x"])
    assert eng.last_timing()[2] == 0
    M = len(eng.describe()["node"])
    for r in recs:
        assert 0.0 <= r["exist_prob"] <= 1.0 + 1e-12
        assert np.logaddexp.reduce(r["start"]) == pytest.approx(np.log(r["exist_prob"]), abs=1e-9)
        assert r["Ys"] == len(r["start"]) - 1 - int(np.argmax(r["start"][::-1]))     # last maximum (util.hpp:232-241)
        assert r["Ye"] == len(r["end"]) - 1 - int(np.argmax(r["end"][::-1]))
        assert np.all(r["inner"] <= 1e-9) and np.all(r["start"] <= 1e-9) and np.all(r["end"] <= 1e-9)
        assert set(r["rss"]) <= set("OLRHBIM ") and np.all((0 <= r["psihat"]) & (r["psihat"] < M))
        assert r["rss"].count("L") == r["rss"].count("R")
        motif = [h for h in r["psihat"] if 0 < h < M - 1]
        assert motif == sorted(motif)                      # nodes of the motif appear in pattern order
    assert np.all(en >= 0) and np.all(np.isfinite(en))
    o, xo = po.oracle_from_model(gpath("trna_a.model"))
    for k in (0, 777, 2047):
        a, b = o.scan_seq(seqs[k], quals[k]), recs[k]
        assert (a["Ys"], a["Ye"]) == (b["Ys"], b["Ye"])
        for key in ("start", "end", "inner"):
            assert_log_close(b[key], a[key], rtol=1e-8, atol=1e-6, what=key)
        assert list(a["psihat"]) == list(b["psihat"]) and a["rss"] == b["rss"]
    # a second scan of the resident batch gives the same records (tables and trace slots are reused)
    recs2, en2 = eng.scan(m["x"])
    for a, b in zip(recs, recs2):
        assert (a["Ys"], a["Ye"], a["rss"]) == (b["Ys"], b["Ye"], b["rss"]) and list(a["psihat"]) == list(b["psihat"])


def test_scan_in_no_rss_mode():
    """elemdp_scan under ELEMDP_NO_RSS (motif_model.hpp:171-206 with the scanner functors): the reference's own 2.model"""
    case = gload("scan_norss.json")[0]
    m = io.read_model(gpath(case["model"]))
    assert m["no_rss"]
    check_scan_case(case, io.engine_from_model(m), m)


def test_scan_with_a_model_written_by_our_writer_and_read_by_the_reference():
    case = gload("scan_written_model.json")[0]
    m = io.read_model(gpath(case["model"]))
    check_scan_case(case, io.engine_from_model(m), m)


@pytest.mark.parametrize("model,fq", [("syn_b.model", "syn_L150_n8.fq"), ("trna_a.model", "positive_head6.fq"), ("1.model", "0.fq")])
def test_pruned_and_complete_transition_lists_give_the_same_numbers(model, fq):
    """option "prune" (default 1): the lists of Automaton::flatten without the transitions that cannot occur in a complete
    parse -- same fn / gr, same scan records, log posteriors to rounding."""
    m = io.read_model(gpath(model))
    recs = io.read_fastq(gpath(fq))
    seqs, quals = [s for _, s, _ in recs], [q for _, _, q in recs]
    res = {}
    for prune in (1, 0):
        eng = io.engine_from_model(m)
        eng.set_option("prune", prune)
        eng.load_batch(seqs, quals)
        res[prune] = (eng.train_eval(m["x"]), eng.scan(m["x"]))
    (a, sa), (b, sb) = res[1], res[0]
    assert a[0] == pytest.approx(b[0], rel=1e-12) and a[2:] == b[2:]
    np.testing.assert_allclose(a[1], b[1], rtol=1e-10, atol=1e-12)
    for x, y in zip(sa[0], sb[0]):
        assert (x["Ys"], x["Ye"], x["rss"]) == (y["Ys"], y["Ye"], y["rss"]) and list(x["psihat"]) == list(y["psihat"])
        for k in ("start", "end", "inner"):
            assert_log_close(x[k], y[k], rtol=1e-10, atol=1e-10, what=k)


def test_eval_subcommand_prints_the_reference_numbers(tmp_path):
    """`cli eval`: fn / gr of tests/golden/eval.json (compiled reference, 17 digits) in the layout of `RNAelem eval`"""
    c = gload("eval.json")[0]
    out1, out2 = str(tmp_path / "fn.txt"), str(tmp_path / "gr.txt")
    cli.main(["eval", "--fastq", gpath(c["fq"]), "--motif-model", gpath(c["model"]), "--out1", out1, "--out2", out2])
    l1, l2 = open(out1).read(), open(out2).read()
    assert l1.startswith("fn: ") and l2.startswith("gr: [") and l2.endswith("]\n")
    assert float(l1[4:]) == pytest.approx(c["fn"], rel=1e-9)
    np.testing.assert_allclose([float(v) for v in l2[5:-2].split(",")], arr(c["gr"]), rtol=1e-7, atol=1e-7)
    # array-eval: the parts add up to the whole (motif_array_trainer.hpp:20-58 sums the same three keys)
    fn, gr, eff = 0.0, 0.0, 0.0
    for tid in (1, 2):
        cli.main(["array-eval", "--fastq", gpath(c["fq"]), "--motif-model", gpath(c["model"]), "--array", "2", "--task-id", str(tid),
                  "--out4", str(tmp_path / "part")])
        d = dict(line.split(": ", 1) for line in open(str(tmp_path / "part") + "-%d" % tid).read().strip().split("\n"))
        fn += float(d["fn"])
        gr = gr + np.array([float(v) for v in d["gr"].strip("[]").split(",")])
        eff += float(d["sum eff"])
    assert fn == pytest.approx(c["fn"], rel=1e-9) and eff == pytest.approx(c["sum_eff"], rel=1e-12)
    np.testing.assert_allclose(gr, arr(c["gr"]), rtol=1e-7, atol=1e-7)


def test_rejected_batch_leaves_the_handle_without_a_batch():
    """load_batch validates before it commits: after a rejected batch the handle is in the 'no batch' state (ELEMDP_ESTATE)
    instead of the new sizes over the old device buffers."""
    eng = api.Engine("((.*.))")
    seqs, quals = synth.synth_batch(4, 40)
    eng.load_batch(seqs, quals)
    x = eng.initial_params(1.0)
    ref = eng.train_eval(x)
    big_s, big_q = synth.synth_batch(64, 90)
    big_q[40] = big_q[40][:-1]                       # one bad record in a LARGER batch
    with pytest.raises(api.ElemdpError):
        eng.load_batch(big_s, big_q)
    with pytest.raises(api.ElemdpError) as e:
        eng.train_eval(x)
    assert e.value.code == -4
    with pytest.raises(api.ElemdpError) as e:
        eng.scan(x)
    assert e.value.code == -4
    eng.load_batch(seqs, quals)                      # and the handle is still usable
    again = eng.train_eval(x)
    assert again[0] == pytest.approx(ref[0], rel=1e-12)


def test_engine_binds_its_device_in_every_entry_point():
    """A handle used from another host thread (MiniBatches' prefetch thread) must run on ITS device: every entry point sets
    it.  With one GPU the thread case is exercised on device 0; with two, an engine on device 1 is driven from a thread whose
    current device is 0."""
    import torch
    n_dev = torch.cuda.device_count()
    dev = 1 if n_dev >= 2 else 0
    eng = api.Engine("((.*.))", device=dev)
    seqs, quals = synth.synth_batch(6, 60)
    x = eng.initial_params(1.0)
    box = {}

    def work():
        try:
            eng.load_batch(seqs, quals)
            box["res"] = eng.train_eval(x)
        except Exception as ex:      # noqa: BLE001
            box["err"] = ex

    th = threading.Thread(target=work)
    th.start()
    th.join()
    assert "err" not in box, box.get("err")
    o = po.make_oracle("((.*.))", 50, 30, min_bpp=1e-4, tau=0.1)
    fo, go, eo, no = o.train_eval(x, seqs, quals)
    assert box["res"][0] == pytest.approx(fo, rel=1e-9)
    np.testing.assert_allclose(box["res"][1], go, rtol=1e-7, atol=1e-7)
    fn2 = eng.train_eval(x)[0]                       # and from the creating thread again
    assert fn2 == pytest.approx(fo, rel=1e-9)


def test_gradient_of_a_medium_batch_against_the_oracle_on_the_concurrent_stream_path():
    """192 x L=200: large enough for the concurrent groups / second outside stream (n >= 64), small enough for the oracle.
    gr against the oracle (1e-7), and the serial-stream evaluation (group_streams = 1, two_streams = 0) to 1e-10."""
    m = io.read_model(gpath("syn_b.model"))
    eng = io.engine_from_model(m)
    seqs, quals = synth.synth_batch(192, 200, seed=4242)
    for k in range(0, 192, 5):
        quals[k][-1] = 5                              # a fifth of the records without motif
    eng.load_batch(seqs, quals)
    x = m["x"]
    fn, gr, eff, nsk = eng.train_eval(x)
    o, xo = po.oracle_from_model(gpath("syn_b.model"))
    fo, go, eo, no = o.train_eval(xo, seqs, quals, n_threads=min(16, len(os.sched_getaffinity(0))))
    assert nsk == no and fn == pytest.approx(fo, rel=1e-9) and eff == pytest.approx(eo, rel=1e-12)
    np.testing.assert_allclose(gr, go, rtol=1e-7, atol=1e-7)
    eng.set_option("group_streams", 1)
    eng.set_option("two_streams", 0)
    fn1, gr1, _, _ = eng.train_eval(x)
    assert fn1 == pytest.approx(fn, rel=1e-12)
    np.testing.assert_allclose(gr1, gr, rtol=1e-10, atol=1e-10)


def test_full_size_gradient_subsample_against_the_oracle():
    """BASELINE config C at full size: the gradient of the 10 000-sequence batch minus the gradient of the batch without a
    64-sequence subsample equals the subsample's own gradient, which equals the oracle's (1e-7): the full-size evaluation is
    checked against the CPU restatement through linearity, not only against itself."""
    m = io.read_model(gpath("syn_l1.model"))
    x = m["x"]
    seqs, quals = synth.synth_batch(10000, 200)
    pick = list(range(17, 10000, 157))[:64]
    rest = sorted(set(range(10000)) - set(pick))
    eng = io.engine_from_model(m)
    eng.load_batch(seqs, quals)
    fa, ga, ea, na = eng.train_eval(x)
    assert eng.last_timing()[2] == 0
    # serial streams give the same full-size numbers as the concurrent default
    eng.set_option("group_streams", 1)
    eng.set_option("two_streams", 0)
    fs, gs, _, _ = eng.train_eval(x)
    assert fs == pytest.approx(fa, rel=1e-12)
    np.testing.assert_allclose(gs, ga, rtol=1e-9, atol=1e-8)
    eng.set_option("group_streams", 2)
    eng.set_option("two_streams", 1)
    eng.load_batch([seqs[k] for k in rest], [quals[k] for k in rest])
    fr, gr, er, nr = eng.train_eval(x)
    sub = io.engine_from_model(m)
    sub.load_batch([seqs[k] for k in pick], [quals[k] for k in pick])
    f64, g64, e64, n64 = sub.train_eval(x)
    o, xo = po.oracle_from_model(gpath("syn_l1.model"))
    fo, go, eo, no = o.train_eval(xo, [seqs[k] for k in pick], [quals[k] for k in pick], n_threads=min(16, len(os.sched_getaffinity(0))))
    assert f64 == pytest.approx(fo, rel=1e-9) and n64 == no == 0
    np.testing.assert_allclose(g64, go, rtol=1e-7, atol=1e-7)
    assert fa - fr == pytest.approx(fo, rel=1e-7)
    np.testing.assert_allclose(ga - gr, go, rtol=1e-6, atol=1e-6)     # (differences of sums of 10^4 terms)
    assert ea - er == pytest.approx(eo, rel=1e-9)


def test_in_library_collective_with_one_rank():
    """elemdp_comm_init / elemdp_train_eval with a communicator (RCCL, dlopen'ed): a world of one rank all-reduces in place
    and returns the single-rank numbers; a rank without a batch contributes zeros (and gets the zeros of its world of one)."""
    m = io.read_model(gpath("syn_b.model"))
    recs = io.read_fastq(gpath("syn_L150_n8.fq"))
    seqs, quals = [s for _, s, _ in recs], [q for _, _, q in recs]
    eng = io.engine_from_model(m)
    eng.load_batch(seqs, quals)
    ref = eng.train_eval(m["x"])
    uid = api.Engine.comm_unique_id()
    assert len(uid) == 128
    eng.comm_init(0, 1, uid)
    got = eng.train_eval(m["x"])
    assert got[0] == pytest.approx(ref[0], rel=1e-12) and got[2:] == ref[2:]
    np.testing.assert_allclose(got[1], ref[1], rtol=1e-10, atol=1e-12)
    eng.comm_destroy()
    again = eng.train_eval(m["x"])
    assert again[0] == pytest.approx(ref[0], rel=1e-12)
    empty = io.engine_from_model(m)
    with pytest.raises(api.ElemdpError):
        empty.train_eval(m["x"])                  # no batch and no communicator: ELEMDP_ESTATE
    empty.comm_init(0, 1, api.Engine.comm_unique_id())
    fn, gr, eff, nsk = empty.train_eval(m["x"])
    assert fn == 0.0 and eff == 0.0 and nsk == 0 and not np.any(gr)


def test_streamed_batch_gives_the_resident_numbers():
    """option "max_resident": a batch larger than the budget is evaluated / scanned in chunks (BPP filter + plan of the next
    chunk built on a second inner engine while the current one runs) -- same fn / gr (1e-11), per-sequence statistics,
    bpp_eff and scan records as the resident path, on a ragged batch with both labels."""
    m = io.read_model(gpath("syn_b.model"))
    seqs, quals = [], []
    for L, n in ((60, 30), (200, 25), (110, 30), (35, 18)):
        s_, q_ = synth.synth_batch(n, L, seed=500 + L)
        seqs += s_
        quals += q_
    for k in range(0, len(seqs), 7):
        quals[k][-1] = 5
    x = m["x"]
    res = io.engine_from_model(m)
    res.load_batch(seqs, quals)
    ref = res.train_eval(x)
    ref_stats, ref_eff = res.seq_stats(), res.bpp_eff()
    ref_recs, ref_en = res.scan(x)
    eng = io.engine_from_model(m)
    eng.set_option("max_resident", 40)          # 103 sequences -> chunks of 40, 40, 23
    eng.load_batch(seqs, quals)
    got = eng.train_eval(x)
    assert got[0] == pytest.approx(ref[0], rel=1e-11) and got[2] == pytest.approx(ref[2], rel=1e-12) and got[3] == ref[3]
    np.testing.assert_allclose(got[1], ref[1], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(eng.seq_stats(), ref_stats, rtol=1e-11, atol=1e-12)
    np.testing.assert_array_equal(eng.bpp_eff(), ref_eff)
    part = eng.train_partial(x)                  # the partial / finish pair of the multi-GPU path streams, too
    fin = eng.train_finish(part)
    assert fin[0] == pytest.approx(ref[0], rel=1e-11)
    np.testing.assert_allclose(fin[1], ref[1], rtol=1e-11, atol=1e-11)     # (second pass over the chunks: the filter's masks
    np.testing.assert_array_equal(eng.bpp_eff(), ref_eff)                  #  and kept fractions come from the cache)
    recs, en = eng.scan(x)
    for a, b in zip(recs, ref_recs):
        assert (a["Ys"], a["Ye"], a["rss"]) == (b["Ys"], b["Ye"], b["rss"]) and list(a["psihat"]) == list(b["psihat"])
        assert_log_close(a["start"], b["start"], rtol=1e-10, atol=1e-10, what="start")
        assert_log_close(a["end"], b["end"], rtol=1e-10, atol=1e-10, what="end")
        assert a["exist_prob"] == pytest.approx(b["exist_prob"], rel=1e-10)
    np.testing.assert_allclose(en, ref_en, rtol=1e-10, atol=1e-12)
    with pytest.raises(api.ElemdpError):
        eng.pairs(0)                             # needs a resident batch
    eng.set_option("max_resident", 0)
    eng.load_batch(seqs, quals)                  # back to resident
    again = eng.train_eval(x)
    assert again[0] == pytest.approx(ref[0], rel=1e-12)


def test_linear_bpp_filter_equals_the_log_space_filter_and_the_oracle():
    """K1 in the linear semiring without a plan of the unfiltered mask (bpp_kernels.hip, the default) against the log-space
    filter over the item list (option "bpp_log") on a ragged batch: identical kept sets, ln BPP to 1e-9; and against the
    oracle on a sequence long enough (L = 1500) that only the logarithmic exterior chains keep Z in range."""
    seqs, quals = [], []
    for L, n in ((40, 6), (300, 12), (97, 9), (7, 3), (160, 10)):
        s_, q_ = synth.synth_batch(n, L, seed=900 + L)
        seqs += s_
        quals += q_
    res = {}
    for mode in (0, 1):
        eng = api.Engine("(.)", "~T2004~", 50, 30, 1e-4)
        eng.set_option("bpp_log", mode)
        eng.set_option("keep_lnbpp", 1)
        eng.load_batch(seqs, quals)
        res[mode] = ([eng.pairs(k, with_lnbpp=True) for k in range(len(seqs))], eng.bpp_eff())
    for (ka, la), (kb, lb) in zip(res[0][0], res[1][0]):
        assert np.array_equal(ka, kb)
        assert_log_close(la, lb, rtol=1e-9, atol=1e-9, what="lnbpp")
    np.testing.assert_array_equal(res[0][1], res[1][1])
    (long_seq,), (long_q,) = synth.synth_batch(1, 1500, seed=31)
    eng = api.Engine("(.)", "~T2004~", 50, 30, 1e-4)
    eng.set_option("keep_lnbpp", 1)
    eng.load_batch([long_seq], [long_q])
    o = po.make_oracle("(.)", 50, 30, min_bpp=1e-4)
    ln_o, kept_o, eff_o, lnz_o = o.bpp(long_seq)
    kept, ln = eng.pairs(0, with_lnbpp=True)
    assert np.array_equal(kept, kept_o) and eng.bpp_eff()[0] == eff_o
    assert_log_close(ln, ln_o, rtol=1e-9, atol=1e-9, what="lnbpp, L = 1500")


@pytest.mark.parametrize("W,C", [(20, 5), (50, 12), (100, 30), (200, 30), (255, 30), (33, 0)])
def test_bpp_filter_kernels_on_odd_shapes_against_the_oracle(W, C):
    """The filter kernels stage mask rows / bases / first-pair spans of a workgroup in LDS and take candidates four at a time:
    band widths from 20 to 200 (the linear range, kBppLinMaxSpan) and 255 (log-space filter), interior loops capped at 0 .. 30,
    lengths around the workgroup's 32 cells and the band width, N bases, one batch -- kept sets and kept fractions exact, ln BPP
    to 1e-9 against the oracle.  Wide bands also get a poly-GC hairpin that spans the whole band: the heaviest Boltzmann
    weight a band of that width can hold (the linear filter must stay inside the double range up to its cut-off)."""
    rng = np.random.default_rng(1000 * W + C)
    lens = [1, 2, 6, 31, 32, 33, 64, 65, W - 1, W, W + 1, W + 34, 2 * W + 7, 180]
    seqs, quals = [], []
    for k, L in enumerate(lens):
        (s_,), (q_,) = synth.synth_batch(1, max(1, L), seed=int(rng.integers(1 << 30)))
        if k % 3 == 0 and L > 8:
            s_ = s_.copy()
            s_[rng.integers(0, L, size=max(1, L // 15))] = 0          # N
        seqs.append(s_)
        quals.append(q_)
    if W >= 100:
        h = W // 2 - 2
        seqs.append(np.array([3] * h + [1, 1, 1, 1] + [2] * h, dtype=np.uint8))      # G^h AAAA C^h
        quals.append(np.zeros(2 * h + 5, dtype=np.uint8))
    eng = api.Engine("(.)", "~T2004~", W, C, 1e-4)
    eng.set_option("keep_lnbpp", 1)
    eng.load_batch(seqs, quals)
    o = po.make_oracle("(.)", W, C, min_bpp=1e-4)
    eff = eng.bpp_eff()
    for k, s_ in enumerate(seqs):
        ln_o, kept_o, eff_o, _ = o.bpp(s_)
        kept, ln = eng.pairs(k, with_lnbpp=True)
        assert np.array_equal(kept, kept_o), (k, len(s_))
        assert eff[k] == eff_o or (np.isnan(eff[k]) and np.isnan(eff_o))     # (no canonical pair at all: 0 / 0 in both)
        assert_log_close(ln, ln_o, rtol=1e-9, atol=1e-9, what="lnbpp L=%d" % len(s_))


def test_wide_band_against_the_oracle():
    """max_span = 300 (six times the default band): the LDS windows of the band kernels, the pair-mask rows of the filter
    and the exterior-chain staging all scale with the span.  fn / gr of two sequences of L = 330 against the oracle."""
    W = 300
    eng = api.Engine("((.*.))", "~T2004~", W, 30, 1e-4, 0.1, 0, 0)
    seqs, quals = synth.synth_batch(2, 330, seed=11)
    quals[1][-1] = 5
    eng.load_batch(seqs, quals)
    x = eng.initial_params(1.0)
    fn, gr, eff, nsk = eng.train_eval(x)
    o = po.make_oracle("((.*.))", W, 30, min_bpp=1e-4, tau=0.1)
    fo, go, eo, no = o.train_eval(x, seqs, quals)
    assert nsk == no and fn == pytest.approx(fo, rel=1e-9) and eff == pytest.approx(eo, rel=1e-12)
    np.testing.assert_allclose(gr, go, rtol=1e-7, atol=1e-7)

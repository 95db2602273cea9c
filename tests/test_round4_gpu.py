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


def test_default_block_policy_at_size():
    """At a size where the launcher itself takes several blocks per workgroup (more workgroups than a launch keeps resident):
    the default against nblk = 1, and the rate of both."""
    seqs, quals = synth.synth_batch(1024, 200, seed=77)
    eng = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
    eng.load_batch(seqs, quals)
    x = eng.initial_params(1.0)
    a = eng.train_eval(x)
    a = eng.train_eval(x)
    ms_auto = eng.last_timing()[1]
    eng.set_option("nblk", 1)
    b = eng.train_eval(x)
    b = eng.train_eval(x)
    ms_one = eng.last_timing()[1]
    print("1024 x L=200: %.1f ms with the default blocks per workgroup, %.1f ms with one" % (ms_auto, ms_one))
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

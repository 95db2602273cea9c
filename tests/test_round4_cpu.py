"""Round 4, CPU: the TABLE-DRIVEN forms of the band targets -- what k4_in / k4_out run by default (lin_fast.h: unary programs,
weight tables, cell records, ScanFlag words; the pair records and tuple column records of the fast blobs) -- and the merged
outside sweep on the automaton with the shadow copy of state (0,0), compiled into the serial test driver (tests/emul) and
pinned to the oracle on the same cases as the generic rule code.  The GPU-less container thereby checks the path the bench
measures, not only the rules it was derived from."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.emul.pyemul import lib
from tests.test_emul_vs_oracle import CASES, PAR, SCAN, model_pair
from tests.util import assert_log_close, gpath


@pytest.mark.parametrize("model,fq", CASES)
@pytest.mark.parametrize("schedule", [0, 2])
@pytest.mark.parametrize("prune", [0, 1])
def test_table_driven_train_forms_match_oracle(model, fq, schedule, prune):
    """schedule 0: the reference's two outside passes (RNAelemTrainDP::operator(), motif_trainer.hpp:204-227); 2: ONE sweep on
    the automaton with the shadow state, as elemdp_train_eval runs it.  Inside tables, partition functions,
    f, expected counts and energy statistics against the oracle."""
    o, e, x = model_pair(model)
    e.set_prune(prune)
    mask = e.set_fast(1)
    if schedule == 2 and not mask & 8:
        pytest.skip("automaton without a shadow state")
    full = schedule == 0 and fq in ("tiny.fq", "0.fq", "syn_L40_n3.fq") and not prune
    before = lib().emu_fast_cells()
    for rid, seq, qual in po.read_fastq(gpath(fq)):
        a = o.train_seq(seq, qual, tables=full)
        b = e.train_seq(x, seq, qual, tables=full, linear=schedule)
        for k in ("Zo", "Zari", "Znasi"):
            assert_log_close(b[k], a[k], rtol=1e-11, what=k)
        if a["skipped"] or (schedule == 2 and not np.isfinite(a["Znasi"])):
            assert b["skipped"] == 2
            continue
        assert b["skipped"] == 0
        assert b["f"] == pytest.approx(a["f"], rel=1e-10, abs=1e-12)
        for k in ("ENo", "ENx", "EHo", "EHx"):
            np.testing.assert_allclose(b[k], a[k], rtol=1e-9, atol=1e-11, err_msg=k)
        if schedule == 0:
            assert_log_close(b["inside_o"], a["inside_o"], rtol=1e-10, what="inside_o")
            assert_log_close(b["outside_o"], a["outside_o"], rtol=1e-10, what="outside_o")
    took = lib().emu_fast_cells() - before
    fits = mask & (4 if schedule == 2 else 1)
    assert (took > 0) == bool(fits), "table-driven forms taken for %d cells, fp_ok mask %d" % (took, mask)


@pytest.mark.parametrize("model,fq", CASES[:4])
def test_merged_sweep_with_the_generic_rules_matches_oracle(model, fq):
    """The merged sweep alone (generic rule code, worlds chosen per state / per pair): separates a fault of the shadow automaton
    from one of the table-driven forms."""
    o, e, x = model_pair(model)
    e.set_prune(1)
    if not e.set_fast(0) & 8:
        pytest.skip("automaton without a shadow state")
    for rid, seq, qual in po.read_fastq(gpath(fq)):
        a = o.train_seq(seq, qual)
        b = e.train_seq(x, seq, qual, linear=2)
        if a["skipped"] or not np.isfinite(a["Znasi"]):
            continue
        assert b["f"] == pytest.approx(a["f"], rel=1e-10, abs=1e-12)
        for k in ("ENo", "ENx", "EHo", "EHx"):
            np.testing.assert_allclose(b[k], a[k], rtol=1e-9, atol=1e-11, err_msg=k)


@pytest.mark.parametrize("model,fq", SCAN)
@pytest.mark.parametrize("prune", [0, 1])
def test_table_driven_scan_passes_match_oracle(model, fq, prune):
    """The scan's four sum passes through the table-driven forms: start / inner posteriors, the start constraint as cell flags
    and ScanFlag words, end posteriors with the cells left of Ys left out (motif_scanner.hpp:186-252, :546-573, :594-622,
    :715-747); the Viterbi pass stays the generic max-plus code."""
    o, e, x = model_pair(model)
    e.set_prune(prune)
    mask = e.set_fast(1)
    before = lib().emu_fast_cells()
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
    assert (lib().emu_fast_cells() > before) == bool(mask & 1)


# ---- --lik-ratio together with shuffled negatives (the default Adam mode): motif_trainer.hpp:156-202 with :145-152 ----------
from rnaelem_amd import api, io, train   # noqa: E402
from tests.util import gload             # noqa: E402

LIK_SHUFFLE = gload("train_trace_lik_shuffle.json")


@pytest.mark.parametrize("t", LIK_SHUFFLE, ids=["%s-%s-batch%d" % (c["fq"], c["pattern"], c["batch_size"]) for c in LIK_SHUFFLE])
def test_lik_ratio_with_shuffled_negatives_reproduces_the_reference_trace_with_the_oracle(t):
    """`RNAelem train --lik-ratio` without --no-shuffle: every record and its per-iteration negative under the likelihood-ratio
    objective (a negative contributes Z(ari) - Z(ari,nasi)), Adam, whole batch and mini-batches of 2 (into the third epoch).
    The data term and |gr|^2 the reference prints at every iteration, to their 6 printed digits."""
    recs = io.read_fastq(gpath(t["fq"]))
    seqs, quals = [s for _, s, _ in recs], [q for _, _, q in recs]
    o = po.make_oracle(t["pattern"], 50, 30, min_bpp=1e-4, tau=t["tau"], flags=po.LIK_RATIO)

    def ev_batch(s2, q2, x):
        o.set_params(x)
        skipped = np.array([o.train_seq(s, q)["skipped"] for s, q in zip(s2, q2)], dtype=bool)
        return o.train_eval(x, s2, q2, n_threads=4) + (skipped,)

    ev = train.MiniBatches(seqs, quals, t["batch_size"], ev_batch, kmer_shuf=t["kmer_shuf"])
    x0 = api.Engine(t["pattern"]).initial_params(t["lambda_init"])
    r = train.minimize_adam(ev, x0, train.regularisation(len(x0), t["rho_theta"], t["rho_lambda"]), max_iter=t["max_iter"])
    assert len(r["trace"]) == len(t["iter_fn"])
    for row, y, g2 in zip(r["trace"], t["iter_fn"], t["iter_gnorm"]):
        assert row[4] == pytest.approx(y, rel=2e-5, abs=1e-9)
        assert row[2] == pytest.approx(g2, rel=2e-5)


@pytest.mark.parametrize("pattern", ["((.*.))", "(.....)", ".(.).", "(.(.).)", "(.*)", "((((.....))))"])
@pytest.mark.parametrize("prune", [0, 1])
def test_deterministic_mode_tuple_lists(pattern, prune):
    """The tuple lists of the deterministic mode (AutomatonLayout::qd_* / fqd_*): the tuples of every interior-loop column-record
    list dealt to the four waves of a workgroup by target -- each live tuple in exactly one share, the share of its target, dead
    tuples in none, and every list inside the run of the blob its kernel stages (plain, one-state and shadow automaton)."""
    from tests.emul.pyemul import Emul
    e = Emul(pattern, PAR)
    e.set_prune(prune)
    assert lib().emu_check_det_lists(e.h) == 0


def test_candidate_table_of_the_filter_reproduces_loop_weight():
    """The BPP filter's candidate table (bpp_cand.h: coefficient per shape, the other factors folded into planes by the kernels)
    against loop_weight for every shape u1 + u2 <= 30, every pair of pair types and every choice of the four neighbour bases
    incl. N, for both parameter sets: each shape is a special one or exactly one entry, entries sorted by size with the right
    prefix counts, the runs of the generic class repeat the entries, and coefficient x closing factor x inner factor =
    loop_weight to 1e-14 (the same products in another order)."""
    import ctypes
    from tests.emul import pyemul
    from oracle import pyoracle as po
    L = pyemul.lib()
    L.emu_check_bpp_cand.restype = ctypes.c_long
    L.emu_check_bpp_cand.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
    for par in (po.DEFAULT_PAR, os.path.join(os.path.dirname(po.DEFAULT_PAR), "andronescu2007.elempar")):
        if not os.path.exists(par):
            continue
        e = pyemul.Emul("(.)", open(par).read())
        worst = ctypes.c_double(0.)
        assert L.emu_check_bpp_cand(e.h, ctypes.byref(worst)) == 0
        assert worst.value <= 1e-14

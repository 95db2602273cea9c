"""Round-3 GPU tests: the reference's own host bound to the library (oracle/shim), the deterministic reduction mode, the
table-driven train kernels against the generic ones, multi-rank readiness."""
import os
import re
import subprocess

import numpy as np
import pytest

from rnaelem_amd import api, io, synth
from tests.util import REPO, gload, gpath

pytestmark = pytest.mark.gpu

SHIM = os.path.join(REPO, "oracle", "_ref", "RNAelem_gpu")


@pytest.mark.parametrize("case", [0, 1])
def test_reference_host_on_the_library_reproduces_the_reference_lbfgsb_run(case, tmp_path):
    """oracle/_ref/RNAelem_gpu = the reference's option parser, bounds, regulariser, Lbfgsb (the embedded L-BFGS-B 2.1,
    optimizer.hpp:262-334) and model writer compiled UNCHANGED from /root/reference, with RNAelemTrainer::operator()
    (motif_trainer.hpp:595-633) replaced by elemdp_load_batch + elemdp_train_eval (oracle/shim/motif_trainer_gpu.hpp, the shim of
    INTEGRATION.md).  The binary without a sub-command trains (--no-shuffle), writes the model and scans the training set with
    the reference's scanner.  Against the same run of the reference binary (tests/golden/train_final.json): the objective after
    EVERY iteration to the digits the optimizer prints, the parameters of the final model as the writer prints them, and the
    scan records under the final model (config A: material/positive.fa as FASTQ, pattern (.....); and its first six records)."""
    if not os.path.exists(SHIM):
        pytest.skip("oracle/_ref/RNAelem_gpu is built where /root/reference exists (make -C oracle ref_gpu)")
    t = gload("train_final.json")[case]
    m, raw = str(tmp_path / "m.model"), str(tmp_path / "scan.raw")
    r = subprocess.run([SHIM, "--fastq", gpath(t["fq"]), "--motif-pattern", t["pattern"], "--out1", m, "--out2", raw, "--max-iter", str(t["max_iter"]),
                        "--no-shuffle", "--batch-size", "-1", "-t", "4", "--lambda-init", str(t["lambda_init"]), "--epsilon", "%g" % t["epsilon"]],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    f = [float(x.group(2)) for x in re.finditer(r"^iter: (\d+) , f: ([-0-9.e+]+)", r.stdout + r.stderr, re.M)]
    assert len(f) == len(t["iter_f"]), (f, t["iter_f"])
    # five significant digits are printed: one unit of the last one
    np.testing.assert_allclose(f, t["iter_f"], rtol=2e-5, atol=0)
    x = io.read_model(m)["x"]
    # the writer prints six significant digits: one unit of the last one (measured: identical text, tools/shim_control.py --
    # and the reference binary's scan of the two model FILES gives identical records)
    np.testing.assert_allclose(x, t["x"], rtol=2e-6, atol=2e-7)
    from tests.golden.make_golden import parse_scan
    recs = parse_scan(open(raw).read())
    assert len(recs) == len(t["records"])
    # (the reference's scanner writes the records in the order its threads finish them)
    # The two models agree to the six digits the writer prints.  The Viterbi parse is a maximum over alternatives, and these
    # models end with lambda[1] = 0 (no energy term on the motif's states): foldings AWAY from the motif tie or nearly tie, so
    # that part of the structure string follows the seventh digit of the parameters -- in the reference itself, too: its
    # in-process scan differs from its scan of the model file it wrote in 58 of 76 records (tools/shim_control.py).  Compared
    # exactly: the motif occurrence of every record -- start, end, and the stretch of the structure / alignment / state strings
    # it covers.
    n_other = 0
    for a, b in zip(sorted(recs, key=lambda r: r["id"]), sorted(t["records"], key=lambda r: r["id"])):
        assert (a["id"], a["Ys"], a["Ye"]) == (b["id"], b["Ys"], b["Ye"])
        assert a["exist_prob"] == pytest.approx(b["exist_prob"], rel=1e-3, abs=1e-6)
        lo, hi = a["Ys"], a["Ye"]          # (Ye = the position behind the last motif base)
        assert a["rss"][lo:hi] == b["rss"][lo:hi] and a["mot"][lo:hi] == b["mot"][lo:hi] and a["psihat"][lo:hi] == b["psihat"][lo:hi], a["id"]
        n_other += (a["rss"], a["mot"], a["psihat"]) != (b["rss"], b["mot"], b["psihat"])
    print("records whose parse differs away from the motif: %d of %d" % (n_other, len(recs)))


def test_python_lbfgsb_loop_converges_where_the_reference_converges_on_config_a():
    """The shipped host loop (rnaelem_amd/train.py: SciPy's L-BFGS-B 3.0) on config A to convergence, against the reference's
    converged run (the embedded L-BFGS-B 2.1; tests/golden/train_converged.json): final regularised objective rel 1e-4,
    parameters abs 1e-3, and the same Ys / Ye for every training record under the trained model."""
    from rnaelem_amd import train
    t = gload("train_converged.json")[1]
    recs = io.read_fastq(gpath(t["fq"]))
    seqs, quals = [s for _, s, _ in recs], [q for _, _, q in recs]
    eng = api.Engine(t["pattern"], "~T2004~", 50, 30, 1e-4, t["tau"])
    eng.load_batch(seqs, quals)
    x0 = eng.initial_params(t["lambda_init"])
    r = train.train(eng.train_eval, x0, t["rho_theta"], t["rho_lambda"], max_iter=400, epsilon=t["epsilon"])
    assert r["message"].startswith("CONVERGENCE"), r["message"]
    assert r["f"] == pytest.approx(t["final_f"], rel=1e-4)
    np.testing.assert_allclose(r["x"], t["x"], rtol=0, atol=1e-3)
    got, _ = eng.scan(np.asarray(r["x"]))
    ref = {rec["id"].strip(): rec for rec in t["records"]}
    for (rid, _, _), a in zip(recs, got):
        assert (a["Ys"], a["Ye"]) == (ref[rid.strip()]["Ys"], ref[rid.strip()]["Ye"]), rid


def test_deterministic_mode_gives_bit_identical_evaluations():
    """Option "deterministic" = 1: every sum that lanes of different waves share gets a copy per wave, added up in wave order;
    the expected counts of a workgroup go to a row of its own (sequence, block), summed in block order (k4_combine); the sum
    over sequences is a fixed tree (k_reduce).  Two evaluations of the same batch -- 10 000 x L=200, BASELINE config C, with
    table slots reused in between -- are then bit-identical, as the reference's are at --thread 1 (motif_trainer.hpp:248-271);
    the default mode (LDS / global fp64 atomics) agrees with it to 1e-11."""
    eng = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
    seqs, quals = synth.synth_batch(10000, 200)
    for k in range(0, 10000, 3):
        quals[k][-1] = 5                       # a third of the records without the motif
    eng.load_batch(seqs, quals)
    x = eng.initial_params(1.0)
    x[:-2] += np.linspace(-0.2, 0.2, len(x) - 2)
    ref = eng.train_eval(x)
    eng.set_option("deterministic", 1)
    a = eng.train_eval(x)
    eng.train_eval(x + 0.01)                   # (other values through the same table slots and count rows)
    b = eng.train_eval(x)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3]
    assert a[0] == pytest.approx(ref[0], rel=1e-11)
    np.testing.assert_allclose(a[1], ref[1], rtol=1e-9, atol=1e-9)
    ms_det = eng.last_timing()[1]
    eng.set_option("deterministic", 0)
    eng.train_eval(x)
    print("deterministic mode: %.1f ms per evaluation against %.1f ms" % (ms_det, eng.last_timing()[1]))
    # the generic kernels (option fast = 0) take the same switches
    eng2 = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
    eng2.set_option("fast", 0)
    eng2.set_option("deterministic", 1)
    eng2.load_batch(seqs[:300], quals[:300])
    c, d = eng2.train_eval(x), eng2.train_eval(x)
    assert c[0] == d[0] and np.array_equal(c[1], d[1])


def test_table_driven_train_kernels_equal_the_generic_ones():
    """Option "fast" (default 1: per-state programs, pair records, weight tables, cell records -- lin_fast.h) against the generic
    rule code (the one the CPU emulation pins to the oracle): fn, gr, kept fractions on a ragged batch, both schedules."""
    seqs, quals = [], []
    for L, n in ((40, 5), (200, 9), (97, 7), (130, 6)):
        s_, q_ = synth.synth_batch(n, L, seed=700 + L)
        seqs += s_
        quals += q_
    for k in range(0, len(quals), 2):
        quals[k][-1] = 5
    res = {}
    for fast in (0, 1):
        for sched in (0, 1):
            eng = api.Engine("((.*.))", "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
            eng.set_option("fast", fast)
            eng.set_option("schedule", sched)
            eng.load_batch(seqs, quals)
            x = eng.initial_params(0.7)
            x[:-2] += np.linspace(-0.3, 0.3, len(x) - 2)
            res[(fast, sched)] = eng.train_eval(x)
    for key, r in res.items():
        assert r[0] == pytest.approx(res[(0, 0)][0], rel=1e-10), key
        np.testing.assert_allclose(r[1], res[(0, 0)][1], rtol=1e-8, atol=1e-9, err_msg=str(key))
        assert r[2] == res[(0, 0)][2] and r[3] == res[(0, 0)][3]


@pytest.mark.parametrize("pattern", ["((.*.))", "(.....)", ".(.).", "(.(.).)"])
def test_table_driven_scan_passes_equal_the_generic_ones(pattern):
    """The scan's four sum passes on the table-driven kernels (scanner node tests as ScanFlag words, lin_fast.h) against the
    generic rule code (option fast = 0; pinned to the oracle by tests/emul and tests/test_gpu_parity.py): posteriors of start,
    inner and end positions, Ys / Ye, the parse and E[N] on a ragged batch with masked positions."""
    seqs, quals = [], []
    for L, n in ((35, 4), (180, 6), (97, 5), (64, 5)):
        s_, q_ = synth.synth_batch(n, L, seed=900 + L)
        seqs += s_
        quals += q_
    for k in range(0, len(quals), 3):
        quals[k][len(quals[k]) // 2] = 5
    out = {}
    for fast in (0, 1):
        eng = api.Engine(pattern, "~T2004~", 50, 30, 1e-4, 0.1, 0, 0)
        eng.set_option("fast", fast)
        eng.load_batch(seqs, quals)
        x = eng.initial_params(0.7)
        x[:-2] += np.linspace(-0.4, 0.4, len(x) - 2)
        out[fast] = eng.scan(x)
    (r0, en0), (r1, en1) = out[0], out[1]
    np.testing.assert_allclose(en1, en0, rtol=1e-9, atol=1e-12)
    for a_, b_ in zip(r0, r1):
        assert (a_["Ys"], a_["Ye"]) == (b_["Ys"], b_["Ye"])
        assert a_["rss"] == b_["rss"] and np.array_equal(a_["psihat"], b_["psihat"])
        assert b_["exist_prob"] == pytest.approx(a_["exist_prob"], rel=1e-9)
        for key in ("start", "inner", "end"):
            fa, fb = np.isfinite(a_[key]), np.isfinite(b_[key])
            assert np.array_equal(fa, fb), key
            np.testing.assert_allclose(b_[key][fb], a_[key][fa], rtol=1e-8, atol=1e-9, err_msg=key)


def test_in_library_collective_with_two_ranks(tmp_path):
    """elemdp_comm_init(rank, 2, id) + elemdp_train_eval on two GPUs, one process each (tests/two_rank_worker.py): the batch is
    sharded by assigned_range, every rank's evaluation all-reduces the partial vector (ncclAllReduce on the engine's stream,
    librccl dlopen'ed) and both return the numbers of the single-engine evaluation of the whole batch.  Skipped on a box with
    one GPU; exists so that the first multi-GPU lease exercises the in-library RCCL path beyond a world of one."""
    import json
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    m = io.read_model(gpath("syn_b.model"))
    recs = io.read_fastq(gpath("syn_L150_n8.fq"))
    eng = io.engine_from_model(m, device=0)
    eng.load_batch([s for _, s, _ in recs], [q for _, _, q in recs])
    ref = eng.train_eval(m["x"])
    del eng
    uid_file = str(tmp_path / "uid.bin")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "two_rank_worker.py"), str(r), "2", uid_file, str(tmp_path / ("out%d.json" % r))],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    for r in range(2):
        got = json.load(open(str(tmp_path / ("out%d.json" % r))))
        assert got["fn"] == pytest.approx(ref[0], rel=1e-10) and got["eff"] == pytest.approx(ref[2], rel=1e-12) and got["nsk"] == ref[3]
        np.testing.assert_allclose(got["gr"], ref[1], rtol=1e-9, atol=1e-10)


def test_andronescu_2007_parameters_on_the_gpu_against_the_reference():
    """~A2007~ (energy_model.hpp:155-160) through the whole hot path -- BPP filter, plan, train evaluation -- against
    RNAelemTrainer::operator() of the compiled reference, sequence by sequence (tests/golden/dp_A2007.json)."""
    from tests.util import assert_log_close
    case = gload("dp_A2007.json")[0]
    m = io.read_model(gpath(case["model"]))
    assert m["ene_param"] == "~A2007~"
    recs = io.read_fastq(gpath(case["fq"]))
    eng = io.engine_from_model(m)
    for (rid, seq, qual), r in zip(recs, case["seqs"]):
        eng.load_batch([seq], [qual])
        fn, gr, eff, nsk = eng.train_eval(m["x"])
        st = eng.seq_stats()[0]
        for k, name in enumerate(("Zo", "Zari", "Znasi")):
            assert_log_close(st[k], r[name], rtol=1e-10, what=name)
        assert fn == pytest.approx(r["f"], rel=1e-9, abs=1e-10) and eff == pytest.approx(r["bpp_eff"], rel=1e-12)
        g_ref = np.r_[np.array([v for row in r["ENo"] for v in row]) - np.array([v for row in r["ENx"] for v in row]),
                      np.array(r["EHo"]) - np.array(r["EHx"])]
        np.testing.assert_allclose(gr, g_ref, rtol=1e-7, atol=1e-8)


def test_mask_trainer_with_adam_on_the_gpu_reproduces_the_reference_trace():
    """--param-set with the default optimizer (Adam, shuffled negatives; motif_mask_trainer.hpp:66-108) with the GPU as the
    evaluator: the objective trace and the final parameters of `RNAelem train --param-set` (train_trace_mask.json), the
    parameters outside the set untouched."""
    from rnaelem_amd import cli, train
    t = [c for c in gload("train_trace_mask.json") if not c["no_shuffle"]][0]
    recs = io.read_fastq(gpath(t["fq"]))
    seqs, quals = [s for _, s, _ in recs], [q for _, _, q in recs]
    pos = api.Engine(t["pattern"], "~T2004~", 50, 30, 1e-4, 0.1)
    neg = api.Engine(t["pattern"], "~T2004~", 50, 30, 1e-4, 0.1)
    pos.load_batch(seqs, quals)

    def ev_batch(s2, q2, x):
        neg.load_batch(s2, q2)
        return neg.train_eval(x)

    vary = cli.parse_param_set(t["param_set"])
    x0 = pos.initial_params(t["lambda_init"])
    ev = train.ShuffledNegatives(seqs, pos.train_eval, lambda: pos.seq_stats()[:, 4] != 0, ev_batch, k=2)
    r = train.train(ev, x0, 0.1, 0.1, max_iter=t["max_iter"], optimizer="adam", vary=vary)
    fn = [row[4] for row in r["trace"]]
    assert len(fn) == len(t["trace"])
    for a, b in zip(fn, t["trace"]):
        assert a == pytest.approx(b, rel=2e-5), (fn, t["trace"])
    np.testing.assert_allclose(r["x"], t["final_x"], rtol=2e-5, atol=2e-6)
    fixed = [i for i in range(len(x0)) if i not in vary]
    assert np.array_equal(r["x"][fixed], x0[fixed])

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

"""Round-3 CPU tests: fixtures of tests/golden/make_golden_r3.py against the oracle, host-side pieces of the new options."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.util import assert_log_close, gload, gpath


def test_oracle_with_the_andronescu_2007_parameters_against_the_reference():
    """fn and the expected counts of every sequence under ~A2007~ (energy_model.hpp:155-160) against RNAelemTrainer::operator() of
    the compiled reference (tests/golden/dp_A2007.json): the second energy parameter set takes the same path as ~T2004~."""
    case = gload("dp_A2007.json")[0]
    o, x = po.oracle_from_model(gpath(case["model"]))
    recs = po.read_fastq(gpath(case["fq"]))
    for (rid, seq, qual), r in zip(recs, case["seqs"]):
        g = o.train_seq(seq, qual)
        for k in ("Zo", "Zari", "Znasi"):
            assert_log_close(g[k], r[k], rtol=1e-13, what=k)
        assert g["f"] == pytest.approx(r["f"], rel=1e-12, abs=1e-13)
        np.testing.assert_allclose(g["ENo"], [v for row in r["ENo"] for v in row], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(g["ENx"], [v for row in r["ENx"] for v in row], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(g["EHo"], r["EHo"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(g["EHx"], r["EHx"], rtol=1e-10, atol=1e-12)

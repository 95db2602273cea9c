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


def test_loop_weight_is_the_exponential_of_loop_energy():
    """The BPP filter takes the weight of a candidate interior loop from pre-exponentiated energy tables (energy_rules.h:
    loop_weight, exp_tables) instead of exp(loop_energy): both forms of the reference's EnergyParam::loop_energy
    (energy_param.hpp:744-795) must agree on every kind of loop -- stack, bulges with and without the stacking term, 1x1 / 2x1 /
    1x2 / 2x2 tables, generic interior loops of the three mismatch classes, and log 0 beyond 30 unpaired bases."""
    import math
    from tests.emul.pyemul import Emul
    from oracle import pyoracle as po
    e = Emul("(.)", po.energy_param_text("~T2004~"))
    rng = np.random.RandomState(5)
    kinds = set()
    n = 0
    for _ in range(4000):
        L = 60
        s = rng.randint(1, 5, L).astype(np.uint8)
        i = rng.randint(0, 10)
        u1 = int(rng.choice([0, 0, 1, 1, 2, 2, 3, 4, 7, 12, 20, 31]))
        u2 = int(rng.choice([0, 0, 1, 1, 2, 2, 3, 5, 9, 15, 31]))
        p = i + 1 + u1
        q = p + 4
        j = q + 1 + u2
        if j >= L:
            continue
        a, b = e.loop_energy(s, i, j, p, q), e.loop_weight(s, i, j, p, q)
        kinds.add((min(u1, 3), min(u2, 3), u1 + u2 > 30))
        if a == -math.inf:
            assert b == 0.0
        else:
            assert b == pytest.approx(math.exp(a), rel=4e-15)
        n += 1
    assert n > 3000 and len(kinds) >= 17


@pytest.mark.parametrize("pattern", ["((.*.))", "(.....)", ".(.).", "(.(.).)", "..."])
def test_scan_flag_words_follow_the_scanner_node_tests(pattern):
    """The table-driven scan passes replace the scanner functors' tests on the nodes of an emitting transition
    (motif_scanner.hpp:546-573 start / inner, :594-622 start constraint, :715-747 end posteriors; allow_* / lstat_* in the
    generic rule code) by one flag word per forward transition of the flattened automaton: re-derive every word from the nodes."""
    from tests.emul.pyemul import Emul
    rows, M = Emul(pattern, po.energy_param_text("~T2004~")).scan_flags()
    assert len(rows) > 0
    SL, SR, IL, IR, EL, ER, PM2 = 1, 2, 4, 8, 16, 32, 64
    for kind, pl, pr, cl, cr, fl in rows:
        want = ((SL if (pl == 0 and cl == 1) else 0) | (SR if (cr == 0 and pr == 1) else 0) | (IL if cl not in (0, M - 1) else 0) |
                (IR if pr not in (0, M - 1) else 0) | (EL if (pl == M - 2 and cl == M - 1) else 0) |
                (ER if (cr == M - 2 and pr == M - 1) else 0) | (PM2 if pr == M - 2 else 0))
        assert fl == want, (kind, pl, pr, cl, cr, fl, want)
    # a right transition moves only the r-node, a left one only the l-node (the flags of the other side never fire for it)
    assert all(pl == cl for kind, pl, pr, cl, cr, fl in rows if kind == 0)
    assert all(pr == cr for kind, pl, pr, cl, cr, fl in rows if kind == 1)

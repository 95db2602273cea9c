"""Shared helpers for the test-suite (golden loading, tolerant comparisons)."""
import json
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")


def gpath(name):
    return os.path.join(GOLDEN, name)


def gload(name):
    return json.load(open(gpath(name)))


def arr(x):
    """JSON list (with -Infinity / NaN) -> float64 array."""
    return np.array(x, dtype=np.float64)


def assert_log_close(a, b, rtol=1e-9, atol=1e-9, what=""):
    """Compare log-space arrays: identical -inf pattern, finite values within tolerance."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    ia, ib = np.isneginf(a), np.isneginf(b)
    assert np.array_equal(ia, ib), "%s: -inf pattern differs at %s" % (what, np.argwhere(ia != ib)[:5].tolist())
    fa, fb = a[~ia], b[~ib]
    if fa.size:
        err = np.abs(fa - fb) / (atol / rtol + np.abs(fb))
        assert err.max() <= rtol, "%s: max rel err %.3e" % (what, err.max())

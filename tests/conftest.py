import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
# The compact tables of the scaled-linear pipeline hold garbage wherever an entry is structurally zero (nothing is stored
# there).  Under the test-suite every evaluation starts from tables filled with NaN, so that a read of such an entry that is
# not masked by the reader poisons the result instead of passing as a zero left over from a fresh allocation.
os.environ.setdefault("ELEMDP_POISON", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """GPU runs: let torch initialise its HIP runtime BEFORE libelemdp creates its first stream, as bench.py and the command
    line do (torch ships its own ROCm libraries; with the order reversed its device query has been seen to come back empty)."""
    if "not gpu" in (config.getoption("markexpr", "") or ""):
        return
    if any(item.get_closest_marker("gpu") for item in items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.zeros(1, device="cuda")
        except Exception:      # noqa: BLE001  (no torch / no GPU: the gpu tests will say so themselves)
            pass

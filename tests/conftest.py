import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")
    config.addinivalue_line("markers", "slow: the long tail of an exhaustive parametrisation (deselect with -m 'gpu and not slow' for a quick pass)")


@pytest.fixture(scope="session")
def oracle_factory():
    """kktsolver factory backed by the CPU oracle (test infrastructure)."""
    from oracle.kkt_oracle import OracleKKTSolver

    def fac(P, A, cones, m, n, settings, ordering="mmd"):
        return OracleKKTSolver(P, A, cones, m, n, settings, ordering=ordering)

    return fac


@pytest.fixture(autouse=True)
def _hipkkt_env_is_restored():
    """Every HIPKKT_* switch a test sets (or leaks) is gone again for the next test: the suite must give the same result as one
    process in file order (the driver's `pytest -x -q -m gpu`) and spread over xdist workers."""
    before = {k: v for k, v in os.environ.items() if k.startswith("HIPKKT_")}
    yield
    for k in [k for k in os.environ if k.startswith("HIPKKT_")]:
        if k not in before:
            del os.environ[k]
    os.environ.update(before)

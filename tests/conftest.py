import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The suite runs against the TESTING build of the library (libclarabel_hipkkt_testing.so: the product's sources + the switches that
# tests compare a mechanism's on / off states with).  HIPKKT_TEST_PRODUCTION=1 runs it against the production library instead; tests
# that need a switch are then skipped (clarabel.jl_amd/hipkkt.py sync_debug_switches refuses, the fixture below turns that into a skip).
PRODUCTION = os.environ.get("HIPKKT_TEST_PRODUCTION", "0") == "1"
os.environ["CLARABEL_HIPKKT_TESTING"] = "0" if PRODUCTION else "1"


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


@pytest.fixture(autouse=True)
def _switches_need_the_testing_build(monkeypatch):
    if not PRODUCTION:
        yield
        return
    from clarabel_jl_amd import hipkkt

    def refuse():
        if any(os.environ.get("HIPKKT_" + k) is not None for k in hipkkt.DEBUG_KEYS):
            pytest.skip("this test sets a switch of the testing build; running against the production library")

    monkeypatch.setattr(hipkkt, "sync_debug_switches", refuse)
    yield

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_usable():
    """True when libdiffqc_hip.so loads and sees at least one device."""
    try:
        from diffquantum_amd.engine import device_count
        return device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a GPU (or without the built library) skips the gpu-marked tests instead of
    erroring in each of them.  When the gpu tests are asked for explicitly (`-m gpu`, as the GPU box runs them) nothing is
    skipped: a missing library or device must fail loudly there."""
    if "gpu" in (config.getoption("-m") or ""):
        return
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if gpu_items and not _gpu_usable():
        skip = pytest.mark.skip(reason="no MI355X / libdiffqc_hip.so here (run with -m gpu on the GPU box)")
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement of the reference path (test infrastructure, see oracle/trotter_oracle.py), behind the memo of its slow
    evaluations (tests/oracle_memo.py: answers of byte-identical earlier calls come from tests/golden/oracle_memo.npz)."""
    from oracle import trotter_oracle
    from oracle_memo import OracleMemo
    memo = OracleMemo(trotter_oracle)
    yield memo
    memo.flush()
    if memo.hits or memo.misses:
        print("\noracle memo: {} calls answered from the fixture ({:.0f} s of oracle time), {} computed".format(
            memo.hits, memo.seconds_saved, memo.misses))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")

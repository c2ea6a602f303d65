import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)          # tests/bounds.py


# the library's test hooks (ssym_comm_inject_fault, ssym_comm_replay_bounds) answer only to a process that asks for them
os.environ.setdefault("SSYM_TEST_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; the product never imports it)."""
    import oracle as _o
    return _o.load()


@pytest.fixture(scope="session")
def native_lib():
    import soundsym_amd._native as nat
    if not os.path.exists(nat.LIB_PATH):
        nat.build()
    return nat.lib()

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture
def make_oracle(oracle_mod):
    made = []

    def _mk():
        e = oracle_mod.OracleEngine()
        made.append(e)
        return e
    yield _mk
    for e in made:
        e.close()


@pytest.fixture
def make_gpu():
    """Factory of product engines (HIP path through the C ABI). No fall-back: fails if the
    extension is missing or no gfx950 device is present."""
    from chemlab_amd.engine import Engine
    made = []

    def _mk(precision=64):
        e = Engine(device=0, precision=precision)
        made.append(e)
        return e
    yield _mk
    for e in made:
        e.close()


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)

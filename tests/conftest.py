import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def unit_rows(n, d, seed):
    import numpy as np
    rng = np.random.Generator(np.random.Philox(seed))
    x = rng.standard_normal((n, d), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


@pytest.fixture(scope="session")
def ssa():
    """The product package; importing it loads libssmi.so (fails loudly if missing)."""
    import semantic_search_amd
    return semantic_search_amd

import os
import sys
from pathlib import Path

import pytest

# the seeded sweeps draw from default_rng(base + seed + SALT): SSMI_FUZZ_SALT=<n> re-runs them on other cases (a soak run;
# the committed suite runs with 0)
SALT = 100000 * int(os.environ.get("SSMI_FUZZ_SALT", "0"))

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


def _poison_free_hbm():
    """Fill what is free of HBM with 0xFF bytes and give it back: whatever the library allocates afterwards
    holds NaNs (as fp16 / fp32), -1 (as ints) and the largest possible key, not the zeros of a fresh box,
    so a kernel that reads memory the library never wrote fails a parity test instead of passing by luck.
    (Found this way in round 2: an uninitialised score floor that only showed after another job had used the GPU.)"""
    import torch
    if not torch.cuda.is_available() or os.environ.get("SSMI_TEST_NO_POISON"):
        return
    free, _ = torch.cuda.mem_get_info()
    left = max(0, min(free - (4 << 30), 240 << 30))
    blocks = []
    while left >= (1 << 30):
        n = min(left, 32 << 30)
        try:
            blocks.append(torch.empty(n, dtype=torch.uint8, device="cuda").fill_(255))
        except RuntimeError:
            break
        left -= n
    torch.cuda.synchronize()
    del blocks
    torch.cuda.empty_cache()


@pytest.fixture(scope="session")
def ssa():
    """The product package; importing it loads libssmi.so (fails loudly if missing)."""
    import semantic_search_amd
    _poison_free_hbm()
    return semantic_search_amd

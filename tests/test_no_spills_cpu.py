"""No hot kernel may spill: a spilled register comes back through scratch behind `s_waitcnt vmcnt(0)`, which
drains the kernel's memory pipeline (the per-query IVF scan lost a fifth of its bandwidth that way until the
spill was found).  tools/audit_scratch.py reads the private-segment size of every kernel from the built objects."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "semantic-search_amd", "csrc", "build")

ALLOWED = ()  # no exceptions


def test_kernels_have_no_scratch():
    if not os.path.isdir(BUILD) or not any(f.endswith(".o") for f in os.listdir(BUILD)):
        pytest.skip("library not built")
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("llvm tools not available")
    spec = importlib.util.spec_from_file_location("audit_scratch", os.path.join(ROOT, "tools", "audit_scratch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ks = mod.kernels_with_scratch(BUILD)
    assert len(ks) > 100
    bad = {k: v for k, v in ks.items() if v and not any(a in k for a in ALLOWED)}
    assert not bad, bad

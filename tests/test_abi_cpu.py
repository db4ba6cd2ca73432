"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol
include/ssmi.h declares, and reports errors without a GPU (no compute calls)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "ssmi.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ssmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(ssa):
    from semantic_search_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert "ssmi_index_search" in names and len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"libssmi.so does not export {name}"
    # and the Python binding table covers the header exactly
    assert sorted(_lib.EXPORTED_SYMBOLS) == names


def test_abi_version_and_error_string(ssa):
    from semantic_search_amd import _lib
    lib = _lib.load()
    assert lib.ssmi_abi_version() == 1
    h = ctypes.c_void_p()
    rc = lib.ssmi_index_create_flat(0, 0, 0, ctypes.byref(h))  # invalid dimension: fails before touching a device
    assert rc != 0 and b"dimension" in lib.ssmi_last_error()
    with pytest.raises(RuntimeError, match="dimension"):
        _lib.check(rc)
    assert lib.ssmi_index_free(None) == 0


def test_missing_library_fails_loudly(monkeypatch, ssa):
    from semantic_search_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("SSMI_LIB", "/nonexistent/libssmi.so")
    with pytest.raises(RuntimeError, match="only compute path"):
        _lib.load()


def test_product_never_imports_the_oracle():
    for path in (ROOT / "semantic-search_amd").rglob("*"):
        if path.suffix in (".py", ".hip", ".h") and path.is_file():
            assert "oracle" not in path.read_text(), f"{path} mentions the oracle"

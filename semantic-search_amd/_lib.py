"""ctypes binding of libssmi.so (the C-ABI declared in include/ssmi.h).

The HIP library is the only compute path: if it is missing or fails to load this
module raises, it never falls back to a CPU implementation.

torch is imported first on purpose: the PyTorch-ROCm wheel bundles its own HIP
runtime (soname libamdhip64.so.7); loading it before libssmi.so makes the
dynamic loader reuse that one runtime for both, so torch device pointers and
streams are valid inside libssmi.
"""
import ctypes
import os
from pathlib import Path

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "csrc" / "libssmi.so"

F32, F16, BF16 = 0, 1, 2
HOST, DEVICE = 0, 1
KIND_FLAT, KIND_IVFFLAT = 0, 1
MAX_K = 1024

_c_index_p = ctypes.c_void_p


class Timing(ctypes.Structure):
    _fields_ = [
        ("last_search_ms", ctypes.c_double),
        ("last_scan_ms", ctypes.c_double),
        ("last_merge_ms", ctypes.c_double),
        ("last_scan_bytes", ctypes.c_int64),
        ("last_scan_flops", ctypes.c_int64),
        ("last_path", ctypes.c_int),
        ("last_grid", ctypes.c_int),
    ]


# name -> (argtypes); every function returns int except ssmi_last_error
_SIGNATURES = {
    "ssmi_abi_version": [],
    "ssmi_device_count": [ctypes.POINTER(ctypes.c_int)],
    "ssmi_index_create_flat": [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_c_index_p)],
    "ssmi_index_create_ivfflat": [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_c_index_p)],
    "ssmi_index_free": [_c_index_p],
    "ssmi_index_get_kind": [_c_index_p, ctypes.POINTER(ctypes.c_int)],
    "ssmi_index_get_d": [_c_index_p, ctypes.POINTER(ctypes.c_int)],
    "ssmi_index_get_ntotal": [_c_index_p, ctypes.POINTER(ctypes.c_int64)],
    "ssmi_index_get_is_trained": [_c_index_p, ctypes.POINTER(ctypes.c_int)],
    "ssmi_index_get_nlist": [_c_index_p, ctypes.POINTER(ctypes.c_int)],
    "ssmi_index_get_nprobe": [_c_index_p, ctypes.POINTER(ctypes.c_int)],
    "ssmi_index_set_nprobe": [_c_index_p, ctypes.c_int],
    "ssmi_index_get_storage_dtype": [_c_index_p, ctypes.POINTER(ctypes.c_int)],
    "ssmi_index_get_device": [_c_index_p, ctypes.POINTER(ctypes.c_int)],
    "ssmi_index_train": [_c_index_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int],
    "ssmi_index_set_centroids": [_c_index_p, ctypes.c_void_p, ctypes.c_int],
    "ssmi_index_get_centroids": [_c_index_p, ctypes.c_void_p],
    "ssmi_index_add": [_c_index_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int],
    "ssmi_index_add_ivf_shard": [_c_index_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                 ctypes.c_int, ctypes.c_int],
    "ssmi_index_assign": [_c_index_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p],
    "ssmi_index_add_preassigned": [_c_index_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                   ctypes.c_void_p],
    "ssmi_index_add_random_unit": [_c_index_p, ctypes.c_int64, ctypes.c_uint64],
    "ssmi_index_search": [_c_index_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                          ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p],
    "ssmi_index_search_min": [_c_index_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p],
    "ssmi_index_search_static": [_c_index_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p],
    "ssmi_static_scratch_bytes": [ctypes.POINTER(ctypes.c_size_t)],
    "ssmi_topk_merge": [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                        ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    "ssmi_index_reconstruct_n": [_c_index_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p],
    "ssmi_index_reconstruct_n_dev": [_c_index_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p],
    "ssmi_index_score_ids": [_c_index_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p],
    "ssmi_index_make_direct_map": [_c_index_p],
    "ssmi_index_write": [_c_index_p, ctypes.c_char_p],
    "ssmi_index_write_append": [_c_index_p, ctypes.c_char_p],
    "ssmi_index_read": [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_c_index_p)],
    "ssmi_index_get_timing": [_c_index_p, ctypes.POINTER(Timing)],
    "ssmi_index_get_scan_times": [_c_index_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)],
    "ssmi_index_set_shadow_policy": [_c_index_p, ctypes.c_int],
    "ssmi_index_get_shadow_state": [_c_index_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64),
                                    ctypes.POINTER(ctypes.c_int)],
    "ssmi_index_get_shadow_dtype": [_c_index_p, ctypes.POINTER(ctypes.c_int)],
    "ssmi_index_get_last_kernel": [_c_index_p, ctypes.POINTER(ctypes.c_char_p)],
    "ssmi_index_get_kernel_clock": [_c_index_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)],
    "ssmi_index_set_timing": [_c_index_p, ctypes.c_int],
    "ssmi_index_get_list_sizes": [_c_index_p, ctypes.c_void_p],
    "ssmi_index_get_assignments": [_c_index_p, ctypes.c_void_p],
    "ssmi_index_get_list_members": [_c_index_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.POINTER(ctypes.c_int64)],
}
EXPORTED_SYMBOLS = ["ssmi_last_error"] + sorted(_SIGNATURES)

_lib = None


def load() -> ctypes.CDLL:
    """Load libssmi.so once; raise (never fall back) when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("SSMI_LIB", LIB_PATH))
    if not path.exists():
        raise RuntimeError(
            f"{path} not found: the HIP extension is the only compute path of semantic_search_amd. "
            "Build it with `python -m semantic_search_amd.build` (needs hipcc, no GPU required).")
    lib = ctypes.CDLL(str(path))
    lib.ssmi_last_error.restype = ctypes.c_char_p
    lib.ssmi_last_error.argtypes = []
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = argtypes
    if lib.ssmi_abi_version() != 1:
        raise RuntimeError("libssmi.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def check(rc: int) -> None:
    """Nonzero return code -> RuntimeError, the exception type the reference catches
    around FAISS calls (core/index_factory.py:275,376; core/index_manager.py:610)."""
    if rc != 0:
        msg = load().ssmi_last_error()
        raise RuntimeError(msg.decode("utf-8", "replace") if msg else f"libssmi error {rc}")

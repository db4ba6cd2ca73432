"""ctypes loader for the C restatement (oracle/flat_ip.c).  TEST INFRASTRUCTURE."""
import ctypes
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = _HERE / "_build" / "liboracle.so"


def build():
    subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)
    return _LIB


def _load():
    if not _LIB.exists():
        build()
    lib = ctypes.CDLL(str(_LIB))
    lib.oracle_flat_ip_search.restype = ctypes.c_int
    lib.oracle_flat_ip_search.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.oracle_num_threads.restype = ctypes.c_int
    return lib


def num_threads() -> int:
    return int(_load().oracle_num_threads())


def search(X: np.ndarray, Q: np.ndarray, k: int):
    X = np.ascontiguousarray(X, dtype=np.float32)
    Q = np.ascontiguousarray(np.atleast_2d(np.asarray(Q, dtype=np.float32)))
    nq = Q.shape[0]
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    rc = _load().oracle_flat_ip_search(X.ctypes.data, X.shape[0], X.shape[1] if X.ndim == 2 else Q.shape[1],
                                       Q.ctypes.data, nq, k, D.ctypes.data, I.ctypes.data)
    if rc:
        raise RuntimeError(f"oracle_flat_ip_search failed rc={rc}")
    return D, I

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


_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB.exists():
        build()
    lib = ctypes.CDLL(str(_LIB))
    _lib = lib
    lib.oracle_flat_ip_search.restype = ctypes.c_int
    lib.oracle_flat_ip_search.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.oracle_num_threads.restype = ctypes.c_int
    lib.oracle_ivf_flat_search.restype = ctypes.c_int
    lib.oracle_ivf_flat_search.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
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


class GroupedIVF:
    """Rows grouped by list for oracle/ivf_flat.c: build once, search many times (the CPU IVF baseline)."""

    def __init__(self, X: np.ndarray, C: np.ndarray, labels: np.ndarray, ids=None):
        X = np.asarray(X, dtype=np.float32)
        labels = np.asarray(labels, dtype=np.int64)
        self.C = np.ascontiguousarray(C, dtype=np.float32)
        nlist = self.C.shape[0]
        order = np.argsort(labels, kind="stable")
        self.Xs = np.ascontiguousarray(X[order])
        self.ids = np.ascontiguousarray((np.arange(len(X), dtype=np.int64) if ids is None else np.asarray(ids, np.int64))[order])
        self.off = np.zeros((nlist + 1,), np.int64)
        np.cumsum(np.bincount(labels, minlength=nlist), out=self.off[1:])

    def search(self, Q: np.ndarray, k: int, nprobe: int):
        Q = np.ascontiguousarray(np.atleast_2d(np.asarray(Q, dtype=np.float32)))
        nq, nlist, d = Q.shape[0], self.C.shape[0], self.C.shape[1]
        nprobe = min(nprobe, nlist)
        D = np.empty((nq, k), np.float32); I = np.empty((nq, k), np.int64)
        P = np.empty((nq, nprobe), np.int64)
        rc = _load().oracle_ivf_flat_search(self.C.ctypes.data, nlist, d, self.Xs.ctypes.data, self.ids.ctypes.data,
                                            self.off.ctypes.data, Q.ctypes.data, nq, nprobe, k, D.ctypes.data,
                                            I.ctypes.data, P.ctypes.data)
        if rc:
            raise RuntimeError(f"oracle_ivf_flat_search failed rc={rc}")
        return D, I, P

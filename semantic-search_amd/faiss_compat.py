"""`faiss`-shaped Python facade over libssmi.so (HIP kernels on MI355X).

The reference never subclasses or registers anything with FAISS: its boundary is
the duck type of `faiss.Index` as used by core/index_factory.py and
core/index_manager.py.  This module re-creates exactly that
surface:

    IndexFlatIP(d)                                  core/index_factory.py:140,144
    IndexIVFFlat(quantizer, d, nlist, metric)       core/index_factory.py:145
    .d .ntotal .is_trained  (.nlist .nprobe on IVF) core/index_factory.py:169-170,223-233
    .add(x) .train(x) .search(x, k) .reconstruct(i) core/index_manager.py:299,319,371,609
    read_index(path) / write_index(index, path)     core/index_manager.py:185,241
    METRIC_INNER_PRODUCT                            core/index_factory.py:145

Class names matter: IndexFactory.detect_index_type (core/index_factory.py:206-217)
sniffs the substrings "IVFPQ" / "IVFFlat" / "HNSW" in type(index).__name__.
Flat objects deliberately have no `nlist` / `nprobe` / `hnsw` attributes, because
get_index_stats / set_search_params branch on hasattr (core/index_factory.py:230-236,317-320).

Additive (not in the reference): `storage="f16"` / `storage="bf16"` row storage (bf16 is LOSSLESS for the
reference's embeddings, which are bf16 model outputs cast to fp32: core/model_wrapper.py:201-203), torch tensors on a
ROCm device accepted by add/search (device pointers are handed straight to the
kernels), `search(..., id_offset=)` for row-sharded databases, `reconstruct_batch`.

All engine failures surface as RuntimeError (what SWIG-FAISS raises and the
reference catches).  There is no CPU fallback.
"""
import ctypes
import os
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib

_NO_FLOOR = -3.4028234663852886e38  # -FLT_MAX: ssmi_index_search_min's "no score floor"
METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1

_STORAGE = {"f32": _lib.F32, "fp32": _lib.F32, "float32": _lib.F32, "f16": _lib.F16, "fp16": _lib.F16,
            "float16": _lib.F16, "bf16": _lib.BF16, "bfloat16": _lib.BF16}
_STORAGE_NAME = {_lib.F32: "f32", _lib.F16: "f16", _lib.BF16: "bf16"}
# torch dtypes the engine takes as they are (anything else is widened to float32 first)
_TORCH_SRC = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


def _device_rows(x: torch.Tensor):
    """(contiguous tensor, source dtype code) of a device tensor of vectors."""
    if x.dtype not in _TORCH_SRC:
        x = x.float()
    return x.contiguous(), _TORCH_SRC[x.dtype]


def _floor32(min_score) -> float:
    """The fp32 floor the kernels compare with, chosen so that `score_f32 >= floor` decides exactly what the
    reference's `r.score >= min_score` decides in double precision (nodes/search_nodes.py:152-159): the SMALLEST
    float32 that is >= min_score.  Plain narrowing rounds to nearest, which keeps a score one ulp below a floor
    that rounded down (and drops one equal to a floor that rounded up)."""
    if min_score is None:
        return _NO_FLOOR
    m = float(min_score)
    if m != m:
        raise RuntimeError("search: min_score is NaN")
    f = np.float32(m)
    if np.isfinite(f) and float(f) < m:
        f = np.nextafter(f, np.float32(np.inf), dtype=np.float32)
    return max(float(f), _NO_FLOOR) if np.isfinite(f) else (_NO_FLOOR if m < 0 else float(np.finfo(np.float32).max))


def _default_device() -> int:
    env = os.environ.get("SSMI_DEVICE")
    if env is not None:
        return int(env)
    if torch.cuda.is_available():
        return int(torch.cuda.current_device())
    return 0


def _as_f32_matrix(x, d: int, what: str) -> np.ndarray:
    x = np.asarray(x)
    if x.ndim == 1:
        x = x.reshape(1, -1)
    if x.ndim != 2:
        raise ValueError(f"{what}: expected a 2-D array, got shape {x.shape}")
    # the FAISS SWIG wrapper asserts on the dimension
    assert x.shape[1] == d, f"{what}: dimension {x.shape[1]} != index dimension {d}"
    return np.ascontiguousarray(x, dtype=np.float32)


class _IndexBase:
    """Common handle plumbing of the two index classes."""

    metric_type = METRIC_INNER_PRODUCT

    def __init__(self):
        self._h = ctypes.c_void_p()
        self._lib = _lib.load()

    # -- lifetime ---------------------------------------------------------
    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), ctypes.c_void_p()
        if h:
            self._lib.ssmi_index_free(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _handle(self):
        if not self._h:
            raise RuntimeError("index has been closed")
        return self._h

    # -- attributes -------------------------------------------------------
    def _get_int(self, fn) -> int:
        out = ctypes.c_int()
        _lib.check(fn(self._handle(), ctypes.byref(out)))
        return int(out.value)

    @property
    def d(self) -> int:
        return self._get_int(self._lib.ssmi_index_get_d)

    @property
    def ntotal(self) -> int:
        out = ctypes.c_int64()
        _lib.check(self._lib.ssmi_index_get_ntotal(self._handle(), ctypes.byref(out)))
        return int(out.value)

    @property
    def is_trained(self) -> bool:
        return bool(self._get_int(self._lib.ssmi_index_get_is_trained))

    @property
    def device(self) -> int:
        return self._get_int(self._lib.ssmi_index_get_device)

    @property
    def storage(self) -> str:
        return _STORAGE_NAME[self._get_int(self._lib.ssmi_index_get_storage_dtype)]

    # -- add / train ------------------------------------------------------
    def train(self, x) -> None:
        if isinstance(x, torch.Tensor) and x.is_cuda:
            self._check_tensor(x, "train")
            x = x.contiguous().float()
            torch.cuda.current_stream(x.device).synchronize()
            _lib.check(self._lib.ssmi_index_train(self._handle(), x.shape[0], x.data_ptr(), _lib.DEVICE))
            return
        if isinstance(x, torch.Tensor):
            x = x.numpy()
        x = _as_f32_matrix(x, self.d, "train")
        _lib.check(self._lib.ssmi_index_train(self._handle(), x.shape[0], x.ctypes.data, _lib.HOST))

    def _check_tensor(self, x: torch.Tensor, what: str) -> None:
        if x.dim() != 2:
            raise ValueError(f"{what}: expected a 2-D tensor, got shape {tuple(x.shape)}")
        assert x.shape[1] == self.d, f"{what}: dimension {x.shape[1]} != index dimension {self.d}"
        if x.device.index != self.device:
            raise RuntimeError(f"{what}: tensor is on {x.device}, index lives on cuda:{self.device}")

    def add(self, x) -> None:
        """Append vectors; ids are the running row numbers (FAISS add() without ids)."""
        if isinstance(x, torch.Tensor) and x.is_cuda:
            self._check_tensor(x, "add")
            x, dt = _device_rows(x)
            torch.cuda.current_stream(x.device).synchronize()  # the library copies on its own stream
            _lib.check(self._lib.ssmi_index_add(self._handle(), x.shape[0], x.data_ptr(), dt, _lib.DEVICE))
            return
        if isinstance(x, torch.Tensor):
            x = x.numpy()
        x = _as_f32_matrix(x, self.d, "add")
        _lib.check(self._lib.ssmi_index_add(self._handle(), x.shape[0], x.ctypes.data, _lib.F32, _lib.HOST))

    def add_random_unit(self, n: int, seed: int = 1234) -> None:
        """Bench/test helper: n rows of N(0,1) components, L2-normalised, generated on the device."""
        _lib.check(self._lib.ssmi_index_add_random_unit(self._handle(), int(n), int(seed)))

    # -- search -----------------------------------------------------------
    def search(self, x, k: int, id_offset: int = 0, out=None, min_score: Optional[float] = None):
        """(D, I) = top-k inner products of each query row.

        min_score (additive; the search nodes' filter, nodes/search_nodes.py:152-159): only results with
        score >= min_score are reported, the rest of a row is -1 / -FLT_MAX padding; applied inside the kernels.

        numpy in -> numpy out (D float32 (n,k) sorted descending, I int64, -1 padded).
        A torch tensor on the index's device -> torch tensors on that device,
        enqueued on the current torch stream (no host synchronisation).
        """
        k = int(k)
        if k < 1:
            raise RuntimeError("search: k must be >= 1")
        floor = _floor32(min_score)
        if isinstance(x, torch.Tensor) and x.is_cuda:
            if x.dim() == 1:
                x = x.reshape(1, -1)
            self._check_tensor(x, "search")
            x = x.contiguous().float()
            nq = x.shape[0]
            if out is not None:  # caller-provided contiguous (nq,k) device tensors (e.g. views of one gather buffer)
                D, I = out
                if (tuple(D.shape) != (nq, k) or tuple(I.shape) != (nq, k) or D.dtype != torch.float32 or
                        I.dtype != torch.int64 or not D.is_contiguous() or not I.is_contiguous() or D.device != x.device):
                    raise ValueError("search: out must be contiguous (nq,k) float32 / int64 tensors on the index's device")
            else:
                D = torch.empty((nq, k), dtype=torch.float32, device=x.device)
                I = torch.empty((nq, k), dtype=torch.int64, device=x.device)
            stream = torch.cuda.current_stream(x.device).cuda_stream
            _lib.check(self._lib.ssmi_index_search_min(self._handle(), nq, x.data_ptr(), _lib.DEVICE, k, floor, D.data_ptr(),
                                                       I.data_ptr(), _lib.DEVICE, int(id_offset), stream))
            return D, I
        if isinstance(x, torch.Tensor):
            x = x.numpy()
        x = _as_f32_matrix(x, self.d, "search")
        nq = x.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        _lib.check(self._lib.ssmi_index_search_min(self._handle(), nq, x.ctypes.data, _lib.HOST, k, floor, D.ctypes.data,
                                                   I.ctypes.data, _lib.HOST, int(id_offset), None))
        return D, I

    def search_static(self, x: torch.Tensor, k: int, out, id_offset: int = 0) -> None:
        """Allocation-free search (ssmi_index_search_static; no counterpart in the reference): x (nq,d) float32 and
        out=(D,I) are contiguous tensors on the index's device; enqueued on the current torch stream.  Nothing is
        allocated, on the device or by torch: the kernels' scratch is a fixed per-GPU array inside libssmi, shared
        by every call on that GPU (overlapping calls race)."""
        D, I = out
        nq = x.shape[0]
        if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2):
            raise ValueError("search_static: x must be a contiguous 2-D float32 tensor on the index's device")
        if (tuple(D.shape) != (nq, k) or tuple(I.shape) != (nq, k) or D.dtype != torch.float32 or I.dtype != torch.int64
                or not D.is_contiguous() or not I.is_contiguous() or D.device != x.device or I.device != x.device):
            raise ValueError("search_static: out must be contiguous (nq,k) float32 / int64 tensors on the index's device")
        self._check_tensor(x, "search_static")
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(self._lib.ssmi_index_search_static(self._handle(), nq, x.data_ptr(), int(k), D.data_ptr(),
                                                      I.data_ptr(), int(id_offset), stream))

    # -- read back --------------------------------------------------------
    def reconstruct_batch(self, ids) -> np.ndarray:
        ids = np.ascontiguousarray(np.asarray(ids, dtype=np.int64).reshape(-1))
        out = np.empty((ids.shape[0], self.d), dtype=np.float32)
        _lib.check(self._lib.ssmi_index_reconstruct_n(self._handle(), ids.shape[0], ids.ctypes.data, out.ctypes.data))
        return out

    def reconstruct_batch_device(self, ids) -> torch.Tensor:
        """Stored vectors `ids` as a float32 (n,d) tensor on the index's GPU (no host copy of the vectors)."""
        ids = np.ascontiguousarray(np.asarray(ids, dtype=np.int64).reshape(-1))
        out = torch.empty((ids.shape[0], self.d), dtype=torch.float32, device=torch.device("cuda", self.device))
        torch.cuda.current_stream(out.device).synchronize()
        _lib.check(self._lib.ssmi_index_reconstruct_n_dev(self._handle(), ids.shape[0], ids.ctypes.data, out.data_ptr()))
        return out

    def score_ids(self, ids, vectors) -> np.ndarray:
        """Inner products of the stored vectors `ids` with each row of `vectors`: (len(ids), len(vectors)) float32,
        computed on the device (the exclusion filter of core/index_manager.py:604-621 in one call)."""
        ids = np.ascontiguousarray(np.asarray(ids, dtype=np.int64).reshape(-1))
        v = _as_f32_matrix(vectors, self.d, "score_ids")
        out = np.empty((ids.shape[0], v.shape[0]), dtype=np.float32)
        _lib.check(self._lib.ssmi_index_score_ids(self._handle(), ids.shape[0], ids.ctypes.data, v.shape[0], v.ctypes.data,
                                                  out.ctypes.data))
        return out

    def reconstruct(self, i: int) -> np.ndarray:
        """Stored vector i as float32 (d,).  RuntimeError when it cannot be produced,
        which is what the reference catches (core/index_factory.py:275)."""
        return self.reconstruct_batch([int(i)])[0]

    def reconstruct_n(self, i0: int = 0, ni: int = -1) -> np.ndarray:
        if ni < 0:
            ni = self.ntotal - i0
        return self.reconstruct_batch(np.arange(i0, i0 + ni, dtype=np.int64))

    # -- timing of the last search (bench / tests) --------------------------
    def last_timing(self) -> dict:
        t = _lib.Timing()
        _lib.check(self._lib.ssmi_index_get_timing(self._handle(), ctypes.byref(t)))
        return {name: getattr(t, name) for name, _ in _lib.Timing._fields_}


    # -- fp16 shadow of fp32 rows (large batches on the matrix cores) ---------
    def _shadow_state(self):
        pol, rows, failed = ctypes.c_int(), ctypes.c_int64(), ctypes.c_int()
        _lib.check(self._lib.ssmi_index_get_shadow_state(self._handle(), ctypes.byref(pol), ctypes.byref(rows), ctypes.byref(failed)))
        return int(pol.value), int(rows.value), bool(failed.value)

    @property
    def fp16_shadow(self) -> bool:
        """True (default): batches of >= 9 queries over fp32 rows may build an fp16 copy of the rows (+50 % HBM) and
        run on the matrix cores with fp16-rounded operands.  False: every batch stays on the exact fp32 scan."""
        return self._shadow_state()[0] != 0

    @fp16_shadow.setter
    def fp16_shadow(self, enabled: bool) -> None:
        _lib.check(self._lib.ssmi_index_set_shadow_policy(self._handle(), 1 if enabled else 0))

    @property
    def fp16_shadow_rows(self) -> int:
        """Rows the shadow holds right now (0: not built)."""
        return self._shadow_state()[1]

    @property
    def shadow_dtype(self) -> Optional[str]:
        """Element type of the shadow: "f16", "bf16" (chosen while every fp32 row is bf16-valued: the shadow is then an
        exact copy), or None while there is none."""
        dt = self._get_int(self._lib.ssmi_index_get_shadow_dtype)
        return None if dt < 0 else _STORAGE_NAME[dt]

    @property
    def fp16_shadow_failed(self) -> bool:
        """The last attempt to build the shadow ran out of HBM; searches fell back to the exact scan."""
        return self._shadow_state()[2]

    def last_kernel(self) -> str:
        """Name of the dominant kernel of the last search (as rocprofv3 shows it)."""
        out = ctypes.c_char_p()
        _lib.check(self._lib.ssmi_index_get_last_kernel(self._handle(), ctypes.byref(out)))
        return (out.value or b"").decode()

    def last_kernel_clock(self) -> dict:
        """Shader clock under the last search's 256-query MFMA kernel: {"mhz", "cycles", "ms"} (RuntimeError when the last
        search ran another kernel)."""
        mhz, cyc, ms = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        _lib.check(self._lib.ssmi_index_get_kernel_clock(self._handle(), ctypes.byref(mhz), ctypes.byref(cyc), ctypes.byref(ms)))
        return {"mhz": float(mhz.value), "cycles": int(cyc.value), "ms": float(ms.value)}

    def set_timing(self, enabled: bool) -> None:
        """Per-search hipEvent records on (default) or off: each costs the stream ~6 us, so serving loops switch them off."""
        _lib.check(self._lib.ssmi_index_set_timing(self._handle(), 1 if enabled else 0))

    def scan_times_ms(self, max_n: int = 256) -> np.ndarray:
        """hipEvent durations of the dominant kernel of the last searches (oldest first)."""
        out = np.zeros((max_n,), dtype=np.float32)
        n = ctypes.c_int()
        _lib.check(self._lib.ssmi_index_get_scan_times(self._handle(), int(max_n), out.ctypes.data, ctypes.byref(n)))
        return out[: n.value].copy()


class IndexFlatIP(_IndexBase):
    """Exact maximum-inner-product index (stand-in for faiss.IndexFlatIP)."""

    def __new__(cls, d: int = 0, storage: str = "f32", device: Optional[int] = None, fp16_shadow: bool = True,
                devices=None, _handle=None):
        """devices=[...] (additive): ONE index over several GPUs inside this process - rows sharded, results merged on
        the first device (multi_device.MultiDeviceFlatIP); a device may be named several times (logical shards)."""
        if devices is not None and cls is IndexFlatIP:
            from .multi_device import MultiDeviceFlatIP
            return MultiDeviceFlatIP(d, devices, storage=storage, fp16_shadow=fp16_shadow)
        return super().__new__(cls)

    def __init__(self, d: int, storage: str = "f32", device: Optional[int] = None, fp16_shadow: bool = True,
                 devices=None, _handle=None):
        super().__init__()
        if _handle is not None:
            self._h = _handle
            return
        dev = _default_device() if device is None else int(device)
        h = ctypes.c_void_p()
        _lib.check(self._lib.ssmi_index_create_flat(int(d), _STORAGE[storage], dev, ctypes.byref(h)))
        self._h = h
        if not fp16_shadow:
            self.fp16_shadow = False


class IndexIVFFlat(_IndexBase):
    """Inverted-file index with exact inner products inside the probed lists
    (stand-in for faiss.IndexIVFFlat with METRIC_INNER_PRODUCT)."""

    def __new__(cls, quantizer=None, d: int = 0, nlist: int = 0, metric: int = METRIC_INNER_PRODUCT, storage: str = "f32",
                device: Optional[int] = None, fp16_shadow: bool = True, devices=None, _handle=None):
        """devices=[...] (additive): the inverted lists sharded over several GPUs inside this process, list l on shard
        l % len(devices) (multi_device.MultiDeviceIVFFlat)."""
        if devices is not None and cls is IndexIVFFlat:
            from .multi_device import MultiDeviceIVFFlat
            return MultiDeviceIVFFlat(quantizer, d, nlist, metric, devices, storage=storage, fp16_shadow=fp16_shadow)
        return super().__new__(cls)

    def __init__(self, quantizer, d: int, nlist: int, metric: int = METRIC_INNER_PRODUCT, storage: str = "f32",
                 device: Optional[int] = None, fp16_shadow: bool = True, devices=None, _handle=None):
        super().__init__()
        self.quantizer = quantizer
        if _handle is not None:
            self._h = _handle
            return
        if metric != METRIC_INNER_PRODUCT:
            raise RuntimeError("IndexIVFFlat: only METRIC_INNER_PRODUCT is supported")
        if quantizer is not None and getattr(quantizer, "d", d) != d:
            raise RuntimeError("IndexIVFFlat: quantizer dimension differs from d")
        if device is None:
            device = quantizer.device if isinstance(quantizer, _IndexBase) else _default_device()
        h = ctypes.c_void_p()
        _lib.check(self._lib.ssmi_index_create_ivfflat(int(d), int(nlist), _STORAGE[storage], int(device),
                                                       ctypes.byref(h)))
        self._h = h
        if not fp16_shadow:
            self.fp16_shadow = False

    @property
    def nlist(self) -> int:
        return self._get_int(self._lib.ssmi_index_get_nlist)

    @property
    def nprobe(self) -> int:
        return self._get_int(self._lib.ssmi_index_get_nprobe)

    @nprobe.setter
    def nprobe(self, v: int) -> None:
        _lib.check(self._lib.ssmi_index_set_nprobe(self._handle(), int(v)))

    def add_shard(self, x, id0: int, shard: int, nshards: int) -> None:
        """List-sharded add: keep the vectors of x whose list l has l % nshards == shard, with global
        ids id0 + i (every rank of a sharded index passes the same x)."""
        if isinstance(x, torch.Tensor) and x.is_cuda:
            self._check_tensor(x, "add_shard")
            x, dt = _device_rows(x)
            torch.cuda.current_stream(x.device).synchronize()
            _lib.check(self._lib.ssmi_index_add_ivf_shard(self._handle(), x.shape[0], x.data_ptr(), dt, _lib.DEVICE,
                                                          int(id0), int(shard), int(nshards)))
            return
        if isinstance(x, torch.Tensor):
            x = x.numpy()
        x = _as_f32_matrix(x, self.d, "add_shard")
        _lib.check(self._lib.ssmi_index_add_ivf_shard(self._handle(), x.shape[0], x.ctypes.data, _lib.F32, _lib.HOST,
                                                      int(id0), int(shard), int(nshards)))

    def _vectors_arg(self, x, what: str):
        """(keepalive, pointer, n, src dtype, mem kind) of a numpy array / CPU tensor / device tensor of vectors."""
        if isinstance(x, torch.Tensor) and x.is_cuda:
            self._check_tensor(x, what)
            x, dt = _device_rows(x)
            torch.cuda.current_stream(x.device).synchronize()  # the library works on its own stream
            return x, x.data_ptr(), x.shape[0], dt, _lib.DEVICE
        if isinstance(x, torch.Tensor):
            x = x.numpy()
        x = _as_f32_matrix(x, self.d, what)
        return x, x.ctypes.data, x.shape[0], _lib.F32, _lib.HOST

    def assign(self, x) -> np.ndarray:
        """List (max-inner-product centroid) of every row of x, int32 (n,): the assignment step of add()
        on its own, for the routed multi-GPU add (sharded.ShardedIVFFlat.add_local)."""
        keep, ptr, n, dt, mk = self._vectors_arg(x, "assign")
        out = np.empty((n,), dtype=np.int32)
        _lib.check(self._lib.ssmi_index_assign(self._handle(), n, ptr, dt, mk, out.ctypes.data))
        return out

    def add_preassigned(self, x, labels, ids) -> None:
        """Append rows of x to the given lists under the given (global) ids: the owner's half of the routed add."""
        keep, ptr, n, dt, mk = self._vectors_arg(x, "add_preassigned")
        labels = np.ascontiguousarray(np.asarray(labels, dtype=np.int32).reshape(-1))
        ids = np.ascontiguousarray(np.asarray(ids, dtype=np.int64).reshape(-1))
        if labels.shape[0] != n or ids.shape[0] != n:
            raise ValueError("add_preassigned: labels and ids must have one entry per vector")
        _lib.check(self._lib.ssmi_index_add_preassigned(self._handle(), n, ptr, dt, mk, labels.ctypes.data, ids.ctypes.data))

    def make_direct_map(self, _enable: bool = True) -> None:
        """Like faiss: without a direct map reconstruct() raises RuntimeError."""
        _lib.check(self._lib.ssmi_index_make_direct_map(self._handle()))

    def set_centroids(self, centroids) -> None:
        c = np.ascontiguousarray(np.asarray(centroids, dtype=np.float32))
        if c.shape != (self.nlist, self.d):
            raise ValueError(f"centroids must have shape {(self.nlist, self.d)}")
        _lib.check(self._lib.ssmi_index_set_centroids(self._handle(), c.ctypes.data, _lib.HOST))

    def get_centroids(self) -> np.ndarray:
        out = np.empty((self.nlist, self.d), dtype=np.float32)
        _lib.check(self._lib.ssmi_index_get_centroids(self._handle(), out.ctypes.data))
        return out

    def list_sizes(self) -> np.ndarray:
        out = np.empty((self.nlist,), dtype=np.int64)
        _lib.check(self._lib.ssmi_index_get_list_sizes(self._handle(), out.ctypes.data))
        return out


    def list_members(self, l: int, with_vectors: bool = True):
        """(ids int64 (n,), vectors float32 (n,d) or None) of inverted list l, in stored order."""
        n = ctypes.c_int64()
        _lib.check(self._lib.ssmi_index_get_list_members(self._handle(), int(l), 0, None, None, ctypes.byref(n)))
        ids = np.empty((n.value,), dtype=np.int64)
        vecs = np.empty((n.value, self.d), dtype=np.float32) if with_vectors else None
        if n.value:
            _lib.check(self._lib.ssmi_index_get_list_members(self._handle(), int(l), n.value, ids.ctypes.data,
                                                             vecs.ctypes.data if with_vectors else None, ctypes.byref(n)))
        return ids, vecs

    def assignments(self) -> np.ndarray:
        """List id of every stored vector, indexed by vector id (test introspection)."""
        out = np.empty((self.ntotal,), dtype=np.int32)
        _lib.check(self._lib.ssmi_index_get_assignments(self._handle(), out.ctypes.data))
        return out


def write_index(index: _IndexBase, path, incremental: bool = False) -> None:
    """Stand-in for faiss.write_index (core/index_manager.py:241).  incremental=True: if `path` already holds an
    earlier state of this index, only the vectors added since are written: a Flat index appends them to the file
    (same bytes as a full write); an IVF index appends a record to `<path>.journal`, which read_index replays on
    top of the base file (a full write, incremental=False, folds the journal back in and removes it)."""
    if hasattr(index, "shards"):   # one index over several GPUs: ONE FAISS file, written from what the shards hold
        index.write(path)
        return
    fn = index._lib.ssmi_index_write_append if incremental else index._lib.ssmi_index_write
    _lib.check(fn(index._handle(), os.fsencode(str(path))))


def read_index(path, storage: Optional[str] = None, device: Optional[int] = None, devices=None):
    """Stand-in for faiss.read_index (core/index_manager.py:185).  devices=[...]: the file's vectors are spread over
    several GPUs as they are read (multi_device.read_multi)."""
    if devices is not None:
        from .multi_device import read_multi
        return read_multi(path, devices, storage=storage or "f32")
    lib = _lib.load()
    dev = _default_device() if device is None else int(device)
    h = ctypes.c_void_p()
    sd = -1 if storage is None else _STORAGE[storage]
    _lib.check(lib.ssmi_index_read(os.fsencode(str(path)), sd, dev, ctypes.byref(h)))
    kind = ctypes.c_int()
    _lib.check(lib.ssmi_index_get_kind(h, ctypes.byref(kind)))
    if kind.value == _lib.KIND_IVFFLAT:
        return IndexIVFFlat(None, 0, 0, _handle=h)
    return IndexFlatIP(0, _handle=h)


def topk_merge(D_lists: torch.Tensor, I_lists: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge `nlists` result sets (nlists, nq, k) on a ROCm device into the (nq, k) top-k of
    their union (the consumer of the RCCL all-gather).  The list dimension may be strided
    (views into one gathered buffer); the inner (nq, k) block must be contiguous."""
    if D_lists.dim() != 3 or D_lists.shape != I_lists.shape:
        raise ValueError("topk_merge: expected (nlists, nq, k) tensors of equal shape")
    if not D_lists.is_cuda or D_lists.device != I_lists.device:
        raise RuntimeError("topk_merge: tensors must live on the same ROCm device")
    if D_lists.dtype != torch.float32 or I_lists.dtype != torch.int64:
        D_lists, I_lists = D_lists.float(), I_lists.long()
    nl, nq, k = D_lists.shape

    def inner_ok(t):
        return t.stride(2) == 1 and t.stride(1) == k and (nl == 1 or t.stride(0) >= nq * k)
    if not inner_ok(D_lists):
        D_lists = D_lists.contiguous()
    if not inner_ok(I_lists):
        I_lists = I_lists.contiguous()
    D = torch.empty((nq, k), dtype=torch.float32, device=D_lists.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=D_lists.device)
    stream = torch.cuda.current_stream(D_lists.device).cuda_stream
    _lib.check(_lib.load().ssmi_topk_merge(D_lists.device.index, nl, nq, k, D_lists.data_ptr(),
                                           D_lists.stride(0) if nl > 1 else nq * k, I_lists.data_ptr(),
                                           I_lists.stride(0) if nl > 1 else nq * k, D.data_ptr(), I.data_ptr(), stream))
    return D, I

"""One index over several GPUs of a node, inside ONE process: the faiss-shaped surface the reference's single
process can hold (its global index cache, core/index_manager.py:1481-1504, lives in the ComfyUI process; there is
no torchrun there).  SURVEY.md 8e "process model": one shard + one HIP stream + one host thread per device.

    IndexFlatIP(d, devices=[0, 1, 2, 3])                         -> MultiDeviceFlatIP   (rows sharded)
    IndexIVFFlat(quantizer, d, nlist, metric, devices=[...])     -> MultiDeviceIVFFlat  (lists sharded, list l on shard l % S)

Flat: add() sends each batch whole to the least-full shard and records a segment (global start, length, shard,
local start), which keeps the reference's implicit sequential ids (core/index_manager.py:319-325) whatever the
routing; a search runs on every shard at once (each on its own stream, issued by its own host thread), local ids
become global ids through the shard's segment table (one bucketize on the shard's GPU), the (B, k) results travel
to the first device as peer-to-peer copies over xGMI (12*B*k bytes per shard: latency-bound) and ssmi_topk_merge
merges them there - the same merge kernel the one-process-per-GPU form (sharded.py) feeds from its RCCL all-gather.
IVF: centroids are trained on the first shard and installed on all; add() assigns on the first shard
(ssmi_index_assign) and routes each vector to the shard that owns its list (ssmi_index_add_preassigned, global
ids); every shard probes the same lists and scans the ones it owns.

A device may be named more than once: `devices=[0, 0, 0]` gives three logical shards on GPU 0, which is how the
logic is tested on a one-GPU box.  Measured on one GPU only: no multi-GPU lease was available to this repository
(DESIGN.md section 3).  An RCCL form of the gather (ncclCommInitAll + grouped all-gather) is not built: the peer
copies move the same 12*B*k bytes without a communicator.
"""
import struct
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from . import faiss_compat as fc

MAX_GLOBAL_IDS = 0xFFFFFFFF  # merge keys carry 32-bit ids (csrc/common.h)
# FAISS index header behind the fourcc: d:i32 | ntotal:i64 | two i64 | is_trained:u8 | metric_type:i32 (0 = inner product)
_HEADER = struct.Struct("<iqqqBi")


def _as_rows(x, d: int, what: str):
    """numpy (n,d) float32, or a torch tensor left where it is."""
    if isinstance(x, torch.Tensor):
        if x.dim() == 1:
            x = x.reshape(1, -1)
        if x.dim() != 2:
            raise ValueError(f"{what}: expected a 2-D tensor, got shape {tuple(x.shape)}")
        assert x.shape[1] == d, f"{what}: dimension {x.shape[1]} != index dimension {d}"
        return x
    return fc._as_f32_matrix(x, d, what)


class _MultiDeviceBase:
    metric_type = fc.METRIC_INNER_PRODUCT

    def _init_shards(self, shards: List[fc._IndexBase], devices: Sequence[int]) -> None:
        self.shards = shards
        self.devices = [int(v) for v in devices]
        self._tdev = [torch.device("cuda", v) for v in self.devices]
        self._streams = [torch.cuda.Stream(device=t) for t in self._tdev]
        # one host thread per shard: each issues its shard's launches (ctypes releases the GIL inside the library)
        self._pool = ThreadPoolExecutor(max_workers=len(shards), thread_name_prefix="ssmi-shard")
        self._lock = threading.RLock()

    # -- what every shard answers alike
    @property
    def d(self) -> int:
        return self.shards[0].d

    @property
    def device(self) -> int:
        """Where results are merged and device-side outputs live: the first device."""
        return self.devices[0]

    @property
    def storage(self) -> str:
        return self.shards[0].storage

    @property
    def nshards(self) -> int:
        return len(self.shards)

    def set_timing(self, enabled: bool) -> None:
        for s in self.shards:
            s.set_timing(enabled)

    def last_kernel(self) -> str:
        return self.shards[0].last_kernel()

    def close(self) -> None:
        for s in getattr(self, "shards", []):
            s.close()
        pool = getattr(self, "_pool", None)
        if pool is not None:
            pool.shutdown(wait=True)
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the search both kinds share: every shard at once, gather on the first device, merge
    def _to_global(self, si: int, I: torch.Tensor) -> torch.Tensor:
        return I

    def _shard_search(self, si: int, q0: torch.Tensor, q_host, k: int, floor, ready: torch.cuda.Event):
        dev, st = self._tdev[si], self._streams[si]
        with torch.cuda.device(dev), torch.cuda.stream(st):
            if q_host is not None:
                q = torch.from_numpy(q_host).to(dev, non_blocking=True)
            else:
                st.wait_event(ready)                   # q0 is complete on its producer's stream
                q = q0 if q0.device == dev else q0.to(dev, non_blocking=True)
            D, I = self.shards[si].search(q, k, min_score=floor)
            I = self._to_global(si, I)
            first = self._tdev[0]
            if dev != first:                           # peer-to-peer copy of the (B,k) result, still on this shard's stream
                D, I = D.to(first, non_blocking=True), I.to(first, non_blocking=True)
            done = torch.cuda.Event()
            done.record(st)
        return D, I, done

    def search(self, x, k: int, min_score: Optional[float] = None, id_offset: int = 0):
        """(D, I) like a single index: numpy in -> numpy out; a tensor on one of the devices -> tensors on the first
        device, ordered on its current stream."""
        k = int(k)
        if k < 1:
            raise RuntimeError("search: k must be >= 1")
        q = _as_rows(x, self.d, "search")
        on_device = isinstance(q, torch.Tensor) and q.is_cuda
        if isinstance(q, torch.Tensor) and not q.is_cuda:
            q = fc._as_f32_matrix(q.numpy(), self.d, "search")
        with self._lock:
            first = self._tdev[0]
            ready = None
            if on_device:
                q = q.contiguous().float()
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(q.device))
            futs = [self._pool.submit(self._shard_search, si, q if on_device else None, None if on_device else q, k, min_score, ready)
                    for si in range(self.nshards)]
            parts = [f.result() for f in futs]
            cur = torch.cuda.current_stream(first)
            for _, _, done in parts:
                cur.wait_event(done)
            with torch.cuda.device(first):
                if self.nshards == 1:
                    D, I = parts[0][0], parts[0][1]
                else:
                    D, I = fc.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
                if id_offset:
                    I = torch.where(I >= 0, I + int(id_offset), I)
            for p in parts:                             # the merge reads what other streams produced
                p[0].record_stream(cur); p[1].record_stream(cur)
            if on_device:
                return D, I
            return D.cpu().numpy(), I.cpu().numpy()


class MultiDeviceFlatIP(_MultiDeviceBase):
    """IndexFlatIP over several (logical) devices; class name free of "IVFFlat"/"HNSW"/"IVFPQ" on purpose
    (IndexFactory.detect_index_type sniffs it, core/index_factory.py:205-217)."""

    is_trained = True

    def __init__(self, d: int, devices: Sequence[int], storage: str = "f32", fp16_shadow: bool = True):
        if len(devices) < 1:
            raise ValueError("devices must name at least one GPU")
        self._init_shards([fc.IndexFlatIP(d, storage=storage, device=int(v), fp16_shadow=fp16_shadow)
                           for v in devices], devices)
        # segment table, in global-id order: global start, length, shard, local start
        self._seg_start: List[int] = []
        self._seg_len: List[int] = []
        self._seg_shard: List[int] = []
        self._seg_local: List[int] = []
        self._ntotal = 0
        self._maps = [None] * self.nshards   # per shard: (local starts, global - local) device tensors, rebuilt lazily

    @property
    def ntotal(self) -> int:
        return self._ntotal

    def train(self, x) -> None:
        pass

    def segments(self):
        """[(global start, length, shard, local start)] in global-id order."""
        return list(zip(self._seg_start, self._seg_len, self._seg_shard, self._seg_local))

    def add(self, x) -> None:
        """The whole batch goes to the least-full shard; ids continue from ntotal."""
        x = _as_rows(x, self.d, "add")
        n = int(x.shape[0])
        if n == 0:
            return
        with self._lock:
            if self._ntotal + n >= MAX_GLOBAL_IDS:
                raise RuntimeError("index would exceed 2^32-2 vectors")
            fills = [s.ntotal for s in self.shards]
            si = int(np.argmin(fills))
            if isinstance(x, torch.Tensor) and x.is_cuda and x.device != self._tdev[si]:
                torch.cuda.current_stream(x.device).synchronize()
                x = x.to(self._tdev[si])
            with torch.cuda.device(self._tdev[si]):
                self.shards[si].add(x)
            if self._seg_shard and self._seg_shard[-1] == si and self._seg_local[-1] + self._seg_len[-1] == fills[si]:
                self._seg_len[-1] += n          # the shard's rows continue the previous segment
            else:
                self._seg_start.append(self._ntotal); self._seg_len.append(n)
                self._seg_shard.append(si); self._seg_local.append(fills[si])
            self._ntotal += n
            self._maps[si] = None

    def add_random_unit(self, n: int, seed: int = 1234, chunk: int = 1 << 20) -> None:
        """Bench/test helper: device-generated unit rows, spread over the shards chunk by chunk."""
        for s0 in range(0, int(n), chunk):
            m = min(chunk, int(n) - s0)
            with self._lock:
                fills = [s.ntotal for s in self.shards]
                si = int(np.argmin(fills))
                self.shards[si].add_random_unit(m, seed + s0 // chunk)
                self._seg_start.append(self._ntotal); self._seg_len.append(m)
                self._seg_shard.append(si); self._seg_local.append(fills[si])
                self._ntotal += m
                self._maps[si] = None

    def _to_global(self, si: int, I: torch.Tensor) -> torch.Tensor:
        m = self._maps[si]
        if m is None:
            sel = [j for j, s in enumerate(self._seg_shard) if s == si]
            starts = torch.tensor([self._seg_local[j] for j in sel] or [0], dtype=torch.int64, device=self._tdev[si])
            delta = torch.tensor([self._seg_start[j] - self._seg_local[j] for j in sel] or [0], dtype=torch.int64, device=self._tdev[si])
            m = self._maps[si] = (starts, delta)
        starts, delta = m
        if starts.numel() == 1:
            return torch.where(I >= 0, I + delta[0], I)
        seg = torch.bucketize(I.clamp_min(0), starts, right=True) - 1
        return torch.where(I >= 0, I + delta[seg], I)

    # -- read back: global ids -> (shard, local row)
    def _locate(self, ids: np.ndarray):
        ids = np.asarray(ids, dtype=np.int64).reshape(-1)
        if len(ids) and (ids.min() < 0 or ids.max() >= self._ntotal):
            bad = ids[(ids < 0) | (ids >= self._ntotal)][0]
            raise RuntimeError(f"reconstruct: id {int(bad)} out of range [0, {self._ntotal})")
        starts = np.asarray(self._seg_start, dtype=np.int64)
        seg = np.searchsorted(starts, ids, side="right") - 1
        shard = np.asarray(self._seg_shard, dtype=np.int64)[seg] if len(ids) else np.zeros(0, np.int64)
        local = ids - starts[seg] + np.asarray(self._seg_local, dtype=np.int64)[seg] if len(ids) else ids
        return ids, shard, local

    def reconstruct_batch(self, ids) -> np.ndarray:
        ids, shard, local = self._locate(ids)
        out = np.empty((len(ids), self.d), dtype=np.float32)
        for si in range(self.nshards):
            sel = np.nonzero(shard == si)[0]
            if len(sel):
                out[sel] = self.shards[si].reconstruct_batch(local[sel])
        return out

    def reconstruct_batch_device(self, ids) -> torch.Tensor:
        ids, shard, local = self._locate(ids)
        first = self._tdev[0]
        out = torch.empty((len(ids), self.d), dtype=torch.float32, device=first)
        for si in range(self.nshards):
            sel = np.nonzero(shard == si)[0]
            if len(sel):
                with torch.cuda.device(self._tdev[si]):
                    part = self.shards[si].reconstruct_batch_device(local[sel])
                out[torch.from_numpy(sel).to(first)] = part.to(first)
        torch.cuda.synchronize(first)
        return out

    def reconstruct(self, i: int) -> np.ndarray:
        return self.reconstruct_batch([int(i)])[0]

    def reconstruct_n(self, i0: int = 0, ni: int = -1) -> np.ndarray:
        if ni < 0:
            ni = self._ntotal - i0
        return self.reconstruct_batch(np.arange(i0, i0 + ni, dtype=np.int64))

    def score_ids(self, ids, vectors) -> np.ndarray:
        ids, shard, local = self._locate(ids)
        v = fc._as_f32_matrix(vectors, self.d, "score_ids")
        out = np.empty((len(ids), v.shape[0]), dtype=np.float32)
        for si in range(self.nshards):
            sel = np.nonzero(shard == si)[0]
            if len(sel):
                out[sel] = self.shards[si].score_ids(local[sel], v)
        return out

    # -- persistence: ONE FAISS IxFI file, rows in global-id order (csrc/persist.hip states the layout)
    def write(self, path, chunk_rows: int = 16384) -> None:
        with self._lock, open(path, "wb") as f:
            f.write(b"IxFI" + _HEADER.pack(self.d, self._ntotal, 1 << 20, 1 << 20, 1, 0))
            f.write(struct.pack("<Q", self._ntotal * self.d))
            for s0 in range(0, self._ntotal, chunk_rows):
                f.write(self.reconstruct_n(s0, min(chunk_rows, self._ntotal - s0)).tobytes())


class MultiDeviceIVFFlat(_MultiDeviceBase):
    """IndexIVFFlat (inner product) with its inverted lists sharded over several (logical) devices: list l lives
    wholly on shard l % S under global ids."""

    def __init__(self, quantizer, d: int, nlist: int, metric: int, devices: Sequence[int], storage: str = "f32",
                 fp16_shadow: bool = True):
        if metric != fc.METRIC_INNER_PRODUCT:
            raise RuntimeError("IndexIVFFlat: only METRIC_INNER_PRODUCT is supported")
        if len(devices) < 1:
            raise ValueError("devices must name at least one GPU")
        self.quantizer = quantizer
        self._init_shards([fc.IndexIVFFlat(None, d, nlist, metric, storage=storage, device=int(v), fp16_shadow=fp16_shadow)
                          for v in devices], devices)
        self._ntotal = 0

    @property
    def ntotal(self) -> int:
        return self._ntotal

    @property
    def nlist(self) -> int:
        return self.shards[0].nlist

    @property
    def nprobe(self) -> int:
        return self.shards[0].nprobe

    @nprobe.setter
    def nprobe(self, v: int) -> None:
        for s in self.shards:
            s.nprobe = v

    @property
    def is_trained(self) -> bool:
        return self.shards[0].is_trained

    def train(self, x) -> None:
        """k-means on the first shard; every shard gets the same centroids."""
        with self._lock:
            if self.is_trained:
                return
            x = _as_rows(x, self.d, "train")
            if isinstance(x, torch.Tensor) and x.is_cuda and x.device != self._tdev[0]:
                torch.cuda.current_stream(x.device).synchronize()
                x = x.to(self._tdev[0])
            with torch.cuda.device(self._tdev[0]):
                self.shards[0].train(x)
            self.set_centroids(self.shards[0].get_centroids(), _skip_first=True)

    def set_centroids(self, centroids, _skip_first: bool = False) -> None:
        for s in self.shards[1 if _skip_first else 0:]:
            s.set_centroids(centroids)

    def get_centroids(self) -> np.ndarray:
        return self.shards[0].get_centroids()

    def add(self, x) -> None:
        x = _as_rows(x, self.d, "add")
        n = int(x.shape[0])
        if n == 0:
            return
        with self._lock:
            if not self.is_trained:
                raise RuntimeError("add: the index is not trained")
            if self._ntotal + n >= MAX_GLOBAL_IDS:
                raise RuntimeError("index would exceed 2^32-2 vectors")
            if isinstance(x, torch.Tensor) and x.is_cuda and x.device != self._tdev[0]:
                torch.cuda.current_stream(x.device).synchronize()
                x = x.to(self._tdev[0])
            with torch.cuda.device(self._tdev[0]):
                labels = self.shards[0].assign(x)             # one assignment per vector, on the first shard
            ids = np.arange(self._ntotal, self._ntotal + n, dtype=np.int64)
            owner = labels % self.nshards
            for si in range(self.nshards):
                sel = np.nonzero(owner == si)[0]
                if not len(sel):
                    continue
                if isinstance(x, torch.Tensor):
                    part = x[torch.from_numpy(sel).to(x.device)]
                    if part.is_cuda and part.device != self._tdev[si]:
                        torch.cuda.current_stream(part.device).synchronize()
                        part = part.to(self._tdev[si])
                else:
                    part = x[sel]
                with torch.cuda.device(self._tdev[si]):
                    self.shards[si].add_preassigned(part, labels[sel], ids[sel])
            self._ntotal += n

    def list_sizes(self) -> np.ndarray:
        return np.sum([s.list_sizes() for s in self.shards], axis=0)

    # -- persistence: ONE FAISS IwFl file (csrc/persist.hip states the layout), list l from the shard that owns it
    def write(self, path) -> None:
        with self._lock, open(path, "wb") as f:
            d, nlist = self.d, self.nlist
            trained = self.is_trained
            f.write(b"IwFl" + _HEADER.pack(d, self._ntotal, 1 << 20, 1 << 20, 1 if trained else 0, 0))
            f.write(struct.pack("<QQ", nlist, self.nprobe))
            ncent = nlist if trained else 0
            f.write(b"IxFI" + _HEADER.pack(d, ncent, 1 << 20, 1 << 20, 1, 0) + struct.pack("<Q", ncent * d))
            if trained:
                f.write(np.ascontiguousarray(self.get_centroids(), dtype=np.float32).tobytes())
            f.write(struct.pack("<BQ", 0, 0))                       # direct map: none
            sizes = self.list_sizes().astype(np.uint64)
            f.write(b"ilar" + struct.pack("<QQ", nlist, d * 4) + b"full" + struct.pack("<Q", nlist) + sizes.tobytes())
            for l in range(nlist):
                if sizes[l] == 0:
                    continue
                ids, vecs = self.shards[l % self.nshards].list_members(l)
                f.write(vecs.tobytes())
                f.write(ids.tobytes())

    def make_direct_map(self, _enable: bool = True) -> None:
        raise RuntimeError("make_direct_map: a list-sharded index stores its vectors under global ids; reconstruct is not "
                           "available (like FAISS without a direct map)")

    def reconstruct(self, i: int) -> np.ndarray:
        raise RuntimeError("reconstruct: this list-sharded IVF index has no direct map")

    def reconstruct_batch(self, ids) -> np.ndarray:
        raise RuntimeError("reconstruct: this list-sharded IVF index has no direct map")

    def reconstruct_batch_device(self, ids) -> torch.Tensor:
        raise RuntimeError("reconstruct: this list-sharded IVF index has no direct map")

    def score_ids(self, ids, vectors) -> np.ndarray:
        raise RuntimeError("reconstruct: this list-sharded IVF index has no direct map")


# ---------------------------------------------------------------------------------------------------- read
def _need(f, n: int, what: str) -> bytes:
    b = f.read(n)
    if len(b) != n:
        raise RuntimeError(f"read_index: truncated file ({what})")
    return b


def _read_header(f):
    d, ntotal, _, _, trained, metric = _HEADER.unpack(_need(f, _HEADER.size, "index header"))
    if metric != 0:
        raise RuntimeError(f"read_index: metric type {metric} is not supported (only METRIC_INNER_PRODUCT)")
    if d < 1 or ntotal < 0:
        raise RuntimeError(f"read_index: corrupt header (d={d} ntotal={ntotal})")
    return d, ntotal, bool(trained)


def _read_flat_payload(f, d: int, ntotal: int, left: int, sink, chunk_rows: int = 16384) -> None:
    (nfl,) = struct.unpack("<Q", _need(f, 8, "payload size"))
    if nfl != ntotal * d:
        raise RuntimeError(f"read_index: payload size {nfl} != ntotal*d")
    if nfl * 4 > left:
        raise RuntimeError("read_index: truncated vector payload")
    for s0 in range(0, ntotal, chunk_rows):
        m = min(chunk_rows, ntotal - s0)
        sink(np.frombuffer(_need(f, m * d * 4, "vectors"), dtype=np.float32).reshape(m, d))


def read_multi(path, devices: Sequence[int], storage: str = "f32"):
    """faiss.read_index for an index spread over several GPUs: the vectors of a FAISS IxFI / IwFl file
    (core/index_manager.py:185) go to the shards as they are read; a Flat file's rows are dealt out in chunks (the
    segment table keeps their ids), an IVF file's lists go to the shards that own them."""
    import os
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        cc = _need(f, 4, "fourcc")
        if cc == b"IxFI":
            d, ntotal, _ = _read_header(f)
            index = MultiDeviceFlatIP(d, devices, storage=storage)
            try:
                _read_flat_payload(f, d, ntotal, size - f.tell(), index.add)
            except Exception:
                index.close()
                raise
            return index
        if cc != b"IwFl":
            raise RuntimeError(f"read_index: unsupported index type {cc!r} (this engine reads IxFI and IwFl)")
        d, ntotal, trained = _read_header(f)
        nlist, nprobe = struct.unpack("<QQ", _need(f, 16, "IVF header"))
        if nlist < 1 or nlist > (1 << 24) or nprobe > 0x7FFFFFFF:
            raise RuntimeError("read_index: corrupt IVF header")
        index = MultiDeviceIVFFlat(None, d, int(nlist), fc.METRIC_INNER_PRODUCT, devices, storage=storage)
        try:
            if _need(f, 4, "quantizer") != b"IxFI":
                raise RuntimeError("read_index: unsupported quantizer (expected IxFI)")
            qd, ncent, _ = _read_header(f)
            if qd != d or ncent not in (0, nlist):
                raise RuntimeError(f"read_index: quantizer holds {ncent} centroids of dimension {qd}, nlist is {nlist}")
            cents = []
            _read_flat_payload(f, d, ncent, size - f.tell(), cents.append)
            if ncent and trained:
                index.set_centroids(np.concatenate(cents))
            index.nprobe = max(1, int(nprobe))
            dm_type, dm_n = struct.unpack("<BQ", _need(f, 9, "direct map"))
            if dm_type > 1 or dm_n > (1 << 40):
                raise RuntimeError("read_index: unsupported or corrupt direct map")
            f.seek(dm_n * 8, 1)
            if _need(f, 4, "inverted lists") != b"ilar":
                raise RuntimeError("read_index: unsupported inverted-list type")
            il_nlist, code_size = struct.unpack("<QQ", _need(f, 16, "inverted lists"))
            if il_nlist != nlist or code_size != d * 4:
                raise RuntimeError("read_index: inverted lists do not match the index header")
            enc = _need(f, 4, "list sizes")
            sizes = np.zeros((nlist,), dtype=np.uint64)
            if enc == b"full":
                (nsz,) = struct.unpack("<Q", _need(f, 8, "list sizes"))
                if nsz != nlist:
                    raise RuntimeError("read_index: corrupt list sizes")
                sizes = np.frombuffer(_need(f, nlist * 8, "list sizes"), dtype=np.uint64).copy()
            elif enc == b"sprs":
                (nsz,) = struct.unpack("<Q", _need(f, 8, "list sizes"))
                if nsz % 2 or nsz > 2 * nlist:
                    raise RuntimeError("read_index: corrupt sparse list sizes")
                pairs = np.frombuffer(_need(f, nsz * 8, "list sizes"), dtype=np.uint64).reshape(-1, 2)
                if len(pairs) and pairs[:, 0].max() >= nlist:
                    raise RuntimeError("read_index: corrupt sparse list sizes")
                sizes[pairs[:, 0].astype(np.int64)] = pairs[:, 1]
            else:
                raise RuntimeError(f"read_index: unknown list-size encoding {enc!r}")
            left = size - f.tell()
            if int(sizes.sum()) != ntotal or int(sizes.sum()) * (d * 4 + 8) > left:
                raise RuntimeError("read_index: inverted lists do not match the header or the file length")
            for l in range(int(nlist)):
                n = int(sizes[l])
                if not n:
                    continue
                vecs = np.frombuffer(_need(f, n * d * 4, "list codes"), dtype=np.float32).reshape(n, d)
                ids = np.frombuffer(_need(f, n * 8, "list ids"), dtype=np.int64)
                if len(ids) and (ids.min() < 0 or ids.max() >= MAX_GLOBAL_IDS):
                    raise RuntimeError("read_index: id out of the supported range")
                si = l % index.nshards
                with torch.cuda.device(index._tdev[si]):
                    index.shards[si].add_preassigned(vecs, np.full((n,), l, dtype=np.int32), ids)
            index._ntotal = int(ntotal)
        except Exception:
            index.close()
            raise
        return index

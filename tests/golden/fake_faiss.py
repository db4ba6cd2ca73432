"""A numpy stand-in for the `faiss` module, backed by oracle/: ONLY for tests/golden/make_factory_golden.py, which
needs something importable under the name `faiss` to load the reference's core/index_factory.py in the build
container (faiss itself is not installed anywhere in reach).  It offers what that file touches
(core/index_factory.py:140-158, 169-170, 191-192, 208-236, 274, 307, 317-320): IndexFlatIP, IndexIVFFlat,
IndexHNSWFlat / IndexIVFPQ as names for isinstance, METRIC_INNER_PRODUCT, and on the objects d / ntotal /
is_trained / nlist / nprobe / train / add / search / reconstruct / make_direct_map.  FAISS behaviours restated
[FAISS-lib]: add() on an untrained IVF index and train() with fewer points than clusters raise RuntimeError (what the
SWIG wrapper turns FAISS's exceptions into), reconstruct() on an IVF index without a direct map raises RuntimeError.
Never shipped to the GPU box as part of any test that runs there; the product never sees it."""
import numpy as np

from oracle import flat_ip, ivf_flat

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1


class Index:
    pass


class IndexFlatIP(Index):
    def __init__(self, d):
        self.d = int(d)
        self._x = np.zeros((0, self.d), np.float32)
        self.is_trained = True

    @property
    def ntotal(self):
        return self._x.shape[0]

    def train(self, x):
        pass

    def add(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.ndim == 2 and x.shape[1] == self.d
        self._x = np.concatenate([self._x, x])

    def search(self, q, k):
        return flat_ip.search(self._x, np.ascontiguousarray(q, dtype=np.float32), k)

    def reconstruct(self, i):
        if not 0 <= int(i) < self.ntotal:
            raise RuntimeError("Error in reconstruct: index out of range")
        return self._x[int(i)].copy()


class IndexIVFFlat(Index):
    def __init__(self, quantizer, d, nlist, metric=METRIC_L2):
        assert metric == METRIC_INNER_PRODUCT
        self.quantizer, self.d, self.nlist, self.nprobe = quantizer, int(d), int(nlist), 1
        self.is_trained = False
        self._x = np.zeros((0, self.d), np.float32)
        self._labels = np.zeros((0,), np.int64)
        self._c = None
        self._direct_map = False

    @property
    def ntotal(self):
        return self._x.shape[0]

    def train(self, x):
        if self.is_trained:
            return
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.shape[0] < self.nlist:
            raise RuntimeError(f"Error: 'nx >= k' failed: Number of training points ({x.shape[0]}) should be at least as large as number of clusters ({self.nlist})")
        self._c, _ = ivf_flat.kmeans(x, self.nlist)
        self.is_trained = True

    def add(self, x):
        if not self.is_trained:
            raise RuntimeError("Error: 'is_trained' failed")
        x = np.ascontiguousarray(x, dtype=np.float32)
        self._x = np.concatenate([self._x, x])
        self._labels = np.concatenate([self._labels, ivf_flat.assign(x, self._c)])

    def search(self, q, k):
        q = np.ascontiguousarray(q, dtype=np.float32)
        if not self.is_trained or self.ntotal == 0:
            return (np.full((q.shape[0], k), flat_ip.NEG_MAX, np.float32), np.full((q.shape[0], k), -1, np.int64))
        D, I, _ = ivf_flat.search(self._x, self._c, self._labels, q, k, self.nprobe)
        return D, I

    def make_direct_map(self, enable=True):
        self._direct_map = bool(enable)

    def reconstruct(self, i):
        if not self._direct_map:
            raise RuntimeError("Error: direct map not initialized")
        if not 0 <= int(i) < self.ntotal:
            raise RuntimeError("Error in reconstruct: index out of range")
        return self._x[int(i)].copy()


class IndexHNSWFlat(Index):
    def __init__(self, *a, **k):
        raise RuntimeError("fake_faiss: HNSW is outside the replaced path")


class IndexIVFPQ(Index):
    def __init__(self, *a, **k):
        raise RuntimeError("fake_faiss: IVF-PQ is outside the replaced path")

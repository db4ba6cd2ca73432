"""Oracle: IVF-Flat with inner-product metric (FAISS IndexIVFFlat semantics).  TEST INFRASTRUCTURE.

PARITY UNPINNED (see oracle/__init__.py).  Follows:
  - /root/reference/core/index_factory.py:142-146  (IndexIVFFlat(IndexFlatIP(d), d, nlist,
    METRIC_INNER_PRODUCT); index.nprobe = config.nprobe)
  - /root/reference/core/index_factory.py:173-195  (train), :197-203 (min training size)
  - /root/reference/core/index_manager.py:284-319  (buffer -> train -> add), :371 (search)
  - FAISS documented behaviour [FAISS-lib]: SPHERICAL k-means (IndexIVF's constructor sets
    ClusteringParameters::spherical when the metric is inner product: every centroid is
    L2-normalised after the update + empty-cluster split of an iteration, and once after the
    initialisation; `spherical=False` keeps plain means, what rounds 1-2 restated) with
    max-inner-product assignment (the quantizer is an IndexFlatIP), at most 256
    training points per centroid (random subsample), 10 iterations, random distinct
    training points as initial centroids; add() appends each vector to the list of
    its max-inner-product centroid; search() visits the nprobe lists with the largest
    query.centroid and ranks their members by exact inner product.
k-means is not bit-reproducible across implementations (random init, float order), so
tests compare clustering QUALITY and, for search, feed this oracle the centroids and
the list assignment actually stored in the index under test.
"""
import numpy as np

from . import flat_ip


def assign(X: np.ndarray, C: np.ndarray, block: int = 16384) -> np.ndarray:
    """argmax_c <x, c> (lowest centroid index on exact ties)."""
    X = np.asarray(X, np.float32); C = np.asarray(C, np.float32)
    out = np.empty((X.shape[0],), np.int64)
    for s in range(0, X.shape[0], block):
        out[s:s + block] = np.argmax(X[s:s + block] @ C.T, axis=1)
    return out


def assignment_is_valid(X, C, labels, rtol=1e-5, atol=2e-6):
    """Every vector sits in a list whose centroid score ties the best one."""
    sc = np.asarray(X, np.float64) @ np.asarray(C, np.float64).T
    best = sc.max(axis=1)
    got = sc[np.arange(len(labels)), labels]
    return bool((best - got <= rtol * np.abs(best) + atol).all())


def search(X, C, labels, Q, k, nprobe, C_coarse=None, Q_coarse=None):
    """X: stored rows (n,d); labels: list of each row; returns (D, I, probed lists)."""
    X = np.asarray(X, np.float32); Q = np.atleast_2d(np.asarray(Q, np.float32))
    nlist = C.shape[0]
    nprobe = min(nprobe, nlist)
    _, probes = flat_ip.search(C if C_coarse is None else C_coarse, Q if Q_coarse is None else Q_coarse, nprobe)
    labels = np.asarray(labels)
    members = [np.nonzero(labels == l)[0] for l in range(nlist)]
    D = np.full((Q.shape[0], k), flat_ip.NEG_MAX, np.float32)
    I = np.full((Q.shape[0], k), -1, np.int64)
    for qi in range(Q.shape[0]):
        cand = np.concatenate([members[l] for l in probes[qi] if l >= 0]) if nprobe else np.zeros(0, np.int64)
        if len(cand) == 0:
            continue
        cand = np.sort(cand)
        d, i = flat_ip.topk_total_order((Q[qi:qi + 1] @ X[cand].T), k, cand.astype(np.int64))
        D[qi], I[qi] = d[0], i[0]
    return D, I, probes


def _renorm(C):
    """[FAISS-lib] Clustering::post_process_centroids for spherical k-means: c / ||c||, zero rows left alone."""
    nrm = np.sqrt((C.astype(np.float64) ** 2).sum(axis=1))
    nz = nrm > 0
    C[nz] = (C[nz] / nrm[nz, None]).astype(np.float32)
    return C


def kmeans(X, nlist, niter=10, seed=1234, max_points_per_centroid=256, spherical=True):
    """Restated FAISS-style k-means for the IP quantizer.  Returns (centroids, objective)
    where objective = mean over training points of the best inner product."""
    X = np.asarray(X, np.float32)
    rng = np.random.default_rng(seed)
    n = X.shape[0]
    if n > max_points_per_centroid * nlist:
        X = X[np.sort(rng.choice(n, max_points_per_centroid * nlist, replace=False))]
        n = X.shape[0]
    C = X[rng.choice(n, nlist, replace=False)].copy()
    if spherical:
        _renorm(C)
    for _ in range(niter):
        lab = assign(X, C)
        counts = np.bincount(lab, minlength=nlist)
        sums = np.zeros_like(C, dtype=np.float64)
        np.add.at(sums, lab, X.astype(np.float64))
        nz = counts > 0
        C[nz] = (sums[nz] / counts[nz, None]).astype(np.float32)
        for l in np.nonzero(~nz)[0]:  # empty cluster: split the largest one
            big = int(np.argmax(counts))
            if counts[big] < 2:
                break
            eps = 1.0 / 1024.0
            sign = np.where(np.arange(C.shape[1]) & 1, 1.0, -1.0).astype(np.float32)
            C[l] = C[big] * (1 + sign * eps)
            C[big] = C[big] * (1 - sign * eps)
            counts[l] = counts[big] // 2
            counts[big] -= counts[l]
        if spherical:
            _renorm(C)
    return C, objective(X, C)


def objective(X, C):
    best = np.full((X.shape[0],), -np.inf, np.float32)
    for s in range(0, X.shape[0], 16384):
        best[s:s + 16384] = (X[s:s + 16384] @ C.T).max(axis=1)
    return float(best.mean())

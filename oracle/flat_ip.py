"""Oracle: exact inner-product top-k (FAISS IndexFlatIP semantics).  TEST INFRASTRUCTURE.

PARITY UNPINNED (see oracle/__init__.py): restates documented FAISS behaviour at
the reference call sites, not reference output.

Follows:
  - /root/reference/core/index_factory.py:138-140  (FLAT -> faiss.IndexFlatIP(dim))
  - /root/reference/core/index_manager.py:363-378  (query reshaped to (1,d) fp32
    C-contiguous; scores, indices = index.search(q, top_k); ids < 0 dropped)
  - FAISS IndexFlatIP.search [FAISS-lib, documented]: for every query the k
    largest inner products, sorted by decreasing score, int64 labels; missing
    results are label -1 with score -FLT_MAX (-3.4028235e38).  FAISS leaves the
    order among exactly equal scores unspecified; this oracle (and the HIP
    kernels) fix it as ascending id.
"""
import numpy as np

NEG_MAX = np.float32(-3.4028234663852886e38)


def topk_total_order(scores: np.ndarray, k: int, ids=None):
    """Top-k of each row of `scores` ordered by (score desc, id asc), padded."""
    scores = np.asarray(scores)
    nq, n = scores.shape
    D = np.full((nq, k), NEG_MAX, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    if n == 0:
        return D, I
    if ids is None:
        ids = np.arange(n, dtype=np.int64)
    kk = min(k, n)
    for qi in range(nq):
        s = scores[qi]
        if kk < n:
            # candidates: everything >= the kk-th largest value (keeps all ties at the cut)
            part = np.partition(s, n - kk)[n - kk]
            cand = np.nonzero(s >= part)[0]
        else:
            cand = np.arange(n)
        order = np.lexsort((ids[cand], -s[cand].astype(np.float64)))[:kk]
        sel = cand[order]
        D[qi, :kk] = s[sel].astype(np.float32)
        I[qi, :kk] = ids[sel]
    return D, I


def search(X: np.ndarray, Q: np.ndarray, k: int, block: int = 65536):
    """fp32 oracle: scores = Q @ X.T accumulated in fp32 (BLAS), exact top-k.

    X: (n,d) float32 (the values actually stored; for fp16 storage pass the
    fp16-rounded rows converted back to float32).  Q: (nq,d) or (d,) float32.
    """
    X = np.ascontiguousarray(X, dtype=np.float32)
    Q = np.asarray(Q, dtype=np.float32)
    if Q.ndim == 1:
        Q = Q.reshape(1, -1)  # core/index_manager.py:364-365
    Q = np.ascontiguousarray(Q)
    n = X.shape[0]
    nq = Q.shape[0]
    if n == 0:
        return topk_total_order(np.zeros((nq, 0), np.float32), k)
    # blocked so that 1M-row cases do not need an (nq, n) matrix for large nq
    bestD = np.full((nq, 0), NEG_MAX, np.float32)
    bestI = np.full((nq, 0), -1, np.int64)
    for s in range(0, n, block):
        e = min(n, s + block)
        sc = Q @ X[s:e].T
        D, I = topk_total_order(sc, min(k, e - s), np.arange(s, e, dtype=np.int64))
        allD = np.concatenate([bestD, D], axis=1)
        allI = np.concatenate([bestI, I], axis=1)
        newD = np.full((nq, k), NEG_MAX, np.float32)
        newI = np.full((nq, k), -1, np.int64)
        for qi in range(nq):
            valid = allI[qi] >= 0
            order = np.lexsort((allI[qi][valid], -allD[qi][valid].astype(np.float64)))[:k]
            m = len(order)
            newD[qi, :m] = allD[qi][valid][order]
            newI[qi, :m] = allI[qi][valid][order]
        bestD, bestI = newD, newI
    if bestD.shape[1] < k:
        padD = np.full((nq, k), NEG_MAX, np.float32)
        padI = np.full((nq, k), -1, np.int64)
        padD[:, : bestD.shape[1]] = bestD
        padI[:, : bestI.shape[1]] = bestI
        bestD, bestI = padD, padI
    return bestD, bestI


def scores_f64(X: np.ndarray, Q: np.ndarray, ids: np.ndarray) -> np.ndarray:
    """float64 inner products of each query with the rows named per query: (nq,k)."""
    Q = np.asarray(Q, dtype=np.float64)
    if Q.ndim == 1:
        Q = Q.reshape(1, -1)
    out = np.full(ids.shape, -np.inf)
    for qi in range(ids.shape[0]):
        v = ids[qi] >= 0
        out[qi, v] = X[ids[qi][v]].astype(np.float64) @ Q[qi]
    return out


def check_against(D, I, D_ref, I_ref, X, Q, rtol=1e-3, tie_rtol=1e-3, atol=2e-6):
    """The north-star parity rule: ids equal the reference except where the
    scores involved tie within `tie_rtol` (relative), scores within `rtol`.
    `atol` is an absolute floor for scores near zero (inputs are unit-norm, so fp32
    accumulation error is ~1e-7 absolute whatever the score).
    Returns (ok, message, n_id_mismatch_positions)."""
    D = np.asarray(D); I = np.asarray(I); D_ref = np.asarray(D_ref); I_ref = np.asarray(I_ref)
    if D.shape != D_ref.shape or I.shape != I_ref.shape:
        return False, f"shape mismatch {D.shape} vs {D_ref.shape}", -1
    if ((I < 0) != (I_ref < 0)).any():
        return False, "padding (-1) positions differ", -1
    valid = I_ref >= 0
    scale = np.maximum(np.abs(D_ref), 1e-30)
    with np.errstate(over="ignore", invalid="ignore"):
        excess = np.abs(D.astype(np.float64) - D_ref.astype(np.float64)) - (rtol * scale + atol)
    if valid.any() and excess[valid].max() > 0:
        return False, f"score mismatch: |D-D_ref| exceeds {rtol}*|D_ref|+{atol} by {excess[valid].max():.3e}", -1
    if (~valid).any() and not (D[~valid] == NEG_MAX).all():
        return False, "padding scores are not -FLT_MAX", -1
    mism = (I != I_ref) & valid
    nm = int(mism.sum())
    if nm:
        # every mismatching id must be a near-tie: its true score is within tie_rtol of the
        # reference score at that rank
        true = scores_f64(X, Q, np.where(mism, I, -1))
        bad = mism & (np.abs(true - D_ref.astype(np.float64)) > tie_rtol * scale + atol)
        if bad.any():
            qi, pos = np.argwhere(bad)[0]
            return False, (f"id mismatch beyond tie tolerance at query {qi} rank {pos}: got {I[qi,pos]} "
                           f"(score {true[qi,pos]:.7g}) want {I_ref[qi,pos]} (score {D_ref[qi,pos]:.7g})"), nm
    # no duplicates among valid ids
    for qi in range(I.shape[0]):
        v = I[qi][I[qi] >= 0]
        if len(np.unique(v)) != len(v):
            return False, f"duplicate ids in query {qi}", nm
    return True, "ok", nm

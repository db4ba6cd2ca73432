"""Seeded sweep over boundary-rich shapes: every kernel family of the Flat and IVF searches against the oracle.

The hand-picked cases of test_flat_gpu / test_mfma_gpu / test_ivf_gpu pin the BASELINE shapes and the reference's edge
cases; this file walks the seams between the kernels instead (batch sizes around 4/5, 8/9, 128/129, 256/257; k around
64/65, 256/257 and above n; row counts around the 256-row tile and the storage segments; dimensions that need padding
or miss a kernel's stride), with the rows added in ragged pieces.  Same comparison and tolerances as those files:
exact fp32 paths 1e-5, fp16 operands against the oracle run on the rounded operands 1e-5, the fp16 shadow of fp32
rows against the fp32 oracle at the north star's 1e-3.  Seeds are fixed: a failure names its case.
"""
import numpy as np
import pytest

from conftest import SALT, unit_rows
from oracle import flat_ip, ivf_flat

pytestmark = pytest.mark.gpu

DIMS = [8, 24, 64, 100, 192, 256, 512, 1000, 1024, 2048, 4096]
ROWS = [1, 2, 63, 64, 65, 255, 256, 257, 511, 513, 1000, 3071, 4097, 9000]
BATCHES = [1, 2, 3, 4, 5, 8, 9, 16, 17, 64, 127, 128, 129, 256, 257, 300]
KS = [1, 2, 10, 63, 64, 65, 100, 128, 129, 256, 257, 512, 1024]


def r16(a):
    return a.astype(np.float16).astype(np.float32)


def rbf16(a):
    """fp32 -> bf16 (round to nearest even) -> fp32: bf16-valued rows and queries, what the reference's embedder emits
    (core/model_wrapper.py:201-203).  On them bf16 storage is lossless and every path returns the fp32 oracle's result."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16
    return r.view(np.float32).reshape(np.shape(a))


def add_in_pieces(index, X, rng):
    """Rows arrive as the reference delivers them: one bulk add, then ragged small ones (16 at a time and odd tails)."""
    n = len(X)
    cut = int(rng.integers(0, n + 1))
    if cut:
        index.add(X[:cut])
    while cut < n:
        step = int(rng.choice([1, 3, 16, 16, 250]))
        index.add(X[cut:cut + step])
        cut += step
    assert index.ntotal == n


@pytest.mark.parametrize("seed", range(120))
def test_flat_sweep(ssa, monkeypatch, seed):
    rng = np.random.default_rng(1000 + seed + SALT)
    d = int(rng.choice(DIMS))
    n = int(rng.choice(ROWS))
    nq = int(rng.choice(BATCHES))
    k = int(rng.choice(KS))
    storage = str(rng.choice(["f32", "f16", "bf16"]))
    if rng.random() < 0.4:  # small storage segments: rows of one search spread over several of them
        monkeypatch.setenv("SSMI_SEG_SHIFT", str(int(rng.choice([8, 9, 10]))))
    case = f"seed={seed} n={n} d={d} nq={nq} k={k} storage={storage}"
    X, Q = unit_rows(n, d, 5000 + seed), unit_rows(nq, d, 7000 + seed)
    if storage == "bf16":  # bf16-valued inputs: no rounding anywhere, whatever kernel the shape takes
        X, Q = rbf16(X), rbf16(Q)
    if rng.random() < 0.3 and n >= 8:  # duplicate rows: ties must come back in ascending id
        X[n // 2:n // 2 + 3] = X[0]
    index = ssa.IndexFlatIP(d, storage=storage)
    add_in_pieces(index, X, rng)
    D, I = index.search(Q, k)
    path = index.last_timing()["last_path"]
    Xs = r16(X) if storage == "f16" else X
    if path == 1 and storage == "f32":
        # fp16 shadow of the fp32 rows, fp16 queries: north-star tolerance against the fp32 result.  The 1e-3 bound is
        # relative; rounding both unit operands to fp16 moves a score by about 2^-11 * sqrt(2/d) ABSOLUTE (std), which a
        # relative bound cannot absorb for the near-zero scores a k >= n search returns: 6 sigma of that as the floor
        Dr, Ir = flat_ip.search(X, Q, k)
        atol = 6 * 2.0 ** -11 * np.sqrt(2.0 / d)
        ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, X, Q, rtol=1e-3, tie_rtol=1e-3, atol=atol)
    elif path == 1:
        Qs = r16(Q) if storage == "f16" else Q
        Dr, Ir = flat_ip.search(Xs, Qs, k)
        ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xs, Qs, rtol=1e-5, tie_rtol=1e-5)
    else:
        Dr, Ir = flat_ip.search(Xs, Q, k)
        ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xs, Q, rtol=1e-5, tie_rtol=1e-5)
    assert ok, f"{case} path={path} kernel={index.last_kernel()}: {msg}"
    # rows of -1 padding exactly where the index runs out of vectors, sorted scores
    assert ((I >= 0).sum(1) == min(n, k)).all(), case
    assert (np.diff(D[:, :min(n, k)], axis=1) <= 0).all(), case
    index.close()


@pytest.mark.parametrize("seed", range(60))
def test_ivf_sweep(ssa, monkeypatch, seed):
    rng = np.random.default_rng(2000 + seed + SALT)
    d = int(rng.choice([24, 64, 100, 256, 512, 1024, 2048, 4096]))
    nlist = int(rng.choice([1, 3, 16, 40, 64, 200]))
    n = int(rng.choice([nlist, 257, 1000, 3071, 6000]))
    n = max(n, nlist)
    nq = int(rng.choice([1, 2, 5, 8, 9, 17, 64, 130, 300]))
    k = int(rng.choice([1, 10, 64, 65, 100, 256, 300]))
    nprobe = int(rng.choice([1, 2, 5, nlist, nlist + 3]))
    storage = str(rng.choice(["f32", "f16", "bf16"]))
    if rng.random() < 0.4:
        monkeypatch.setenv("SSMI_SEG_SHIFT", str(int(rng.choice([8, 10]))))
    if storage == "f32":
        monkeypatch.setenv("SSMI_NO_SHADOW", "1")  # fp32 lists stay on the exact scans: one tolerance for the whole sweep
    case = f"seed={seed} n={n} d={d} nlist={nlist} nprobe={nprobe} nq={nq} k={k} storage={storage}"
    cen = unit_rows(max(nlist, 8), d, 9000 + seed)
    C = cen[:nlist]
    X = cen[rng.integers(0, len(cen), n)] + 0.6 * unit_rows(n, d, 9100 + seed)
    X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
    Q = unit_rows(nq, d, 9200 + seed)
    if storage == "bf16":  # bf16-valued vectors, centroids and queries: lossless in bf16 lists
        X, C, Q = rbf16(X), rbf16(C), rbf16(Q)
    index = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage=storage), d, nlist, ssa.METRIC_INNER_PRODUCT, storage=storage)
    index.set_centroids(C)
    add_in_pieces(index, X, rng)
    labels = index.assignments()
    # pieces of 9 vectors and more are assigned on fp16 centroids and fp16-rounded vectors: the absolute floor of the Flat
    # sweep above (a score near zero has no relative tolerance to speak of; found with SSMI_FUZZ_SALT=7, seed 21: two
    # centroids 1.3e-5 apart at a score of -0.0016)
    assert ivf_flat.assignment_is_valid(X, C, labels, rtol=2e-3, atol=6 * 2.0 ** -11 * np.sqrt(2.0 / d)), case
    np.testing.assert_array_equal(index.list_sizes(), np.bincount(labels, minlength=nlist))
    index.nprobe = nprobe
    D, I = index.search(Q, k)
    Xs = r16(X) if storage == "f16" else X
    npr = min(nprobe, nlist)
    kernel = index.last_kernel()
    mfma = "mfma" in kernel  # list-major matrix-core scan: queries are rounded to fp16 as well (bf16 lists: to bf16, which they are)
    Qs = r16(Q) if (mfma and storage != "bf16") else Q
    # a batch of >= 9 queries ranks the lists with fp16 centroids and queries (matrix-core coarse quantizer): the
    # oracle probes with the same rounded operands
    coarse = dict(C_coarse=r16(C), Q_coarse=r16(Q)) if (nq >= 9 and d % 64 == 0) else {}
    Dr, Ir, _ = ivf_flat.search(Xs, C, labels, Qs, k, npr, **coarse)
    ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xs, Qs, rtol=1e-5, tie_rtol=1e-5)
    if not ok and nq >= 9 and npr < nlist:
        # fp16-ranked probe lists: accept when every returned hit is a true member of a list whose centroid score is
        # within the fp16 rounding of the nprobe-th best, and the scores of the hits are exact
        sc = Q @ C.T
        kth = np.sort(sc, axis=1)[:, -npr][:, None]
        near = sc >= kth - 2e-3
        for qi in range(nq):
            ids = I[qi][I[qi] >= 0]
            assert near[qi][labels[ids]].all(), f"{case} kernel={kernel}: hit from a list that was not probable: {msg}"
            np.testing.assert_allclose(D[qi][:len(ids)], Xs[ids] @ Qs[qi], rtol=1e-4, atol=1e-6, err_msg=case)
        ok = True
    assert ok, f"{case} kernel={kernel}: {msg}"
    index.close()


@pytest.mark.parametrize("n,d,nq,k,storage", [
    (300, 64, 5000, 10, "f16"),       # more queries than one MFMA launch takes over a small store (65536 per launch)
    (70000, 64, 4500, 5, "f16"),      # ... and over a large one (4096 per launch): two launches, two merges
    (70000, 64, 4500, 5, "f32"),      # the same through the fp16 shadow
    (2000, 256, 3000, 300, "f32"),    # exact scan with a wide k: the partial-list buffer bounds the queries per pass
])
def test_query_batches_beyond_one_launch(ssa, monkeypatch, n, d, nq, k, storage):
    if storage == "f32" and k == 300:
        monkeypatch.setenv("SSMI_NO_SHADOW", "1")
    X, Q = unit_rows(n, d, 31), unit_rows(nq, d, 32)
    index = ssa.IndexFlatIP(d, storage=storage)
    index.add(X)
    D, I = index.search(Q, k)
    path = index.last_timing()["last_path"]
    Xs = r16(X) if storage == "f16" else X
    if path == 1 and storage == "f32":
        Dr, Ir = flat_ip.search(X, Q, k)
        ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, X, Q, rtol=1e-3, tie_rtol=1e-3, atol=6 * 2.0 ** -11 * np.sqrt(2.0 / d))
    else:
        Qs = r16(Q) if path == 1 else Q
        Dr, Ir = flat_ip.search(Xs, Qs, k)
        ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xs, Qs, rtol=1e-5, tie_rtol=1e-5)
    assert ok, f"path={path} kernel={index.last_kernel()}: {msg}"
    index.close()


@pytest.mark.parametrize("seed", range(16))
def test_ivf_grow_and_search_walk(ssa, monkeypatch, seed):
    """One IVF index that keeps growing (the reference adds 16 vectors at a time) while it is searched with changing
    batch sizes, k and nprobe: per-query scans, list-major scans and (fp16 rows, or the fp16 shadow of fp32 lists) the
    matrix-core scan all read the same page tables, which every add extends; after each step the result must match the
    oracle run on the index's own assignment of everything added so far."""
    rng = np.random.default_rng(2500 + seed + SALT)
    d = int(rng.choice([64, 512, 1024, 4096]))
    nlist = int(rng.choice([4, 16, 50]))
    storage = str(rng.choice(["f32", "f16", "bf16"]))
    cen = unit_rows(nlist, d, 2600 + seed)
    pool = cen[rng.integers(0, nlist, 5000)] + 0.6 * unit_rows(5000, d, 2700 + seed)
    pool = (pool / np.linalg.norm(pool, axis=1, keepdims=True)).astype(np.float32)
    if storage == "bf16":  # bf16-valued vectors, centroids and (below) queries
        cen, pool = rbf16(cen), rbf16(pool)
    index = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage=storage), d, nlist, ssa.METRIC_INNER_PRODUCT, storage=storage)
    index.set_centroids(cen)
    at = 0
    for step in range(14):
        grow = int(rng.choice([1, 16, 16, 40, 400, 1500]))
        grow = min(grow, len(pool) - at)
        if grow:
            index.add(pool[at:at + grow]); at += grow
        nq = int(rng.choice([1, 3, 9, 20, 70, 200]))
        k = int(rng.choice([1, 10, 100, 256]))
        nprobe = int(rng.choice([1, 3, nlist]))
        index.nprobe = nprobe
        Q = unit_rows(nq, d, 2800 + seed * 31 + step)
        if storage == "bf16":
            Q = rbf16(Q)
        D, I = index.search(Q, k)
        case = f"seed={seed} step={step} d={d} nlist={nlist} {storage} n={at} nq={nq} k={k} nprobe={nprobe} kernel={index.last_kernel()}"
        labels = index.assignments()
        assert len(labels) == at and index.ntotal == at, case
        X = pool[:at]
        mfma = "mfma" in index.last_kernel()
        Xs = r16(X) if (storage == "f16" or (mfma and storage == "f32")) else X
        Qs = r16(Q) if (mfma and storage != "bf16") else Q
        coarse = dict(C_coarse=r16(cen), Q_coarse=r16(Q)) if (nq >= 9 and d % 64 == 0) else {}
        Dr, Ir, _ = ivf_flat.search(Xs, cen, labels, Qs, k, min(nprobe, nlist), **coarse)
        tol = 1e-3 if (mfma and storage == "f32") else 1e-5
        ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xs, Qs, rtol=tol, tie_rtol=tol, atol=2e-6 if tol < 1e-4 else 6 * 2.0 ** -11 * np.sqrt(2.0 / d))
        if not ok and coarse and nprobe < nlist:
            # fp16-ranked probe lists may swap two near-tied centroids at the probe boundary: every hit must then still
            # come from a list within rounding of the nprobe-th best, with its exact score
            sc = Q @ cen.T
            kth = np.sort(sc, axis=1)[:, -min(nprobe, nlist)][:, None]
            near = sc >= kth - 2e-3
            for qi in range(nq):
                ids = I[qi][I[qi] >= 0]
                assert near[qi][labels[ids]].all(), f"{case}: {msg}"
                np.testing.assert_allclose(D[qi][:len(ids)], Xs[ids] @ Qs[qi], rtol=1e-4, atol=1e-5, err_msg=case)
            ok = True
        assert ok, f"{case}: {msg}"
    index.close()

"""bf16 row storage and the bf16 MFMA / scan twins (north_star: "MFMA fp16/bf16 contraction").

Why it matters here: the reference's producer casts bf16 model outputs to fp32 (core/model_wrapper.py:201-203), so
its embeddings are bf16-VALUED fp32 numbers.  For such rows and queries bf16 storage is lossless: every path (K1
scan, K2 / K2r / K2n contractions, the IVF scans K8 / K8b / K8m, the bf16 shadow of fp32 rows) multiplies exactly
the fp32 oracle's operands, products are exact in fp32, and only the summation order differs: ids equal the fp32
oracle's (ties within 1e-6 relative aside), scores within 1e-5 relative, reconstruct returns the input bit for bit.
Inputs that are not bf16-valued are rounded to nearest even once, at add."""
import numpy as np
import pytest
import torch

from conftest import unit_rows
from oracle import flat_ip, ivf_flat

pytestmark = pytest.mark.gpu


def rbf16(a):
    """fp32 -> bf16 (round to nearest even) -> fp32, in numpy."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16
    return r.view(np.float32).reshape(np.shape(a))


def test_rounding_helper_is_torchs():
    x = unit_rows(50, 300, 3) * np.float32(3.7)
    np.testing.assert_array_equal(rbf16(x), torch.from_numpy(x).bfloat16().float().numpy())


def exact(ssa, index, Xv, Qv, k, path=None, kernel=None, min_score=None):
    D, I = index.search(Qv, k) if min_score is None else index.search(Qv, k, min_score=min_score)
    if path is not None:
        assert index.last_timing()["last_path"] == path, index.last_kernel()
    if kernel is not None:
        assert index.last_kernel() == kernel, index.last_kernel()
    Dr, Ir = flat_ip.search(Xv, Qv, k)
    ok, msg, nm = flat_ip.check_against(D, I, Dr, Ir, Xv, Qv, rtol=1e-5, tie_rtol=1e-6)
    assert ok, msg
    return D, I


# ------------------------------------------------------------------------------------------------- Flat
@pytest.mark.parametrize("n,d,nq,k,kernel", [
    (20000, 4096, 1, 10, "flat_scan_topk_kernel"),        # K1, query in registers
    (20000, 4096, 2, 10, "flat_scan_topk_kernel"),        # K1, queries in LDS
    (20000, 4096, 3, 10, "flat_scan_mm_kernel"),          # K1m: 3-4 queries on 16-bit rows of whole 8 KiB
    (5000, 2048, 4, 100, "flat_scan_mm_kernel"),
    (3000, 100, 7, 10, "flat_scan_topk_kernel"),          # d not a multiple of 64: always the scan
    (20000, 4096, 5, 10, "flat_mfma_r128_kernel"),        # K2r, 16-query form
    (20000, 4096, 40, 10, "flat_mfma_r128_kernel"),       # 64-query form
    (9000, 1024, 128, 64, "flat_mfma_r128_kernel"),       # 128-query form
    (9000, 192, 33, 10, "flat_mfma_n128_kernel"),         # K2n: rows that are not a multiple of 512 elements
    (3000, 4096, 20, 500, "flat_mfma_n128_kernel"),       # ... and k > 256
    (40000, 4096, 300, 10, "flat_mfma_pp_kernel"),        # K2, two query tiles
    (70000, 512, 260, 100, "flat_mfma_pp_kernel"),        # many row tiles per slice: thresholds, folds
    (10000, 4096, 400, 1, "flat_mfma_pp_kernel"),         # top-1 epilogue
])
def test_flat_bf16_storage_equals_fp32_oracle_on_bf16_valued_inputs(ssa, n, d, nq, k, kernel):
    X, Q = rbf16(unit_rows(n, d, 1234)), rbf16(unit_rows(nq, d, 4321))
    index = ssa.IndexFlatIP(d, storage="bf16")
    assert index.storage == "bf16"
    index.add(X[: n // 2])
    index.add(torch.from_numpy(X[n // 2:]).cuda().bfloat16())       # a bf16 device tensor is taken as it is
    assert index.ntotal == n
    np.testing.assert_array_equal(index.reconstruct_batch([0, n // 2 - 1, n // 2, n - 1]), X[[0, n // 2 - 1, n // 2, n - 1]])
    exact(ssa, index, X, Q, k, kernel=kernel)
    index.close()


def test_bf16_storage_rounds_other_inputs_to_nearest_even_once(ssa):
    n, d = 3000, 576
    X = unit_rows(n, d, 9)
    X[5, :8] = [1.0, -1.0, 0.0, 1.00390625, 1.01171875, 3.0e-39, -65504.0, 1e30]   # ties to even, a subnormal, large values
    index = ssa.IndexFlatIP(d, storage="bf16")
    index.add(X)
    Xs = rbf16(X)
    np.testing.assert_array_equal(index.reconstruct_n(0, n), Xs)
    Q = unit_rows(4, d, 10)                                      # fp32 queries stay fp32 on the scan
    D, I = index.search(Q, 10)
    Dr, Ir = flat_ip.search(Xs, Q, 10)
    ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xs, Q, rtol=1e-5, tie_rtol=1e-5)
    assert ok, msg
    # the batched path narrows the queries to bf16 as well: exact against the oracle on the rounded operands
    Qb = unit_rows(40, d, 11)
    D, I = index.search(Qb, 10)
    assert index.last_timing()["last_path"] == 1
    Dr, Ir = flat_ip.search(Xs, rbf16(Qb), 10)
    ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xs, rbf16(Qb), rtol=1e-5, tie_rtol=1e-5)
    assert ok, msg
    index.close()


def test_static_search_and_score_floor_on_bf16_rows(ssa):
    n, d, k = 12000, 4096, 10
    X, Q = rbf16(unit_rows(n, d, 21)), rbf16(unit_rows(6, d, 22))
    index = ssa.IndexFlatIP(d, storage="bf16")
    index.add(X)
    Dr, Ir = flat_ip.search(X, Q, k)
    Qd = torch.from_numpy(Q).cuda()
    D = torch.empty((6, k), dtype=torch.float32, device="cuda"); I = torch.empty((6, k), dtype=torch.int64, device="cuda")
    index.search_static(Qd, k, (D, I))
    torch.cuda.synchronize()
    ok, msg, _ = flat_ip.check_against(D.cpu().numpy(), I.cpu().numpy(), Dr, Ir, X, Q, rtol=1e-5, tie_rtol=1e-6)
    assert ok, msg
    floor = float((np.float64(Dr[0, 4]) + np.float64(Dr[0, 5])) / 2)   # between two scores: no result sits on the floor
    Dm, Im = index.search(Q, k, min_score=floor)
    keep = Dr >= np.float32(floor)
    np.testing.assert_array_equal(Im >= 0, keep)
    np.testing.assert_array_equal(Im[keep], Ir[keep])
    index.close()


def test_bf16_file_round_trip_and_score_ids(ssa, tmp_path):
    n, d = 2000, 256
    X = rbf16(unit_rows(n, d, 31))
    index = ssa.IndexFlatIP(d, storage="bf16")
    index.add(X)
    p = tmp_path / "v.faiss"
    ssa.write_index(index, p)
    back = ssa.read_index(p, storage="bf16")
    assert back.storage == "bf16" and back.ntotal == n
    np.testing.assert_array_equal(back.reconstruct_n(0, n), X)
    as32 = ssa.read_index(p)                                         # the file holds fp32: any storage can read it
    np.testing.assert_array_equal(as32.reconstruct_n(0, n), X)
    neg = unit_rows(3, d, 32)
    got = back.score_ids([0, 7, n - 1], neg)
    np.testing.assert_allclose(got, X[[0, 7, n - 1]].astype(np.float64) @ neg.astype(np.float64).T, rtol=1e-5, atol=2e-6)
    for ix in (index, back, as32):
        ix.close()


# ------------------------------------------------------------------------------------------------- the shadow of fp32 rows
def test_fp32_rows_that_are_bf16_valued_get_a_lossless_bf16_shadow(ssa):
    n, d, k = 30000, 4096, 10
    X, Q = rbf16(unit_rows(n, d, 41)), rbf16(unit_rows(300, d, 42))
    index = ssa.IndexFlatIP(d)                                       # fp32 storage: what the reference's files hold
    index.add(X)
    assert index.shadow_dtype is None
    exact(ssa, index, X, Q, k, path=1)                                # batch 300 on the matrix cores: still the fp32 result
    assert index.shadow_dtype == "bf16" and index.fp16_shadow_rows == n
    exact(ssa, index, X, Q[:40], k, path=1)
    # more bf16-valued rows keep it
    X2 = rbf16(unit_rows(5000, d, 43))
    index.add(X2)
    Xa = np.concatenate([X, X2])
    exact(ssa, index, Xa, Q, k, path=1)
    assert index.shadow_dtype == "bf16" and index.fp16_shadow_rows == n + 5000
    # one row that is not bf16-valued: the shadow is converted afresh as fp16 (11 significant bits beat 8)
    x3 = unit_rows(1, d, 44)
    assert not np.array_equal(x3, rbf16(x3))
    index.add(x3)
    Xb = np.concatenate([Xa, x3])
    D, I = index.search(Q, k)
    assert index.shadow_dtype == "f16" and index.fp16_shadow_rows == n + 5001
    r16 = lambda a: a.astype(np.float16).astype(np.float32)
    Dr, Ir = flat_ip.search(r16(Xb), r16(Q), k)
    ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, r16(Xb), r16(Q), rtol=1e-5, tie_rtol=1e-5)
    assert ok, msg
    index.close()


def test_shadow_dtype_can_be_fixed_by_environment(ssa, monkeypatch):
    n, d = 5000, 512
    X, Q = rbf16(unit_rows(n, d, 51)), rbf16(unit_rows(30, d, 52))
    monkeypatch.setenv("SSMI_SHADOW_DTYPE", "f16")
    a = ssa.IndexFlatIP(d); a.add(X); a.search(Q, 5)
    assert a.shadow_dtype == "f16"
    monkeypatch.setenv("SSMI_SHADOW_DTYPE", "bf16")
    b = ssa.IndexFlatIP(d); b.add(unit_rows(n, d, 53)); b.search(Q, 5)     # not bf16-valued, bf16 all the same
    assert b.shadow_dtype == "bf16"
    a.close(); b.close()


# ------------------------------------------------------------------------------------------------- IVF
def clustered(n, d, ncl, seed):
    rng = np.random.Generator(np.random.Philox(seed))
    centers = rng.standard_normal((ncl, d), dtype=np.float32)
    x = centers[rng.integers(0, ncl, n)] + 0.7 * rng.standard_normal((n, d), dtype=np.float32)
    return rbf16(x / np.linalg.norm(x, axis=1, keepdims=True))


@pytest.mark.parametrize("n,d,nlist,k", [(8000, 4096, 16, 10), (6000, 1024, 32, 100), (4000, 200, 12, 10), (5000, 512, 24, 300)])
def test_ivf_bf16_lists_equal_fp32_oracle_on_bf16_valued_inputs(ssa, n, d, nlist, k):
    X, C = clustered(n, d, nlist, 61), clustered(nlist, d, nlist, 62)
    index = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage="bf16"), d, nlist, ssa.METRIC_INNER_PRODUCT, storage="bf16")
    index.set_centroids(C)
    index.add(X[: n // 2])
    for s in range(n // 2, n, 16):
        index.add(X[s:s + 16])
    index.make_direct_map()
    np.testing.assert_array_equal(index.reconstruct_batch([0, 17, n - 1]), X[[0, 17, n - 1]])
    labels = index.assignments()
    Q = clustered(6, d, nlist, 63)
    for nprobe in (1, 3, nlist):                                      # per-query scan (K8)
        index.nprobe = nprobe
        D, I = index.search(Q, k)
        assert index.last_timing()["last_path"] == 2
        Dr, Ir, _ = ivf_flat.search(X, C, labels, Q, k, nprobe)
        ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, X, Q, rtol=1e-5, tie_rtol=1e-6)
        assert ok, f"nprobe={nprobe}: {msg}"
    # a batch that revisits every list: the list-major scans (K8m on rows of a multiple of 512 elements and k <= 256,
    # K8b otherwise), queries narrowed to bf16 - which they already are
    Qb = clustered(96, d, nlist, 64)
    index.nprobe = nlist
    D, I = index.search(Qb, k)
    assert index.last_timing()["last_grid"] < 0 or "lists" in index.last_kernel()
    if d % 512 == 0 and k <= 256:
        assert index.last_kernel() == "ivf_scan_lists_mfma_kernel"
    Dr, Ir = flat_ip.search(X, Qb, k)
    ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, X, Qb, rtol=1e-5, tie_rtol=1e-6)
    assert ok, msg
    index.close()


def test_ivf_bf16_trained_file_round_trip(ssa, tmp_path):
    n, d, nlist = 6000, 256, 20
    X = clustered(n, d, nlist, 71)
    index = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage="bf16"), d, nlist, ssa.METRIC_INNER_PRODUCT, storage="bf16")
    index.train(X)
    index.add(X)
    index.nprobe = 5
    Q = clustered(20, d, nlist, 72)
    D, I = index.search(Q, 10)
    p = tmp_path / "ivf.faiss"
    ssa.write_index(index, p)
    back = ssa.read_index(p, storage="bf16")
    assert back.storage == "bf16" and back.nprobe == 5
    D2, I2 = back.search(Q, 10)
    np.testing.assert_array_equal(I2, I); np.testing.assert_array_equal(D2, D)
    index.close(); back.close()

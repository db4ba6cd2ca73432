"""Parity at BASELINE.json's full sizes through size-independent properties (the oracle cannot brute-force
16-82 GB in seconds): planted known answers, sortedness, idempotence, agreement of the independent kernels
(K1 streaming scan vs K2 MFMA contraction), float64 re-scoring of the returned ids against the stored rows,
and sampled exactness (no sampled row beats the k-th result)."""
import numpy as np
import pytest
import torch

from conftest import unit_rows

pytestmark = pytest.mark.gpu


def _rescored(index, Q, D, I):
    """float64 inner products of the returned ids' stored rows with their queries."""
    out = np.zeros_like(D, dtype=np.float64)
    for qi in range(Q.shape[0]):
        rows = index.reconstruct_batch(I[qi]).astype(np.float64)
        out[qi] = rows @ Q[qi].astype(np.float64)
    return out


def _basic_properties(D, I, ntotal):
    assert (np.diff(D, axis=1) <= 0).all(), "scores not sorted descending"
    assert (I >= 0).all() and (I < ntotal).all()
    for row in I:
        assert len(set(row.tolist())) == len(row), "duplicate ids in one result"


def test_flat_1m_f32_batch1_configs1(ssa, monkeypatch):
    """BASELINE configs[1]: Flat IP 1M x 4096 fp32, batch 1 (K1), a 64-query batch and the batch-1024 leg
    (K2 on the fp16 shadow; recall@10 against the exact fp32 scan of the same queries)."""
    n, d, k = 1_000_000, 4096, 10
    index = ssa.IndexFlatIP(d)
    index.add_random_unit(n, seed=1234)
    assert index.ntotal == n
    planted = np.array([0, 1, 65535, 65536, 500_000, 999_999], dtype=np.int64)  # segment edges included
    P = index.reconstruct_batch(planted)
    Q = np.concatenate([P, unit_rows(4, d, 4321)])
    D = np.zeros((len(Q), k), np.float32); I = np.zeros((len(Q), k), np.int64)
    for qi in range(len(Q)):  # batch 1, the reference's only mode (core/index_manager.py:371)
        D[qi], I[qi] = [a[0] for a in index.search(Q[qi], k)]
    assert index.last_timing()["last_path"] == 0
    _basic_properties(D, I, n)
    np.testing.assert_array_equal(I[:len(planted), 0], planted)      # a stored row is its own best match
    np.testing.assert_allclose(D[:len(planted), 0], 1.0, atol=1e-5)
    np.testing.assert_allclose(D, _rescored(index, Q, D, I), rtol=1e-5, atol=2e-6)
    D2, I2 = index.search(Q[:8], k)                                    # one batch of 8: K1 multi-query path (queries in LDS)
    assert index.last_timing()["last_path"] == 0
    np.testing.assert_array_equal(I2, I[:8])
    np.testing.assert_allclose(D2, D[:8], rtol=1e-6, atol=1e-7)
    # sampled exactness: no row of a 40k-row sample beats the k-th result unless it is in the result
    sample = np.random.default_rng(7).choice(n, 40_000, replace=False).astype(np.int64)
    S = index.reconstruct_batch(sample)
    sc = S @ Q.T
    for qi in range(len(Q)):
        better = sample[sc[:, qi] > D[qi, -1] * (1 + 1e-5) + 1e-6]
        assert set(better.tolist()) <= set(I[qi].tolist()), f"query {qi}: a sampled row beats the k-th result"
    # K2 (MFMA, fp16 shadow rows and fp16-rounded queries) against K1: ids agree except at near-ties
    Qb = torch.from_numpy(np.concatenate([Q, unit_rows(54, d, 99)])).cuda()
    Dm, Im = index.search(Qb, k)
    assert index.last_timing()["last_path"] == 1
    Dm, Im = Dm.cpu().numpy()[:len(Q)], Im.cpu().numpy()[:len(Q)]
    np.testing.assert_allclose(Dm, D, rtol=1e-3, atol=1e-4)           # the north star's score tolerance
    for qi in range(len(Q)):
        for j in np.nonzero(Im[qi] != I[qi])[0]:                      # an id may differ only at a score tie within 1e-3 rel
            assert abs(D[qi, j] - Dm[qi, j]) <= 1e-3 * abs(D[qi, j]) + 1e-4
    # the batch-1024 leg of configs[1]: the 256-query MFMA kernel on the shadow, 4 query tiles
    Q1k = np.concatenate([Q, unit_rows(1024 - len(Q), d, 777)])
    Qt = torch.from_numpy(Q1k).cuda()
    Dk, Ik = [t.cpu().numpy() for t in index.search(Qt, k)]
    assert index.last_timing()["last_path"] == 1
    _basic_properties(Dk, Ik, n)
    np.testing.assert_array_equal(Ik[:len(planted), 0], planted)
    Dk2, Ik2 = [t.cpu().numpy() for t in index.search(Qt, k)]
    assert np.array_equal(Ik2, Ik) and np.array_equal(Dk2, Dk)        # idempotence
    sel = np.arange(0, 1024, 8)                                       # 128 of the queries through the exact fp32 scan
    monkeypatch.setenv("SSMI_NO_SHADOW", "1")
    De, Ie = index.search(Q1k[sel], k)
    assert index.last_timing()["last_path"] == 0
    monkeypatch.delenv("SSMI_NO_SHADOW")
    np.testing.assert_allclose(Dk[sel], De, rtol=1e-3, atol=1e-4)     # north-star score tolerance
    recall = np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(Ik[sel], Ie)])
    assert recall >= 0.995, recall                                    # fp16 shadow + fp16 queries vs exact fp32: near-tie swaps only
    for a, b, da, db in zip(Ik[sel], Ie, Dk[sel], De):
        for j in np.nonzero(a != b)[0]:
            assert abs(da[j] - db[j]) <= 1e-3 * abs(db[j]) + 1e-4
    np.testing.assert_allclose(De, _rescored(index, Q1k[sel], De, Ie), rtol=1e-5, atol=2e-6)
    index.close()


def test_flat_10m_f16_batch4096_k100_configs2(ssa):
    """BASELINE configs[2]: Flat IP 10M x 4096 fp16 storage (82 GB), batch 4096, k=100 on the MFMA kernel."""
    n, d, k, B = 10_000_000, 4096, 100, 4096
    index = ssa.IndexFlatIP(d, storage="f16")
    index.add_random_unit(n, seed=1234)
    planted = np.array([0, 131071, 131072, 4_999_999, 9_999_999], dtype=np.int64)
    P = index.reconstruct_batch(planted)
    Q = np.concatenate([P, unit_rows(B - len(planted), d, 4321)])
    Qt = torch.from_numpy(Q).cuda()
    Dt, It = index.search(Qt, k)
    assert index.last_timing()["last_path"] == 1
    D, I = Dt.cpu().numpy(), It.cpu().numpy()
    _basic_properties(D, I, n)
    np.testing.assert_array_equal(I[:len(planted), 0], planted)
    np.testing.assert_allclose(D[:len(planted), 0], 1.0, atol=2e-3)   # fp16-rounded rows are not exactly unit norm
    Dt2, It2 = index.search(Qt, k)                                     # idempotence (atomics only order buffers)
    assert torch.equal(It2, It) and torch.equal(Dt2, Dt)
    # K1 (fp32 queries, exact fp32 products of the stored fp16 rows) on 4 of the queries
    sel = np.array([0, 3, 100, 4095])
    D1, I1 = index.search(Q[sel], k)
    assert index.last_timing()["last_path"] == 0
    np.testing.assert_allclose(D[sel], D1, rtol=1e-3, atol=1e-4)
    agree = np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(I[sel], I1)])
    assert agree >= 0.99, agree                                        # differences only among near-tied tail entries
    np.testing.assert_allclose(D1, _rescored(index, Q[sel], D1, I1), rtol=1e-5, atol=2e-6)
    index.close()


def test_ivf_10m_configs3_clustered(ssa):
    """BASELINE configs[3]: IVF4096 nprobe=32 over 10M x 4096 fp16: every returned score is the exact stored-row
    score, results are a subset of what Flat ranks at or above the IVF k-th score, recall@10 on clustered rows."""
    n, d, nlist, nprobe, k = 10_000_000, 4096, 4096, 32, 10
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(5)
    centres = torch.randn((16384, d), generator=g, device=dev)
    centres /= centres.norm(dim=1, keepdim=True)

    def rows(m):
        x = torch.randn((m, d), generator=g, device=dev) * (0.6 / d ** 0.5)
        x += centres[torch.randint(0, 16384, (m,), generator=g, device=dev)]
        return x / x.norm(dim=1, keepdim=True)

    ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage="f16"), d, nlist, ssa.METRIC_INNER_PRODUCT, storage="f16")
    flat = ssa.IndexFlatIP(d, storage="f16")
    ivf.train(rows(256 * nlist))
    assert ivf.is_trained
    for s0 in range(0, n, 65536):
        x = rows(min(65536, n - s0))
        ivf.add(x); flat.add(x)
    assert ivf.ntotal == n and int(ivf.list_sizes().sum()) == n
    ivf.nprobe = nprobe
    Q = rows(400)
    Di, Ii = [t.cpu().numpy() for t in ivf.search(Q, k)]               # list-major batched scan
    assert ivf.last_timing()["last_grid"] < 0
    D1 = np.zeros_like(Di); I1 = np.zeros_like(Ii)
    for qi in range(20):                                               # per-query scan, batch 1
        a, b = ivf.search(Q[qi:qi + 1], k)
        D1[qi], I1[qi] = a.cpu().numpy()[0], b.cpu().numpy()[0]
    assert ivf.last_timing()["last_grid"] > 0
    Df, If = [t.cpu().numpy() for t in flat.search(Q, k)]
    _basic_properties(Di, Ii, n)
    # batch-1 probes use the fp32 coarse quantizer, the batch the fp16 one: same lists except at coarse near-ties
    assert np.mean(I1[:20] == Ii[:20]) > 0.97
    # an id returned by both IVF and Flat carries the same score (same stored row, same arithmetic up to K1/K2 rounding)
    for qi in range(len(Q)):
        common, ia, fb = np.intersect1d(Ii[qi], If[qi], return_indices=True)
        np.testing.assert_allclose(Di[qi][ia], Df[qi][fb], rtol=1e-3, atol=1e-4)
    recall = np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(Ii, If)])
    assert recall > 0.98, recall
    ivf.close(); flat.close()

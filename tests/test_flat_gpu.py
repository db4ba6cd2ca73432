"""GPU parity: HIP Flat search (through the C-ABI via ctypes) vs the CPU oracle.

Tolerance (BASELINE.json north_star): ids bit-exact except at score ties within
1e-3 relative, scores within 1e-3 relative.  The tests use a much tighter 1e-5 for
fp32 storage (the only difference is fp32 summation order).
"""
import numpy as np
import pytest

from conftest import unit_rows
from oracle import flat_ip

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _check(index, X, Q, k, rtol=RTOL):
    D, I = index.search(Q, k)
    assert D.shape == (len(np.atleast_2d(Q)), k) and D.dtype == np.float32
    assert I.shape == D.shape and I.dtype == np.int64
    Dr, Ir = flat_ip.search(X, Q, k)
    ok, msg, nm = flat_ip.check_against(D, I, Dr, Ir, X, Q, rtol=rtol, tie_rtol=rtol)
    assert ok, msg
    return nm


@pytest.mark.parametrize("n,d,nq,k", [
    (10000, 4096, 1, 10),      # the register-query kernel, batch 1
    (10000, 4096, 100, 10),    # BASELINE config 1 shape (LDS-query kernel, 8 queries per pass)
    (3000, 4096, 3, 100),
    (3000, 4096, 1, 500),      # UI maximum k (core/config.py:64)
    (2000, 2048, 1, 10),
    (2000, 1024, 2, 10),
    (5000, 256, 5, 10),        # Matryoshka-truncated dimension
    (777, 64, 4, 10),
    (513, 100, 3, 7),          # d not a multiple of 8 -> padded rows
    (300, 3, 2, 5),
    (1000, 4096, 1, 1),
    (1500, 4096, 1, 1024),     # largest supported k
])
def test_flat_f32_matches_oracle(ssa, n, d, nq, k):
    X = unit_rows(n, d, 1234)
    Q = unit_rows(nq, d, 4321)
    index = ssa.IndexFlatIP(d)
    index.add(X)
    assert index.ntotal == n and index.d == d and index.is_trained
    if nq >= 9 and d % 64 == 0:
        # large batches over fp32 rows run the MFMA contraction on an fp16 shadow copy:
        # north-star tolerance (scores 1e-3 relative, ids equal except at 1e-3-relative ties)
        D, I = index.search(Q, k)
        assert index.last_timing()["last_path"] == 1
        Dr, Ir = flat_ip.search(X, Q, k)
        ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, X, Q, rtol=1e-3, tie_rtol=1e-3, atol=2e-5)
        assert ok, msg
        recall = np.mean([len(set(a) & set(b)) / k for a, b in zip(I, Ir)])
        assert recall > 0.99
    else:
        _check(index, X, Q, k)
    index.close()


def test_flat_f32_large_batch_exact_path_without_shadow(ssa, monkeypatch):
    """SSMI_NO_SHADOW keeps large batches on the exact fp32 streaming scan (8 queries per pass)."""
    monkeypatch.setenv("SSMI_NO_SHADOW", "1")
    X, Q = unit_rows(10000, 4096, 1234), unit_rows(100, 4096, 4321)
    index = ssa.IndexFlatIP(4096)
    index.add(X)
    _check(index, X, Q, 10)
    assert index.last_timing()["last_path"] == 0


def test_fp16_shadow_is_visible_and_can_be_switched_off(ssa):
    """The shadow is not a silent mode: the handle reports it, recall of the shadow path against the exact path is
    measured on oracle-checked data, and fp16_shadow=False keeps every batch on the exact fp32 scan (bit-identical
    to the per-query results)."""
    X, Q = unit_rows(20000, 1024, 1234), unit_rows(64, 1024, 4321)
    index = ssa.IndexFlatIP(1024)
    index.add(X)
    assert index.fp16_shadow and index.fp16_shadow_rows == 0 and not index.fp16_shadow_failed
    Dr, Ir = flat_ip.search(X, Q, 10)
    Ds, Is = index.search(Q, 10)                      # 64 queries: MFMA on the shadow
    assert index.last_timing()["last_path"] == 1 and index.fp16_shadow_rows == 20000
    assert index.last_kernel().startswith("flat_mfma_")
    recall = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(Is, Ir)])
    assert recall >= 0.995, recall
    ok, msg, _ = flat_ip.check_against(Ds, Is, Dr, Ir, X, Q, rtol=1e-3, tie_rtol=1e-3, atol=2e-5)
    assert ok, msg
    index.add(X[:100])                                # the shadow follows adds at the next batched search
    index.search(Q, 10)
    assert index.fp16_shadow_rows == 20100
    index.fp16_shadow = False                         # frees the copy; batches are exact from here on
    assert not index.fp16_shadow and index.fp16_shadow_rows == 0
    De, Ie = index.search(Q, 10)
    assert index.last_timing()["last_path"] == 0 and index.last_kernel() == "flat_scan_mm_kernel"   # 8 queries per pass: K1m
    for i in (0, 17, 63):                             # search(Q)[i] == search(Q[i:i+1]) up to the order of the fp32 sums
        D1, I1 = index.search(Q[i:i + 1], 10)         # (one query: K1's register form, another summation tree)
        np.testing.assert_array_equal(I1[0], Ie[i]); np.testing.assert_allclose(D1[0], De[i], rtol=2e-6, atol=1e-7)
    exact = ssa.IndexFlatIP(1024, fp16_shadow=False)
    exact.add(X)
    D2, I2 = exact.search(Q, 10)
    assert exact.last_timing()["last_path"] == 0
    ok, msg, _ = flat_ip.check_against(D2, I2, Dr, Ir, X, Q, rtol=RTOL, tie_rtol=RTOL)
    assert ok, msg


@pytest.mark.parametrize("n,d,nq,k", [(10000, 4096, 1, 10), (10000, 4096, 4, 10), (3000, 4096, 2, 100),
                                      (2000, 2048, 1, 10), (777, 72, 3, 10), (5000, 72, 8, 10)])
def test_flat_f16_storage_matches_oracle_on_rounded_rows(ssa, n, d, nq, k):
    """fp16 storage, streaming scan (up to 4 queries, or any batch when d is not a multiple of 64): the kernel
    must be exact against the oracle run on the fp16-rounded rows with fp32 queries."""
    X = unit_rows(n, d, 1234)
    Xh = X.astype(np.float16).astype(np.float32)
    Q = unit_rows(nq, d, 4321)
    index = ssa.IndexFlatIP(d, storage="f16")
    index.add(X)
    _check(index, Xh, Q, k)
    assert index.last_timing()["last_path"] == 0
    # and reconstruct returns exactly the rounded rows
    np.testing.assert_array_equal(index.reconstruct_batch([0, n // 2, n - 1]), Xh[[0, n // 2, n - 1]])
    index.close()


def test_k_larger_than_ntotal_pads_with_minus_one(ssa):
    X = unit_rows(5, 128, 1)
    index = ssa.IndexFlatIP(128)
    index.add(X)
    D, I = index.search(unit_rows(2, 128, 2), 10)
    assert (I[:, 5:] == -1).all() and (I[:, :5] >= 0).all()
    assert (D[:, 5:] == flat_ip.NEG_MAX).all()
    assert all(sorted(row[:5]) == [0, 1, 2, 3, 4] for row in I)
    assert (np.diff(D[:, :5], axis=1) <= 0).all()


def test_empty_index(ssa):
    index = ssa.IndexFlatIP(64)
    D, I = index.search(unit_rows(3, 64, 2), 4)
    assert (I == -1).all() and (D == flat_ip.NEG_MAX).all() and index.ntotal == 0


def test_one_dimensional_query_is_one_row(ssa):
    X = unit_rows(100, 256, 1)
    index = ssa.IndexFlatIP(256)
    index.add(X)
    D, I = index.search(X[7], 3)  # core/index_manager.py:364-365 reshapes 1-D queries
    assert D.shape == (1, 3) and I[0, 0] == 7 and abs(D[0, 0] - 1.0) < 1e-5


def test_duplicate_rows_tie_order_is_ascending_id(ssa):
    X = unit_rows(50, 128, 3)
    X[10] = X[3]; X[40] = X[3]; X[41] = X[3]
    index = ssa.IndexFlatIP(128)
    index.add(X)
    D, I = index.search(X[3], 5)
    assert list(I[0, :4]) == [3, 10, 40, 41]
    assert (D[0, :4] == D[0, 0]).all()


def test_incremental_add_and_segment_table(ssa, monkeypatch):
    """Rows added in many small batches across several storage segments."""
    monkeypatch.setenv("SSMI_SEG_SHIFT", "8")  # 256-row segments
    d = 128
    X = unit_rows(1500, d, 5)
    index = ssa.IndexFlatIP(d)
    for s in range(0, 1500, 97):
        index.add(X[s:s + 97])
    assert index.ntotal == 1500
    _check(index, X, unit_rows(4, d, 6), 10)
    np.testing.assert_array_equal(index.reconstruct_n(250, 20), X[250:270])
    with pytest.raises(RuntimeError):
        index.reconstruct(1500)


def test_torch_device_tensors(ssa):
    import torch
    X = unit_rows(4000, 1024, 7)
    Q = unit_rows(3, 1024, 8)
    index = ssa.IndexFlatIP(1024)
    index.add(torch.from_numpy(X).cuda())
    D, I = index.search(torch.from_numpy(Q).cuda(), 10, id_offset=1000)
    assert D.is_cuda and I.is_cuda and I.dtype == torch.int64
    Dr, Ir = flat_ip.search(X, Q, 10)
    ok, msg, _ = flat_ip.check_against(D.cpu().numpy(), I.cpu().numpy() - 1000, Dr, Ir, X, Q, RTOL, RTOL)
    assert ok, msg


def test_topk_merge_of_shards_equals_unsharded(ssa):
    """Logical 3-way row sharding on one GPU + the all-gather consumer kernel."""
    import torch
    from semantic_search_amd.faiss_compat import topk_merge
    d, k = 512, 20
    X = unit_rows(3001, d, 9)
    Q = unit_rows(5, d, 10)
    bounds = [0, 1000, 1007, 3001]
    Ds, Is = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        shard = ssa.IndexFlatIP(d)
        shard.add(X[a:b])
        D, I = shard.search(torch.from_numpy(Q).cuda(), k, id_offset=a)
        Ds.append(D); Is.append(I)
    D, I = topk_merge(torch.stack(Ds), torch.stack(Is))
    Dr, Ir = flat_ip.search(X, Q, k)
    ok, msg, _ = flat_ip.check_against(D.cpu().numpy(), I.cpu().numpy(), Dr, Ir, X, Q, RTOL, RTOL)
    assert ok, msg


def test_packed_gather_layout_merges_in_place(ssa):
    """The sharded path's single-collective layout: per-shard [ids | scores] buffers stacked as an
    all-gather would, merged through strided views (no repacking)."""
    import torch
    from semantic_search_amd.faiss_compat import topk_merge
    from semantic_search_amd.sharded import packed_result_buffers
    d, k, B = 256, 7, 5          # odd B*k: the fp32 half starts on an 8-byte boundary anyway
    X, Q = unit_rows(2000, d, 31), unit_rows(B, d, 32)
    Qd = torch.from_numpy(Q).cuda()
    bounds = [0, 700, 1300, 2000]
    bufs = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        shard = ssa.IndexFlatIP(d)
        shard.add(X[a:b])
        buf, D, I = packed_result_buffers(B, k, Qd.device)
        shard.search(Qd, k, id_offset=a, out=(D, I))
        bufs.append(buf)
    g = torch.stack(bufs)                                   # (world, words): what all_gather_into_tensor yields
    n = B * k
    Ig = g[:, :n].unflatten(1, (B, k))
    Dg = g[:, n:].view(torch.float32)[:, :n].unflatten(1, (B, k))
    assert not Dg.is_contiguous()
    D, I = topk_merge(Dg, Ig)
    Dr, Ir = flat_ip.search(X, Q, k)
    ok, msg, _ = flat_ip.check_against(D.cpu().numpy(), I.cpu().numpy(), Dr, Ir, X, Q, RTOL, RTOL)
    assert ok, msg


def test_errors_are_runtime_errors(ssa):
    index = ssa.IndexFlatIP(64)
    index.add(unit_rows(10, 64, 1))
    with pytest.raises(RuntimeError):
        index.search(unit_rows(1, 64, 2), 2000)  # k above the engine maximum
    with pytest.raises(AssertionError):
        index.add(unit_rows(2, 32, 1))  # wrong dimension (FAISS asserts too)


def test_large_random_unit_matches_oracle_on_readback(ssa):
    """Device-generated rows (the bench's data path): read back, then oracle."""
    d, n = 4096, 20000
    index = ssa.IndexFlatIP(d)
    index.add_random_unit(n, seed=1234)
    X = index.reconstruct_n(0, n)
    assert np.allclose(np.linalg.norm(X, axis=1), 1.0, atol=1e-4)
    assert abs(float(X.mean())) < 1e-3 and abs(float(X.std()) * np.sqrt(d) - 1.0) < 0.02
    _check(index, X, unit_rows(2, d, 4321), 10)


@pytest.mark.parametrize("storage,nq", [("f32", 1), ("f32", 8), ("f16", 3), ("f16", 64), ("f16", 300), ("f32", 40)])
def test_min_score_floor_inside_the_kernels(ssa, storage, nq):
    """ssmi_index_search_min on every Flat path (K1 register / LDS query, K2r, K2, the fp16 shadow): a row must equal
    the unfloored row cut where the score drops below min_score and padded with -1 / -FLT_MAX - exactly the search
    nodes' `r.score >= min_score` filter (nodes/search_nodes.py:152-159) - and agree with the oracle cut the same way."""
    n, d, k = 30000, 1024, 20
    X = unit_rows(n, d, 61)
    Xs = X if storage == "f32" else X.astype(np.float16).astype(np.float32)
    Q = unit_rows(nq, d, 62)
    index = ssa.IndexFlatIP(d, storage=storage)
    index.add(X)
    D0, I0 = index.search(Q, k)
    path = index.last_timing()["last_path"]
    for floor in (float(np.median(D0[:, k // 2])), float(D0.max()) + 1.0, -1.0, float(D0[0, 3])):
        D, I = index.search(Q, k, min_score=floor)
        assert index.last_timing()["last_path"] == path
        keep = D0 >= np.float32(floor)
        np.testing.assert_array_equal(I, np.where(keep, I0, -1))
        np.testing.assert_array_equal(D, np.where(keep, D0, flat_ip.NEG_MAX))
    if path == 0:   # exact paths: the oracle, cut the same way
        Dr, Ir = flat_ip.search(Xs, Q, k)
        floor = float(np.median(Dr[:, 5]))
        D, I = index.search(Q, k, min_score=floor)
        keep = Dr >= np.float32(floor) * (1 + 1e-6)          # entries at the floor within rounding may go either way
        assert ((I >= 0) >= keep).all() and (np.where(keep, I, 0) == np.where(keep, Ir, 0)).all()
    index.close()


def test_one_handle_searched_from_two_streams_and_adds_in_between(ssa):
    """Device-buffer searches return after enqueueing while the handle's workspaces are shared: searches issued
    from two streams are ordered behind one another by the library, and an add waits for the searches in flight
    (it rewrites buffers they read).  Results must equal the serial ones."""
    import torch
    d, k = 4096, 10
    X = unit_rows(60000, d, 51)
    index = ssa.IndexFlatIP(d)
    index.add(X[:50000])
    Q = torch.from_numpy(unit_rows(24, d, 52)).cuda()
    want = [index.search(Q[i:i + 1], k) for i in range(24)]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for i in range(24):
        with torch.cuda.stream(s1 if i % 2 == 0 else s2):
            outs.append(index.search(Q[i:i + 1], k))
    index.add(X[50000:])                       # while the last searches may still be running
    torch.cuda.synchronize()
    for i in range(24):
        assert torch.equal(outs[i][1], want[i][1]) and torch.equal(outs[i][0], want[i][0]), i
    assert index.ntotal == 60000
    D, I = index.search(X[59999], 1)
    assert I[0, 0] == 59999


def test_distinct_indexes_from_distinct_threads(ssa):
    """The boundary contract (include/ssmi.h): calls on ONE handle are serialised by the caller (the
    reference holds a per-index RLock), distinct handles may be used from distinct threads (ComfyUI worker
    threads, global index cache at core/index_manager.py:1481-1504); ctypes drops the GIL during calls."""
    import threading
    d, k = 512, 10
    data = [(unit_rows(6000 + 500 * t, d, 100 + t), unit_rows(3, d, 200 + t)) for t in range(4)]
    expected = [flat_ip.search(X, Q, k) for X, Q in data]
    errors = []

    def worker(t):
        try:
            X, Q = data[t]
            index = ssa.IndexFlatIP(d, storage="f32")
            for s in range(0, len(X), 1000):
                index.add(X[s:s + 1000])
            for _ in range(25):
                D, I = index.search(Q, k)
                ok, msg, _ = flat_ip.check_against(D, I, expected[t][0], expected[t][1], X, Q, rtol=1e-5, tie_rtol=1e-5)
                assert ok, msg
            index.close()
        except Exception as e:  # surfaced in the main thread below
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_hip_path_reproduces_the_output_faiss_publishes_for_its_flat_tutorial(ssa, storage):
    """The one external known answer in reach (tests/golden/faiss_tutorial_flat.json: what FAISS's documentation prints
    for tutorial/python/1-Flat.py), through the C-ABI: 100000 x 65 augmented rows, the published ids bit for bit (K1,
    d = 65 padded to 72), the published distances after the exact L2 <-> inner-product mapping.  fp16 storage rounds the
    rows (and the augmented norm term), so there the ids are required for the sanity block only (distance 0 to itself)
    and elsewhere at least three of four."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import faiss_tutorial as ft
    xb, xq = ft.data()
    for name, q, I_pub, D_pub in ft.blocks(xb, xq):
        Xa, Qa, qn = ft.as_inner_product(xb, q)
        index = ssa.IndexFlatIP(65, storage=storage)
        index.add(Xa)
        D, I = index.search(Qa, 4)
        if storage == "f32":
            np.testing.assert_array_equal(I, I_pub, err_msg=name)
            if D_pub is not None:
                np.testing.assert_allclose(qn[:, None] - 2.0 * D.astype(np.float64), D_pub, rtol=2e-5, atol=2e-5, err_msg=name)
        else:
            assert (I[:, 0] == I_pub[:, 0]).all(), name
            assert min(len(set(a) & set(b)) for a, b in zip(I, I_pub)) >= 3, (name, I, I_pub)
        index.close()

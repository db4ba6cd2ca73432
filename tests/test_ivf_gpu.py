"""GPU parity of the IVF-Flat path (coarse quantizer, list assignment, probe planning,
bucketed list scan) against the oracle, through the C-ABI."""
import numpy as np
import pytest

from conftest import unit_rows
from oracle import flat_ip, ivf_flat

pytestmark = pytest.mark.gpu


def r16(a):
    return a.astype(np.float16).astype(np.float32)


def clustered_rows(n, d, ncl, seed):
    """Unit rows with cluster structure so that IVF lists are meaningfully different."""
    rng = np.random.Generator(np.random.Philox(seed))
    centers = rng.standard_normal((ncl, d), dtype=np.float32)
    x = centers[rng.integers(0, ncl, n)] + 0.7 * rng.standard_normal((n, d), dtype=np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def make_ivf(ssa, d, nlist, storage="f32"):
    return ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage=storage), d, nlist, ssa.METRIC_INNER_PRODUCT, storage=storage)


@pytest.mark.parametrize("n,d,nlist,storage", [(6000, 256, 40, "f32"), (3000, 4096, 16, "f32"), (5000, 1024, 64, "f16"),
                                                (2000, 100, 10, "f32")])
def test_given_centroids_assignment_and_search_match_oracle(ssa, n, d, nlist, storage):
    X = clustered_rows(n, d, nlist, 11)
    C = clustered_rows(nlist, d, nlist, 12)
    index = make_ivf(ssa, d, nlist, storage)
    assert not index.is_trained and index.nlist == nlist and index.nprobe == 1
    index.set_centroids(C)
    assert index.is_trained
    np.testing.assert_array_equal(index.get_centroids(), C)
    index.add(X[: n // 2])
    for s in range(n // 2, n, 16):  # the reference adds 16 vectors at a time (core/config.py:59)
        index.add(X[s:s + 16])
    assert index.ntotal == n
    labels = index.assignments()
    Cq = r16(C) if False else C
    assert ivf_flat.assignment_is_valid(X, C, labels, rtol=2e-3 if n // 2 >= 9 else 1e-5), "vectors not in their best list"
    np.testing.assert_array_equal(index.list_sizes(), np.bincount(labels, minlength=nlist))
    Xs = r16(X) if storage == "f16" else X
    Q = clustered_rows(5, d, nlist, 13)
    for nprobe in (1, 3, nlist):
        index.nprobe = nprobe
        for k in (10, 100):
            D, I = index.search(Q, k)
            assert index.last_timing()["last_path"] == 2
            Dr, Ir, _ = ivf_flat.search(Xs, C, labels, Q, k, nprobe)
            ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xs, Q, rtol=1e-5, tie_rtol=1e-5)
            assert ok, f"nprobe={nprobe} k={k}: {msg}"
    # probing every list is exact search
    index.nprobe = nlist
    D, I = index.search(Q, 10)
    Dr, Ir = flat_ip.search(Xs, Q, 10)
    ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xs, Q, rtol=1e-5, tie_rtol=1e-5)
    assert ok, msg
    index.close()


def test_large_query_batch_uses_mfma_coarse_quantizer(ssa):
    n, d, nlist = 8000, 512, 32
    X, C = clustered_rows(n, d, nlist, 21), clustered_rows(nlist, d, nlist, 22)
    index = make_ivf(ssa, d, nlist)
    index.set_centroids(C)
    index.add(X)
    labels = index.assignments()
    Q = clustered_rows(64, d, nlist, 23)
    index.nprobe = 4
    D, I = index.search(Q, 10)
    # coarse scores come from fp16-rounded centroids and queries on this path
    Dr, Ir, _ = ivf_flat.search(X, C, labels, Q, 10, 4, C_coarse=r16(C), Q_coarse=r16(Q))
    agree = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(I, Ir)])
    assert agree > 0.97, agree
    index.nprobe = nlist
    D, I = index.search(Q, 10)
    # every list probed by every query: the list-major matrix-core scan, which over fp32 lists reads their fp16
    # shadow with fp16-rounded queries (exact against the oracle on the rounded operands) ...
    assert index.last_timing()["last_grid"] < 0
    Dr, Ir = flat_ip.search(r16(X), r16(Q), 10)
    ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, r16(X), r16(Q), rtol=1e-5, tie_rtol=1e-5)
    assert ok, msg
    # ... and inside the north-star tolerance against the fp32 operands
    Dr, Ir = flat_ip.search(X, Q, 10)
    ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, X, Q, rtol=1e-3, tie_rtol=1e-3, atol=1e-4)
    assert ok, msg


@pytest.mark.parametrize("spherical", [True, False])
def test_train_kmeans_quality_and_recall(ssa, monkeypatch, spherical):
    """train() = k-means of the coarse quantizer (core/index_factory.py:191).  Default: spherical, what FAISS's
    IndexIVF does for METRIC_INNER_PRODUCT (core/index_factory.py:144-146) [FAISS-lib, parity unpinned]: centroids
    come back L2-normalised; SSMI_IVF_SPHERICAL=0 (read when the index is created) keeps plain means."""
    if not spherical:
        monkeypatch.setenv("SSMI_IVF_SPHERICAL", "0")
    n, d, nlist = 20000, 256, 50
    X = clustered_rows(n, d, 50, 31)
    index = make_ivf(ssa, d, nlist)
    with pytest.raises(RuntimeError):
        index.add(X[:10])           # untrained
    D0, I0 = index.search(X[:2], 5)   # FAISS: IndexIVF::search does not test is_trained; no list to probe, no result
    assert (I0 == -1).all() and (D0 == flat_ip.NEG_MAX).all()
    with pytest.raises(RuntimeError):
        index.train(X[:10])         # fewer points than clusters
    assert not index.is_trained
    index.train(X)
    assert index.is_trained
    C = index.get_centroids()
    norms = np.linalg.norm(C.astype(np.float64), axis=1)
    if spherical:
        np.testing.assert_allclose(norms, 1.0, atol=2e-6)
    else:
        assert norms.max() < 0.999   # means of unit rows that do not coincide lie inside the sphere
    # Lloyd iterations end in a local optimum that depends on the initial centroids (the oracle's objective spreads 6-8 %
    # over its seeds on these rows): the index must land inside the oracle's range, not above one run of it
    objs = [ivf_flat.kmeans(X, nlist, seed=sd, spherical=spherical)[1] for sd in (1234, 1, 2, 3)]
    obj = ivf_flat.objective(X, C)
    assert 0.985 * min(objs) <= obj <= 1.02 * max(objs), (obj, objs)
    index.add(X)
    assert int(index.list_sizes().sum()) == n
    assert ivf_flat.assignment_is_valid(X, C, index.assignments(), rtol=2e-3)
    Q = clustered_rows(100, d, 50, 32)
    index.nprobe = 10
    D, I = index.search(Q, 10)
    Dr, Ir = flat_ip.search(X, Q, 10)
    recall = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(I, Ir)])
    assert recall > 0.9, recall
    assert (np.diff(D, axis=1) <= 0).all()


def test_spherical_kmeans_on_rows_of_unequal_norm(ssa):
    """Rows that are NOT unit vectors: plain means keep the scale of their members, so with max-inner-product
    assignment the long clusters' centroids attract everything; spherical centroids (FAISS's choice for the inner
    product) compare by direction only.  Checked against the oracle's k-means in the same mode."""
    n, d, nlist = 12000, 128, 24
    rng = np.random.default_rng(77)
    X = clustered_rows(n, d, nlist, 33) * rng.uniform(0.5, 4.0, (n, 1)).astype(np.float32)
    index = make_ivf(ssa, d, nlist)
    index.train(X)
    C = index.get_centroids()
    np.testing.assert_allclose(np.linalg.norm(C.astype(np.float64), axis=1), 1.0, atol=2e-6)
    objs = [ivf_flat.kmeans(X, nlist, seed=sd, spherical=True)[1] for sd in (1234, 1, 2, 3)]
    assert 0.985 * min(objs) <= ivf_flat.objective(X, C) <= 1.02 * max(objs), objs
    index.add(X)
    sizes = index.list_sizes()
    assert sizes.max() < 6 * n / nlist, sizes   # no list swallows the index


def test_reconstruct_needs_direct_map(ssa):
    n, d, nlist = 500, 64, 8
    X, C = clustered_rows(n, d, nlist, 41), clustered_rows(nlist, d, nlist, 42)
    index = make_ivf(ssa, d, nlist)
    index.set_centroids(C)
    index.add(X[:300])
    with pytest.raises(RuntimeError):
        index.reconstruct(0)        # like FAISS without a direct map (core/index_factory.py:275 catches it)
    index.make_direct_map()
    np.testing.assert_array_equal(index.reconstruct(17), X[17])
    index.add(X[300:])              # the map is maintained by later adds
    np.testing.assert_array_equal(index.reconstruct_batch([0, 299, 300, 499]), X[[0, 299, 300, 499]])
    with pytest.raises(RuntimeError):
        index.reconstruct(500)


def test_empty_lists_and_short_results(ssa):
    d, nlist = 64, 8
    C = clustered_rows(nlist, d, nlist, 51)
    index = make_ivf(ssa, d, nlist)
    index.set_centroids(C)
    D, I = index.search(C[:2], 5)
    assert (I == -1).all()
    index.add(C[:3] * 0.5)          # three vectors, three lists
    index.nprobe = 2
    D, I = index.search(C[0], 5)
    assert I[0, 0] == 0 and (I[0, 2:] == -1).all() and (D[0, 2:] == flat_ip.NEG_MAX).all()


def test_list_sharded_ivf_equals_unsharded(ssa):
    """Logical 3-way sharding by list on one GPU: same centroids everywhere, list l on shard l % 3,
    global ids; merging the shards' results must reproduce the single-index search."""
    import torch
    from semantic_search_amd.faiss_compat import topk_merge
    from semantic_search_amd.sharded import ShardedIVFFlat
    n, d, nlist, world = 4000, 128, 24, 3
    X, C = clustered_rows(n, d, nlist, 61), clustered_rows(nlist, d, nlist, 62)
    whole = make_ivf(ssa, d, nlist)
    whole.set_centroids(C)
    whole.add(X)
    shards = [ShardedIVFFlat(d, nlist, rank=r, world=world) for r in range(world)]
    for sh in shards:
        sh.local.set_centroids(C)
        for s in range(0, n, 1000):
            sh.add(X[s:s + 1000])
        assert sh.ntotal == n
    sizes = np.stack([sh.local.list_sizes() for sh in shards])
    np.testing.assert_array_equal(sizes.sum(axis=0), whole.list_sizes())
    for l in range(nlist):
        assert (sizes[:, l] > 0).sum() <= 1 and (sizes[l % world, l] == whole.list_sizes()[l])
    Q = torch.from_numpy(clustered_rows(6, d, nlist, 63)).cuda()
    for nprobe in (1, 5, nlist):
        whole.nprobe = nprobe
        Dw, Iw = whole.search(Q, 10)
        parts = []
        for sh in shards:
            sh.nprobe = nprobe
            parts.append(sh.local.search(Q, 10))
        D, I = topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
        assert torch.equal(I, Iw) and torch.equal(D, Dw), nprobe


@pytest.mark.parametrize("n,d,nlist,storage,nq,list_mfma", [
    (6000, 256, 40, "f32", 37, "0"), (3000, 4096, 16, "f32", 20, "0"), (5000, 1024, 64, "f16", 130, "0"),
    (2000, 100, 10, "f32", 16, "0"),
    (5000, 1024, 64, "f16", 130, "1"),    # matrix-core form: 16 queries per list pass (130*nprobe/64 probes per list)
    (9000, 512, 8, "f16", 200, "1"),      # ... 64 / 128 queries per list pass, lists of > 1000 rows, ragged last pages
    (3000, 4096, 6, "f16", 33, "1"),
    (3000, 4096, 16, "f32", 20, "1"),     # fp32 lists: the matrix-core form runs on the fp16 shadow of the slot store
    (5000, 1024, 64, "f32", 130, "1"),
])
def test_list_major_batched_scan_matches_oracle_and_per_query_scan(ssa, monkeypatch, n, d, nlist, storage, nq, list_mfma):
    """K8b (ivf_lists.hip) and K8m (ivf_lists_mfma.hip): the list-major scan of a query batch must give what the
    per-query scan gives, and both must match the oracle fed with the index's own list assignment.  The
    register-FMA form is exact in fp32; the matrix-core form rounds the queries to fp16 (north-star tolerance) and,
    over fp32 lists, reads an fp16 shadow copy of the rows."""
    monkeypatch.setenv("SSMI_IVF_LIST_MFMA", list_mfma)
    X = clustered_rows(n, d, nlist, 71)
    C = clustered_rows(nlist, d, nlist, 72)
    index = make_ivf(ssa, d, nlist, storage)
    index.set_centroids(C)
    index.add(X)
    labels = index.assignments()
    Xs = r16(X) if storage == "f16" else X
    Q = clustered_rows(nq, d, nlist, 73)
    for nprobe in (1, 3, nlist):
        index.nprobe = nprobe
        for k in (10, 100, 300):
            mfma = list_mfma == "1" and k <= 256          # the matrix-core form covers k <= 256
            monkeypatch.setenv("SSMI_IVF_LISTMAJOR", "1")
            D, I = index.search(Q, k)
            assert index.last_timing()["last_grid"] < 0, "list-major path not taken"
            if mfma:
                assert -index.last_timing()["last_grid"] in (16, 64, 128)
            monkeypatch.setenv("SSMI_IVF_LISTMAJOR", "0")
            D2, I2 = index.search(Q, k)
            assert index.last_timing()["last_grid"] > 0
            coarse = dict(C_coarse=r16(C), Q_coarse=r16(Q)) if (nq >= 9 and d % 64 == 0) else {}
            if mfma:
                # fp16 rounding of a unit query moves a score by ~2^-11 * |q||x| / sqrt(d) ~ 1.5e-5 absolute:
                # the north star's 1e-3 relative bound needs an absolute floor for scores near zero
                ok, msg, _ = flat_ip.check_against(D, I, D2, I2, Xs, Q, rtol=1e-3, tie_rtol=1e-3, atol=1e-4)
                assert ok, f"nprobe={nprobe} k={k} vs per-query scan: {msg}"
                Xs_m = r16(X)  # rows as the matrix cores see them (the shadow of fp32 lists is rounded like fp16 storage)
                Dr, Ir, _ = ivf_flat.search(Xs_m, C, labels, r16(Q), k, nprobe, **coarse)
            else:
                np.testing.assert_array_equal(I, I2)
                np.testing.assert_array_equal(D, D2)
                Dr, Ir, _ = ivf_flat.search(Xs, C, labels, Q, k, nprobe, **coarse)
            Qs = r16(Q) if mfma else Q
            Xc = r16(X) if mfma else Xs
            if nprobe == nlist or not coarse:
                ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xc, Qs, rtol=1e-5, tie_rtol=1e-5)
                assert ok, f"nprobe={nprobe} k={k}: {msg}"
            else:  # fp16 coarse scores may swap two near-tied centroids at the probe boundary
                agree = np.mean([len(set(a[a >= 0]) & set(b[b >= 0])) / max(1, (b >= 0).sum()) for a, b in zip(I, Ir)])
                assert agree > 0.97, (nprobe, k, agree)
    monkeypatch.delenv("SSMI_IVF_LISTMAJOR")
    index.close()


def test_batched_scan_is_repeatable_between_other_batches(ssa):
    """The list-major MFMA scan (16-query groups: the register ring is deepest there) must return the same
    bits every time, with another batch in between so that nothing of the previous pass survives in the
    workspaces.  (A compiler-inserted copy of in-flight ring registers once made it differ run to run;
    tests/test_ring_regs_cpu.py checks the assembly, this checks the behaviour.)"""
    import torch
    d, n, nlist, nq = 4096, 200_000, 256, 64
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)

    def rows(m):
        x = torch.randn((m, d), generator=g, device=dev)
        return x / x.norm(dim=1, keepdim=True)

    ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage="f16"), d, nlist, ssa.METRIC_INNER_PRODUCT, storage="f16")
    ivf.train(rows(16 * nlist))
    for s0 in range(0, n, 50_000):
        ivf.add(rows(50_000))
    ivf.nprobe = 8
    Q, Q2 = rows(nq), rows(nq)
    D0, I0 = ivf.search(Q, 10)
    assert ivf.last_timing()["last_grid"] == -16
    D0, I0 = D0.clone(), I0.clone()
    for _ in range(5):
        ivf.search(Q2, 10)
        D, I = ivf.search(Q, 10)
        assert torch.equal(I, I0) and torch.equal(D, D0)


@pytest.mark.parametrize("no_tail", ["", "1"])
def test_many_probes_and_both_merge_forms(ssa, monkeypatch, no_tail):
    """nprobe > 64: the coarse quantizer's merge keeps lists of more than 64 keys, so the probe plan comes from
    the separate planning kernel instead of the merge tail; SSMI_NO_TAIL=1: the scans leave every merge to K3."""
    if no_tail:
        monkeypatch.setenv("SSMI_NO_TAIL", no_tail)
    n, d, nlist = 12_000, 512, 200
    X, C = clustered_rows(n, d, nlist, 41), clustered_rows(nlist, d, nlist, 42)
    index = make_ivf(ssa, d, nlist)
    index.set_centroids(C)
    index.add(X)
    labels = index.assignments()
    for nq in (1, 5):
        Q = clustered_rows(nq, d, nlist, 43 + nq)
        for nprobe, k in ((20, 10), (100, 10), (150, 70), (200, 300)):
            index.nprobe = nprobe
            D, I = index.search(Q, k)
            assert index.last_timing()["last_grid"] > 0  # per-query scan
            Dr, Ir, _ = ivf_flat.search(X, C, labels, Q, k, nprobe)
            ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, X, Q, rtol=1e-5, tie_rtol=1e-5)
            assert ok, f"nq={nq} nprobe={nprobe} k={k}: {msg}"
    index.close()


@pytest.mark.parametrize("storage,nq,listmajor", [("f32", 1, "0"), ("f16", 5, "0"), ("f16", 64, "1"), ("f32", 40, "1")])
def test_min_score_floor_on_the_ivf_scans(ssa, monkeypatch, storage, nq, listmajor):
    """The score floor on the per-query IVF scans (thresholds start at the floor, merge tail cuts) and on the
    list-major scans (cut in the merge): floored rows == unfloored rows cut at the floor."""
    monkeypatch.setenv("SSMI_IVF_LISTMAJOR", listmajor)
    n, d, nlist, k = 8000, 512, 16, 20
    X, C = clustered_rows(n, d, nlist, 91), clustered_rows(nlist, d, nlist, 92)
    index = make_ivf(ssa, d, nlist, storage)
    index.set_centroids(C)
    index.add(X)
    index.nprobe = 5
    Q = clustered_rows(nq, d, nlist, 93)
    D0, I0 = index.search(Q, k)
    assert (index.last_timing()["last_grid"] < 0) == (listmajor == "1")
    for floor in (float(np.median(D0[:, k // 2])), 2.0, -1.0):
        D, I = index.search(Q, k, min_score=floor)
        keep = D0 >= np.float32(floor)
        np.testing.assert_array_equal(I, np.where(keep, I0, -1))
        np.testing.assert_array_equal(D, np.where(keep, D0, flat_ip.NEG_MAX))
    index.close()


def test_fp32_lists_shadow_follows_adds(ssa):
    """The fp16 shadow of an fp32 slot store is built by the first batched search and must then follow every add
    (rows appended into half-filled pages included): a batched search after more adds sees all rows."""
    d, nlist = 512, 16
    X, C = clustered_rows(4000, d, nlist, 81), clustered_rows(nlist, d, nlist, 82)
    index = make_ivf(ssa, d, nlist, "f32")
    index.set_centroids(C)
    index.add(X[:1500])
    Q = X[np.arange(0, 4000, 100)]          # 40 queries: rows of the index-to-be
    index.nprobe = nlist
    D, I = index.search(Q, 1)
    assert index.last_timing()["last_grid"] < 0          # list-major (matrix-core) path, shadow created
    assert (I[:15, 0] == np.arange(0, 1500, 100)).all()
    for s in range(1500, 4000, 37):                      # ragged adds into partially filled pages
        index.add(X[s:s + 37])
    D, I = index.search(Q, 1)
    assert index.last_timing()["last_grid"] < 0
    assert (I[:, 0] == np.arange(0, 4000, 100)).all() and (D[:, 0] > 0.999).all()
    index.close()


def _file_lists(path, d, nlist):
    """ids of every inverted list in storage order, parsed from the IwFl file (layout: csrc/persist.hip header)."""
    import struct
    raw = path.read_bytes()
    at = 4 + 33 + 16 + 4 + 33 + 8 + nlist * d * 4 + 9
    assert raw[at:at + 4] == b"ilar" and raw[at + 20:at + 24] == b"full"
    at += 32
    sizes = np.frombuffer(raw, dtype="<u8", count=nlist, offset=at).astype(np.int64)
    at += nlist * 8
    out = []
    for s in sizes:
        at += int(s) * d * 4
        out.append(np.frombuffer(raw, dtype="<i8", count=int(s), offset=at).copy())
        at += int(s) * 8
    assert at == len(raw)
    return out


@pytest.mark.parametrize("n,d,nlist,steps", [(150000, 32, 3, (150000,)),        # few long lists, several blocks and chunks
                                              (70000, 32, 40000, (70000,)),     # more lists than the LDS cursors hold
                                              (9000, 64, 37, (1, 30, 31, 33, 1000, 7905))])  # page tails of every fill
def test_device_bucketing_is_stable_and_complete(ssa, tmp_path, n, d, nlist, steps):
    """K7b (csrc/ivf_bucket.hip): every added vector lands in its list exactly once, in add order, whatever the
    split into add calls - against numpy's stable argsort of the labels."""
    X = clustered_rows(n, d, min(nlist, 64), 71)
    rng = np.random.Generator(np.random.Philox(72))
    C = X[rng.choice(n, nlist, replace=False)] if nlist > 64 else clustered_rows(nlist, d, nlist, 73)
    index = make_ivf(ssa, d, nlist)
    index.set_centroids(C)
    index.make_direct_map()
    at = 0
    for s in steps:
        index.add(X[at:at + s])
        at += s
    assert at == n and index.ntotal == n
    labels = index.assignments()
    if len(steps) == 1:
        np.testing.assert_array_equal(labels, index.assign(X))
    else:  # single vectors are assigned on fp32 centroids, batches on their fp16 copies
        assert ivf_flat.assignment_is_valid(X, C, labels, rtol=2e-3, atol=6 * 2.0 ** -11 * np.sqrt(2.0 / X.shape[1]))
    np.testing.assert_array_equal(index.list_sizes(), np.bincount(labels, minlength=nlist))
    ssa.write_index(index, tmp_path / "ivf.faiss")
    order = np.argsort(labels, kind="stable")
    bounds = np.concatenate([[0], np.cumsum(np.bincount(labels, minlength=nlist))])
    for l, ids in enumerate(_file_lists(tmp_path / "ivf.faiss", d, nlist)):
        np.testing.assert_array_equal(ids, order[bounds[l]:bounds[l + 1]], err_msg=f"list {l}")
    pick = rng.choice(n, 50, replace=False)
    np.testing.assert_array_equal(index.reconstruct_batch(pick), X[pick])   # the direct map followed every add
    # explicit ids through add_preassigned take the same path with host labels
    twin = make_ivf(ssa, d, nlist)
    twin.set_centroids(C)
    ids = (np.arange(n, dtype=np.int64) * 3 + 7)
    twin.add_preassigned(X[: n // 2], labels[: n // 2], ids[: n // 2])
    twin.add_preassigned(X[n // 2:], labels[n // 2:], ids[n // 2:])
    ssa.write_index(twin, tmp_path / "twin.faiss")
    for l, got in enumerate(_file_lists(tmp_path / "twin.faiss", d, nlist)):
        np.testing.assert_array_equal(got, ids[order[bounds[l]:bounds[l + 1]]], err_msg=f"list {l}")
    with pytest.raises(RuntimeError, match="out of range"):
        twin.add_preassigned(X[:2], np.array([0, nlist], dtype=np.int32), ids[:2])
    index.close(); twin.close()

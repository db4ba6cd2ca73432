"""The three-stream gather/merge pipeline of the sharded search (semantic_search_amd/sharded.py)
on one GPU: a one-rank RCCL group exercises the same streams, events and buffer ring that the
multi-GPU bench uses; results must equal the direct local search, step by step, when many searches
are enqueued back to back without synchronising."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from conftest import unit_rows

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def one_rank_group():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("pipelined", [False, True])
@pytest.mark.parametrize("B,k", [(1, 10), (5, 100), (64, 10)])
def test_gathered_search_equals_direct(ssa, one_rank_group, B, k, pipelined):
    from semantic_search_amd.sharded import ShardedFlatIP
    n, d, steps = 20000, 512, 24
    X = unit_rows(n, d, 5)
    idx = ShardedFlatIP(d, storage="f32", device=0)
    idx.add_local(X)
    idx.finalize(device=torch.device("cuda", 0))
    idx.row_offset = 1000  # ids must carry the shard offset through gather and merge
    idx._force_pipeline = True
    idx._pipelined = pipelined   # default one-stream sequence and the opt-in three-stream pipeline
    Q = torch.from_numpy(unit_rows(steps * B, d, 6)).cuda().view(steps, B, d)
    outs = [idx.search(Q[i], k) for i in range(steps)]  # enqueued back to back, ring of 4 slots reused
    torch.cuda.synchronize()
    for i in range(steps):
        Dr, Ir = idx.local.search(Q[i], k, id_offset=1000)
        torch.cuda.synchronize()
        assert torch.equal(outs[i][1], Ir), f"step {i}: ids differ"
        assert torch.equal(outs[i][0], Dr), f"step {i}: scores differ"


def test_routed_ivf_add_and_gathered_search_equal_plain_index(ssa, one_rank_group):
    """ShardedIVFFlat on a one-rank RCCL group with the collectives forced on: broadcast centroids, routed add
    (ssmi_index_assign -> all-to-all over RCCL -> ssmi_index_add_preassigned, device tensors end to end) and
    the gathered search must give exactly what a plain IndexIVFFlat with the same centroids gives."""
    from semantic_search_amd.sharded import ShardedIVFFlat
    n, d, nlist, k = 30000, 512, 64, 10
    X = torch.from_numpy(unit_rows(n, d, 7)).cuda()
    sh = ShardedIVFFlat(d, nlist, storage="f16", device=0)
    sh._force_pipeline = True
    sh.train(X[:8192])
    plain = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage="f16"), d, nlist, ssa.METRIC_INNER_PRODUCT, storage="f16")
    plain.set_centroids(sh.local.get_centroids())
    for s0 in range(0, n, 7000):
        sh.add_local(X[s0:s0 + 7000])
        plain.add(X[s0:s0 + 7000])
    assert sh.ntotal == n and sh.local.ntotal == n
    np.testing.assert_array_equal(sh.local.list_sizes(), plain.list_sizes())
    np.testing.assert_array_equal(sh.local.assignments(), plain.assignments())
    Q = torch.from_numpy(unit_rows(40, d, 8)).cuda()
    for nprobe, B in ((1, 1), (8, 1), (8, 40), (64, 3)):
        sh.nprobe = plain.nprobe = nprobe
        D, I = sh.search(Q[:B], k)
        Dp, Ip = plain.search(Q[:B], k)
        torch.cuda.synchronize()
        assert torch.equal(I, Ip) and torch.equal(D, Dp), (nprobe, B)
    with pytest.raises(ValueError):
        sh.search(Q[:1].cpu(), k)                       # host queries cannot feed the device-side gather
    with pytest.raises(RuntimeError):
        sh.local.add_preassigned(X[:2], [0, nlist], [n, n + 1])   # list out of range: refused, nothing added
    assert sh.local.ntotal == n


def _rescored(index, Q, D, I, off):
    out = np.zeros_like(D, dtype=np.float64)
    for qi in range(Q.shape[0]):
        rows = index.reconstruct_batch(I[qi] - off).astype(np.float64)
        out[qi] = rows @ Q[qi].astype(np.float64)
    return out


def test_configs4_per_shard_shape_through_gather_and_merge(ssa, one_rank_group):
    """BASELINE configs[4], one shard of it: 100M x 4096 fp16 rows over 8 GPUs = 12.5M rows (102.4 GB) per GPU.
    The shard is searched through ShardedFlatIP's gather/merge path (RCCL all-gather of the packed (B,k)
    results + ssmi_topk_merge) with a nonzero row offset, batch 1 (K1) and batch 4096 (K2), k=10.  Full-size
    properties: planted rows (segment edges included) come back first under their GLOBAL ids, sortedness,
    float64 re-scoring of the returned ids, sampled exactness, K1/K2 agreement."""
    from semantic_search_amd.sharded import ShardedFlatIP
    n, d, k = 12_500_000, 4096, 10
    off = 3 * n                                           # rank 3's offset in the 100M-row database
    idx = ShardedFlatIP(d, storage="f16", device=0)
    idx.add_local_random_unit(n, seed=1234 + 3)
    idx.finalize(device=torch.device("cuda", 0))
    assert idx.ntotal == n
    idx.row_offset = off
    idx._force_pipeline = True
    local = idx.local
    planted = np.array([0, 131071, 131072, 6_250_000, n - 1], dtype=np.int64)
    P = local.reconstruct_batch(planted)
    Q = np.concatenate([P, unit_rows(3, d, 4321)])
    Qd = torch.from_numpy(Q).cuda()
    D = np.zeros((len(Q), k), np.float32); I = np.zeros((len(Q), k), np.int64)
    for qi in range(len(Q)):                              # batch 1: K1 + merge tail, then gather + merge
        a, b = idx.search(Qd[qi:qi + 1], k)
        D[qi], I[qi] = a.cpu().numpy()[0], b.cpu().numpy()[0]
    assert local.last_timing()["last_path"] == 0
    assert (np.diff(D, axis=1) <= 0).all() and (I >= off).all() and (I < off + n).all()
    np.testing.assert_array_equal(I[:len(planted), 0], planted + off)
    np.testing.assert_allclose(D[:len(planted), 0], 1.0, atol=2e-3)
    np.testing.assert_allclose(D, _rescored(local, Q, D, I, off), rtol=1e-5, atol=2e-6)
    sample = np.random.default_rng(11).choice(n, 30_000, replace=False).astype(np.int64)
    sc = local.reconstruct_batch(sample) @ Q.T
    for qi in range(len(Q)):
        better = sample[sc[:, qi] > D[qi, -1] * (1 + 1e-5) + 1e-6] + off
        assert set(better.tolist()) <= set(I[qi].tolist()), f"query {qi}: a sampled row beats the k-th result"
    # batch 4096 on the MFMA contraction through the same gather/merge path
    Qb = torch.cat([Qd, torch.from_numpy(unit_rows(4096 - len(Q), d, 99)).cuda()])
    Dm, Im = idx.search(Qb, k)
    torch.cuda.synchronize()
    assert local.last_timing()["last_path"] == 1
    Dm, Im = Dm.cpu().numpy(), Im.cpu().numpy()
    assert (np.diff(Dm, axis=1) <= 0).all() and (Im >= off).all() and (Im < off + n).all()
    np.testing.assert_array_equal(Im[:len(planted), 0], planted + off)
    np.testing.assert_allclose(Dm[:len(Q)], D, rtol=1e-3, atol=1e-4)
    for qi in range(len(Q)):
        for j in np.nonzero(Im[qi] != I[qi])[0]:          # an id may differ only at a score tie within 1e-3 rel
            assert abs(D[qi, j] - Dm[qi, j]) <= 1e-3 * abs(D[qi, j]) + 1e-4
    sel = np.array([100, 2000, 4095])
    np.testing.assert_allclose(Dm[sel], _rescored(local, Qb[sel].cpu().numpy(), Dm[sel], Im[sel], off), rtol=1e-3, atol=1e-4)
    Dm2, Im2 = idx.search(Qb, k)                          # idempotence
    torch.cuda.synchronize()
    assert np.array_equal(Im2.cpu().numpy(), Im) and np.array_equal(Dm2.cpu().numpy(), Dm)
    local.close()


def _clustered_device_rows(d, n_total, seed, dev, ncl=16384, sigma=0.6):
    """The bench's clustered generator (bench.py: ivf_rows_factory): unit rows around `ncl` shared random centres."""
    g = torch.Generator(device=dev).manual_seed(seed)
    gc = torch.Generator(device=dev).manual_seed(999)
    centres = torch.randn((ncl, d), generator=gc, device=dev)
    centres /= centres.norm(dim=1, keepdim=True)

    def rows(m):
        x = torch.randn((m, d), generator=g, device=dev) * (sigma / d ** 0.5)
        x += centres[torch.randint(0, ncl, (m,), generator=g, device=dev)]
        return x / x.norm(dim=1, keepdim=True)
    return rows


def test_configs4_ivf_leg_list_sharded_at_the_per_gpu_shard_size(ssa, one_rank_group):
    """BASELINE configs[4], IVF leg, one GPU's share of it: IVF4096 over 12.5M x 4096 fp16 rows through
    ShardedIVFFlat on a one-rank RCCL group with the collectives forced on (broadcast centroids, routed add =
    assign -> all-to-all -> append, gathered search), nprobe 32, against a row-sharded Flat twin of the same
    rows.  Full-size properties: returned scores re-scored in float64 from the twin's stored rows, every IVF hit that
    beats the twin's k-th score is one of the twin's hits, recall@10 >= 0.98 (clustered rows: IVF's use case),
    batch 1 (per-query scan) and a batch that takes the list-major matrix-core scan, idempotence."""
    from semantic_search_amd.sharded import ShardedFlatIP, ShardedIVFFlat
    n, d, nlist, nprobe, k = 12_500_000, 4096, 4096, 32, 10
    dev = torch.device("cuda", 0)
    off = 5 * n                                            # rank 5's rows of the 100M-row database
    ivf = ShardedIVFFlat(d, nlist, storage="f16", device=0)
    flat = ShardedFlatIP(d, storage="f16", device=0)
    ivf._force_pipeline = flat._force_pipeline = True
    ivf.train(_clustered_device_rows(d, n, 99, dev)(256 * nlist // 4))     # 64 points per centroid keep the test short
    assert ivf.is_trained
    rows = _clustered_device_rows(d, n, 1234 + 5, dev)
    for s0 in range(0, n, 65536):
        m = min(65536, n - s0)
        x = rows(m)
        ivf.add_local(x, ids=np.arange(off + s0, off + s0 + m, dtype=np.int64))
        flat.add_local(x.half())
    del x
    flat.finalize(device=dev)
    flat.row_offset = off
    assert ivf.ntotal == n and ivf.local.ntotal == n and flat.ntotal == n
    sizes = ivf.local.list_sizes()
    assert int(sizes.sum()) == n and sizes.max() < 40 * n / nlist
    ivf.nprobe = nprobe
    Q = _clustered_device_rows(d, n, 4321, dev)(512)
    # batch 1
    nq1 = 24
    D1 = np.zeros((nq1, k), np.float32); I1 = np.zeros((nq1, k), np.int64)
    for qi in range(nq1):
        a, b = ivf.search(Q[qi:qi + 1], k)
        D1[qi], I1[qi] = a.cpu().numpy()[0], b.cpu().numpy()[0]
    t = ivf.local.last_timing()
    assert t["last_path"] == 2 and "ivf_scan" in ivf.local.last_kernel()
    Df, If = flat.search(Q, k)
    torch.cuda.synchronize()
    Df, If = Df.cpu().numpy(), If.cpu().numpy()
    Qh = Q.cpu().numpy()

    def check(D, I, sel, rtol, atol):
        assert (np.diff(D, axis=1) <= 0).all()
        valid = I >= 0
        assert ((I[valid] >= off) & (I[valid] < off + n)).all()
        for j, qi in enumerate(sel):
            v = valid[j]
            got = flat.local.reconstruct_batch(I[j][v] - off).astype(np.float64) @ Qh[qi].astype(np.float64)
            np.testing.assert_allclose(D[j][v], got, rtol=rtol, atol=atol)
            beats = I[j][v][D[j][v] > Df[qi, -1] * (1 + rtol) + atol]
            assert set(beats.tolist()) <= set(If[qi].tolist()), f"query {qi}: an IVF hit above the Flat k-th score is not a Flat hit"
        return float(np.mean([len(set(I[j].tolist()) & set(If[qi].tolist())) / k for j, qi in enumerate(sel)]))
    r1 = check(D1, I1, list(range(nq1)), 1e-5, 2e-6)
    assert r1 >= 0.98, r1
    # a batch on the list-major matrix-core scan (fp16 queries: the north-star tolerance)
    Db, Ib = ivf.search(Q, k)
    torch.cuda.synchronize()
    assert ivf.local.last_kernel() == "ivf_scan_lists_mfma_kernel"
    Db, Ib = Db.cpu().numpy(), Ib.cpu().numpy()
    sel = list(range(0, 512, 16))
    rb = check(Db[sel], Ib[sel], sel, 1e-3, 1e-4)
    assert rb >= 0.98, rb
    np.testing.assert_allclose(Db[:nq1], D1, rtol=1e-3, atol=1e-4)
    Db2, Ib2 = ivf.search(Q, k)
    torch.cuda.synchronize()
    assert np.array_equal(Ib2.cpu().numpy(), Ib) and np.array_equal(Db2.cpu().numpy(), Db)
    ivf.local.close(); flat.local.close()

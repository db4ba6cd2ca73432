"""Seeded sweep over the surfaces around the search kernels: the shard merge with ragged and empty shards, file round
trips of Flat and IVF indexes (full writes, appended Flat rows, IVF journal records), score floors, device tensors in
and out.  Each case checks a property that does not need a tolerance: the same index must answer the same way through
every route (bit-identical D and I).  Companion of tests/test_fuzz_gpu.py, which checks the kernels against the oracle."""
import numpy as np
import pytest

from conftest import SALT, unit_rows
from oracle import flat_ip

pytestmark = pytest.mark.gpu


def _index(ssa, kind, d, storage, nlist, seed):
    if kind == "flat":
        return ssa.IndexFlatIP(d, storage=storage)
    ix = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage=storage), d, nlist, ssa.METRIC_INNER_PRODUCT, storage=storage)
    ix.set_centroids(unit_rows(nlist, d, 400 + seed))
    return ix


@pytest.mark.parametrize("seed", range(40))
def test_shard_merge_sweep(ssa, seed):
    """Row shards of any sizes (empty ones, shards shorter than k) searched with their id offsets and merged by
    ssmi_topk_merge give exactly what one index over all rows gives (same kernels, same keys: bit-identical)."""
    import torch
    from semantic_search_amd.faiss_compat import topk_merge
    rng = np.random.default_rng(3000 + seed + SALT)
    d = int(rng.choice([64, 256, 1024, 4096]))
    n = int(rng.choice([5, 300, 2000, 5000]))
    world = int(rng.choice([1, 2, 3, 8]))
    nq = int(rng.choice([1, 3, 8, 20, 130]))
    k = int(rng.choice([1, 10, 64, 100, 300]))
    storage = str(rng.choice(["f32", "f16"]))
    if storage == "f32" and nq >= 9:
        storage = "f16"  # (the fp16 shadow path is covered by test_fuzz_gpu; here every route must be bit-identical)
    X, Q = unit_rows(n, d, 3100 + seed), unit_rows(nq, d, 3200 + seed)
    X[n // 2] = X[0]  # a tie across shard boundaries: the lower global id must come first
    cuts = np.sort(rng.integers(0, n + 1, world - 1)) if world > 1 else np.zeros(0, int)
    bounds = [0, *cuts.tolist(), n]
    Qd = torch.from_numpy(Q).cuda()
    whole = ssa.IndexFlatIP(d, storage=storage); whole.add(X)
    D0, I0 = whole.search(Qd, k)
    Ds, Is = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        shard = ssa.IndexFlatIP(d, storage=storage)
        if b > a:
            shard.add(X[a:b])
        D, I = shard.search(Qd, k, id_offset=a)
        assert ((I < 0) | ((I >= a) & (I < b))).all(), "shard returned an id outside its row block"
        Ds.append(D); Is.append(I)
    D, I = topk_merge(torch.stack(Ds), torch.stack(Is))
    case = f"seed={seed} n={n} d={d} world={world} bounds={bounds} nq={nq} k={k} {storage}"
    # a shard's K1/K2 slices differ from the whole index's, so fp32 sums may differ in the last bits: compare through the oracle's rule
    Xs = X.astype(np.float16).astype(np.float32) if storage == "f16" else X
    Qs = Q.astype(np.float16).astype(np.float32) if whole.last_timing()["last_path"] == 1 else Q
    ok, msg, _ = flat_ip.check_against(D.cpu().numpy(), I.cpu().numpy(), D0.cpu().numpy(), I0.cpu().numpy(), Xs, Qs, rtol=1e-5, tie_rtol=1e-5)
    assert ok, f"{case}: {msg}"
    assert ((I.cpu().numpy() >= 0).sum(1) == min(n, k)).all(), case


@pytest.mark.parametrize("seed", range(30))
def test_file_roundtrip_sweep(ssa, tmp_path, seed):
    """write_index -> read_index gives an index that answers bit-identically, for Flat and IVF, both storages, after
    incremental checkpoints (appended Flat rows, IVF journal records) as well as after a full write."""
    rng = np.random.default_rng(4000 + seed + SALT)
    kind = str(rng.choice(["flat", "ivf"]))
    d = int(rng.choice([24, 72, 256, 1024]))
    storage = str(rng.choice(["f32", "f16"]))
    nlist = int(rng.choice([1, 7, 40]))
    n0 = int(rng.choice([0, 1, 50, 700, 3000])) if kind == "flat" else int(rng.choice([nlist, 50, 700, 3000]))
    grows = [int(g) for g in rng.choice([1, 16, 16, 300], size=int(rng.integers(0, 4)))]
    nq, k = int(rng.choice([1, 4, 20])), int(rng.choice([1, 10, 100]))
    case = f"seed={seed} {kind} d={d} {storage} nlist={nlist} n0={n0} grows={grows} nq={nq} k={k}"
    X = unit_rows(n0 + sum(grows) + 1, d, 4100 + seed)
    Q = unit_rows(nq, d, 4200 + seed)
    ix = _index(ssa, kind, d, storage, nlist, seed)
    if kind == "ivf":
        ix.nprobe = int(rng.choice([1, 3, nlist]))
    if n0:
        ix.add(X[:n0])
    p = tmp_path / "vectors.faiss"
    ssa.write_index(ix, p)
    at = n0
    for g in grows:  # grow, then checkpoint incrementally where the format allows it
        ix.add(X[at:at + g]); at += g
        ssa.write_index(ix, p, incremental=bool(rng.random() < 0.7))
    back = ssa.read_index(p, storage=storage)
    assert type(back).__name__ == type(ix).__name__ and back.ntotal == ix.ntotal == at and back.d == d, case
    if kind == "ivf":
        assert back.nlist == nlist and back.nprobe == ix.nprobe, case
        np.testing.assert_array_equal(back.list_sizes(), ix.list_sizes(), err_msg=case)
    if at:
        D0, I0 = ix.search(Q, k); D1, I1 = back.search(Q, k)
        np.testing.assert_array_equal(I0, I1, err_msg=case)
        np.testing.assert_array_equal(D0, D1, err_msg=case)
    # a full write of the reloaded index is the same file as a full write of the original
    q1, q2 = tmp_path / "a.faiss", tmp_path / "b.faiss"
    ssa.write_index(ix, q1); ssa.write_index(back, q2)
    assert q1.read_bytes() == q2.read_bytes(), case


@pytest.mark.parametrize("seed", range(30))
def test_floor_and_device_io_sweep(ssa, seed):
    """min_score cuts a row exactly where the unfloored row drops below it, on every path; device tensors in and out
    (fresh or caller-provided outputs) give the bits the numpy route gives."""
    import torch
    rng = np.random.default_rng(5000 + seed + SALT)
    kind = str(rng.choice(["flat", "ivf"]))
    d = int(rng.choice([64, 100, 512, 4096]))
    storage = str(rng.choice(["f32", "f16"]))
    nlist = int(rng.choice([4, 32]))
    n = int(rng.choice([40, 600, 4000]))
    nq = int(rng.choice([1, 2, 5, 9, 33, 140, 260]))
    k = int(rng.choice([1, 10, 70, 256]))
    case = f"seed={seed} {kind} d={d} {storage} n={n} nq={nq} k={k}"
    X, Q = unit_rows(n, d, 5100 + seed), unit_rows(nq, d, 5200 + seed)
    ix = _index(ssa, kind, d, storage, nlist, seed)
    if kind == "ivf":
        ix.nprobe = int(rng.choice([1, 3, nlist]))
    ix.add(X)
    D0, I0 = ix.search(Q, k)
    valid = D0[I0 >= 0]
    floors = [-1.0, 2.0] + ([float(np.median(valid)), float(valid.max()), float(valid.min())] if valid.size else [])
    for floor in floors:
        D, I = ix.search(Q, k, min_score=floor)
        keep = (I0 >= 0) & (D0 >= np.float32(floor))
        np.testing.assert_array_equal(I, np.where(keep, I0, -1), err_msg=f"{case} floor={floor}")
        np.testing.assert_array_equal(D, np.where(keep, D0, flat_ip.NEG_MAX), err_msg=f"{case} floor={floor}")
    Qd = torch.from_numpy(Q).cuda()
    Dd, Id = ix.search(Qd, k)
    np.testing.assert_array_equal(Id.cpu().numpy(), I0, err_msg=case)
    np.testing.assert_array_equal(Dd.cpu().numpy(), D0, err_msg=case)
    Do = torch.full((nq, k), 7.0, dtype=torch.float32, device="cuda"); Io = torch.full((nq, k), 7, dtype=torch.int64, device="cuda")
    off = int(rng.choice([0, 12345, 3_000_000_000]))
    ix.search(Qd, k, id_offset=off, out=(Do, Io)) if kind == "flat" else ix.search(Qd, k, out=(Do, Io))
    exp = np.where(I0 >= 0, I0 + (off if kind == "flat" else 0), -1)
    np.testing.assert_array_equal(Io.cpu().numpy(), exp, err_msg=case)
    np.testing.assert_array_equal(Do.cpu().numpy(), D0, err_msg=case)
    ix.close()


@pytest.mark.parametrize("seed", range(20))
def test_readback_sweep(ssa, seed):
    """reconstruct / reconstruct_n / reconstruct_batch(_device) / score_ids on Flat and IVF (with a direct map) give the
    stored rows (fp16-rounded under f16 storage) and their inner products; ids out of range are an error, not a read."""
    rng = np.random.default_rng(7000 + seed + SALT)
    kind = str(rng.choice(["flat", "ivf"]))
    d = int(rng.choice([8, 100, 256, 4096]))
    storage = str(rng.choice(["f32", "f16"]))
    n = int(rng.choice([1, 33, 700, 5000]))
    X = unit_rows(n, d, 7100 + seed)
    Xs = X.astype(np.float16).astype(np.float32) if storage == "f16" else X
    ix = _index(ssa, kind, d, storage, int(rng.choice([1, 5, 30])), seed)
    if kind == "ivf":
        with pytest.raises(RuntimeError):
            ix.add(X[:1]); ix.reconstruct(0)          # no direct map yet (FAISS raises too)
        ix.make_direct_map()
        ix.add(X[1:])
    else:
        ix.add(X)
    case = f"seed={seed} {kind} d={d} {storage} n={n}"
    ids = rng.integers(0, n, size=int(rng.choice([1, 7, 300])))
    np.testing.assert_array_equal(ix.reconstruct_batch(ids), Xs[ids], err_msg=case)
    np.testing.assert_array_equal(ix.reconstruct_batch_device(ids).cpu().numpy(), Xs[ids], err_msg=case)
    np.testing.assert_array_equal(ix.reconstruct(int(ids[0])), Xs[ids[0]], err_msg=case)
    if kind == "flat":
        a = int(rng.integers(0, n)); b = int(rng.integers(a, n + 1))
        np.testing.assert_array_equal(ix.reconstruct_n(a, b - a), Xs[a:b], err_msg=case)
    neg = unit_rows(int(rng.choice([1, 3])), d, 7200 + seed)
    np.testing.assert_allclose(ix.score_ids(ids, neg), Xs[ids] @ neg.T, rtol=1e-5, atol=1e-6, err_msg=case)
    for bad in ([-1], [n], [0, n + 5]):
        with pytest.raises(RuntimeError):
            ix.reconstruct_batch(bad)
        with pytest.raises(RuntimeError):
            ix.score_ids(bad, neg)
    ix.close()


def test_threads_with_their_own_indexes_streams_and_lifetimes(ssa):
    """ComfyUI calls the engine from several threads, each on its own index (core/index_manager.py:1481-1504): eight
    threads create, grow, search (numpy route, or device tensors on a stream of their own) and free indexes of their own
    at the same time; every search is checked against numpy on the thread's own rows."""
    import threading
    import torch
    errors = []

    def worker(t):
        try:
            rng = np.random.default_rng(100 + t + SALT)
            stream = torch.cuda.Stream() if t % 2 else None
            for rep in range(6):
                d = int(rng.choice([64, 256, 1024]))
                storage = "f32" if t % 4 < 2 else "f16"
                X = unit_rows(int(rng.choice([50, 700, 3000])), d, 1000 * t + rep)
                Xs = X.astype(np.float16).astype(np.float32) if storage == "f16" else X
                ix = ssa.IndexFlatIP(d, storage=storage)
                at = 0
                while at < len(X):
                    step = int(rng.choice([16, 200, 1000]))
                    ix.add(X[at:at + step]); at = min(len(X), at + step)
                    Q = unit_rows(int(rng.choice([1, 3])), d, 7 * t + at)
                    if stream is None:
                        D, I = ix.search(Q, 5)
                    else:
                        with torch.cuda.stream(stream):
                            Dd, Id = ix.search(torch.from_numpy(Q).cuda(), 5)
                        stream.synchronize()
                        D, I = Dd.cpu().numpy(), Id.cpu().numpy()
                    S = Q @ Xs[:at].T
                    for qi in range(len(Q)):
                        kk = min(5, at)
                        exp = np.sort(S[qi])[::-1][:kk]
                        np.testing.assert_allclose(D[qi][:kk], exp, rtol=1e-5, atol=1e-6)
                        np.testing.assert_allclose(S[qi][I[qi][:kk]], exp, rtol=1e-5, atol=1e-6)
                ix.close()
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors

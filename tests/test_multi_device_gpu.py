"""One index over several GPUs inside ONE process (semantic_search_amd/multi_device.py; SURVEY.md 8e process model),
exercised with LOGICAL shards on GPU 0 (`devices=[0, 0, 0]`): every search must equal the unsharded index bit for bit
(same kernels, same merge), whatever the routing of the adds - ragged and empty shards, segments, score floors,
device tensors, read-back by global id, ONE FAISS file in and out, and the reference's manager surface on top
(core/index_manager.py:244-398, 1288-1448).  The multi-GPU leg itself (peer copies over xGMI) is unmeasured: this
repository has one GPU."""
import numpy as np
import pytest
import torch

from conftest import unit_rows

pytestmark = pytest.mark.gpu


def same(a, b):
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[0], b[0])


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0, 0, 0]])
@pytest.mark.parametrize("storage", ["f32", "f16", "bf16"])
def test_flat_over_logical_shards_equals_one_index(ssa, devices, storage):
    d = 256
    X = unit_rows(9000, d, 1)
    multi = ssa.IndexFlatIP(d, storage=storage, devices=devices)
    assert type(multi).__name__ == "MultiDeviceFlatIP" and multi.nshards == len(devices) and multi.is_trained
    one = ssa.IndexFlatIP(d, storage=storage)
    D0, I0 = multi.search(X[:3], 5)
    assert (I0 == -1).all()                                         # empty index, empty shards
    cuts = [0, 1, 17, 18, 2000, 2016, 5000, 5001, 9000]              # the reference adds 16 at a time; also 1 and thousands
    for a, b in zip(cuts[:-1], cuts[1:]):
        multi.add(X[a:b]); one.add(X[a:b])
    assert multi.ntotal == one.ntotal == 9000
    segs = multi.segments()
    assert sum(s[1] for s in segs) == 9000 and [s[0] for s in segs] == sorted(s[0] for s in segs)
    if len(devices) > 1:
        assert len({s[2] for s in segs}) == len(devices)            # least-full routing reaches every shard
    Q = unit_rows(40, d, 2)
    for nq, k in ((1, 10), (3, 100), (40, 10), (40, 600)):
        same(multi.search(Q[:nq], k), one.search(Q[:nq], k))
    Dq, Iq = one.search(Q[:5], 10)
    floor = float((np.float64(Dq[0, 3]) + np.float64(Dq[0, 4])) / 2)
    same(multi.search(Q[:5], 10, min_score=floor), one.search(Q[:5], 10, min_score=floor))
    # device tensors in -> device tensors out, on the first device
    Qd = torch.from_numpy(Q).cuda()
    Dd, Id = multi.search(Qd, 10)
    torch.cuda.synchronize()
    same((Dd.cpu().numpy(), Id.cpu().numpy()), one.search(Q, 10))
    # read back by GLOBAL id
    ids = np.array([0, 16, 17, 18, 1999, 2000, 5000, 8999])
    np.testing.assert_array_equal(multi.reconstruct_batch(ids), one.reconstruct_batch(ids))
    np.testing.assert_array_equal(multi.reconstruct(5000), one.reconstruct(5000))
    np.testing.assert_array_equal(multi.reconstruct_batch_device(ids).cpu().numpy(), one.reconstruct_batch(ids))
    np.testing.assert_array_equal(multi.score_ids(ids, Q[:3]), one.score_ids(ids, Q[:3]))
    with pytest.raises(RuntimeError):
        multi.reconstruct(9000)
    multi.close(); one.close()


def test_flat_file_round_trip_is_the_single_index_file(ssa, tmp_path):
    d = 64
    X = unit_rows(3000, d, 3)
    multi = ssa.IndexFlatIP(d, devices=[0, 0, 0])
    one = ssa.IndexFlatIP(d)
    for s in range(0, 3000, 700):
        multi.add(X[s:s + 700]); one.add(X[s:s + 700])
    pm, po = tmp_path / "m.faiss", tmp_path / "o.faiss"
    ssa.write_index(multi, pm); ssa.write_index(one, po)
    assert pm.read_bytes() == po.read_bytes()                       # ONE FAISS IxFI file, rows in global-id order
    back = ssa.read_index(pm, devices=[0, 0])
    assert back.ntotal == 3000 and back.nshards == 2
    Q = unit_rows(7, d, 4)
    same(back.search(Q, 10), one.search(Q, 10))
    same(ssa.read_index(pm).search(Q, 10), one.search(Q, 10))       # and any single index reads it
    with pytest.raises(RuntimeError):
        (tmp_path / "bad.faiss").write_bytes(pm.read_bytes()[:5000])
        ssa.read_index(tmp_path / "bad.faiss", devices=[0, 0])


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_ivf_lists_over_logical_shards_equal_one_index(ssa, devices, tmp_path):
    d, nlist = 128, 13
    X = unit_rows(6000, d, 5)
    multi = ssa.IndexIVFFlat(None, d, nlist, ssa.METRIC_INNER_PRODUCT, devices=devices)
    assert "IVFFlat" in type(multi).__name__ and not multi.is_trained
    with pytest.raises(RuntimeError):
        multi.add(X[:4])
    multi.train(X[:2000])
    assert multi.is_trained
    one = ssa.IndexIVFFlat(ssa.IndexFlatIP(d), d, nlist, ssa.METRIC_INNER_PRODUCT)
    one.set_centroids(multi.get_centroids())
    for a, b in ((0, 16), (16, 17), (17, 4000), (4000, 6000)):
        multi.add(X[a:b]); one.add(X[a:b])
    assert multi.ntotal == 6000
    np.testing.assert_array_equal(multi.list_sizes(), one.list_sizes())
    for s, sh in enumerate(multi.shards):                          # list l lives wholly on shard l % S
        ls = sh.list_sizes()
        assert all(ls[l] == 0 for l in range(nlist) if l % len(devices) != s)
    Q = unit_rows(30, d, 6)
    for nprobe in (1, 4, nlist):
        multi.nprobe = one.nprobe = nprobe
        for nq in (1, 30):
            same(multi.search(Q[:nq], 10), one.search(Q[:nq], 10))
    with pytest.raises(RuntimeError):
        multi.reconstruct(0)                                        # like FAISS without a direct map (index_factory.py:275)
    # ONE FAISS IwFl file from the lists the shards own; any index reads it back
    pm = tmp_path / "m.faiss"
    multi.nprobe = one.nprobe = 3
    ssa.write_index(multi, pm)
    a = ssa.read_index(pm)
    b = ssa.read_index(pm, devices=[0, 0, 0, 0])
    assert a.ntotal == b.ntotal == 6000 and a.nprobe == b.nprobe == 3 and b.nshards == 4
    same(a.search(Q, 10), one.search(Q, 10))
    same(b.search(Q, 10), one.search(Q, 10))
    for ix in (multi, one, a, b):
        ix.close()


def test_manager_surface_over_several_devices(ssa, tmp_path):
    """SemanticIndex(..., devices=[...]): the reference's add / search / save / reload / delete / compact / rebuild
    flow (core/index_manager.py:244-398, 1288-1448) over a sharded engine gives what a single-device index gives."""
    from semantic_search_amd.index_manager import SemanticIndex
    d = 64
    X = unit_rows(1500, d, 7)

    def build(name, **kw):
        ix = SemanticIndex(name, d, "flat", base_path=tmp_path, check_files_exist=False, **kw)
        for s in range(0, 1500, 16):
            m = min(16, 1500 - s)
            ix.add_vectors(X[s:s + m], [f"/img/{i}.jpg" for i in range(s, s + m)], [""] * m, [(0.0, 0)] * m)
        return ix
    multi, one = build("multi", devices=[0, 0, 0]), build("one")
    assert type(multi._faiss_index).__name__ == "MultiDeviceFlatIP"
    q = unit_rows(1, d, 8)[0]
    ids = lambda r: [(x.vector_id, x.file_path) for x in r.results]
    assert ids(multi.search(q, 20)) == ids(one.search(q, 20))
    assert [ids(r) for r in multi.search_batch(X[:12], 5)] == [ids(r) for r in one.search_batch(X[:12], 5)]
    multi.save(); multi.close()
    multi = SemanticIndex("multi", d, "flat", base_path=tmp_path, check_files_exist=False, devices=[0, 0])
    assert multi._faiss_index.ntotal == 1500 and multi._faiss_index.nshards == 2
    assert ids(multi.search(q, 20)) == ids(one.search(q, 20))
    assert ids(multi.search_with_exclusion(q, [X[3]], 10)) == ids(one.search_with_exclusion(q, [X[3]], 10))
    for ix in (multi, one):
        for i in range(0, 1500, 3):
            assert ix.db.remove_image(f"/img/{i}.jpg")
        assert ix.needs_compaction()[0]
        assert ix.compact()["success"]
    assert multi._faiss_index.ntotal == one._faiss_index.ntotal == 1000
    assert type(multi._faiss_index).__name__ == "MultiDeviceFlatIP"
    assert ids(multi.search(q, 20)) == ids(one.search(q, 20))
    multi.close(); one.close()


# ---- seeded walks (SSMI_FUZZ_SALT gives other cases): one index over 1..5 logical shards against the same index on one
# ---- handle, through ragged adds, searches of every batch class, score floors, read-backs and file round trips into other
# ---- shard counts
from conftest import SALT


@pytest.mark.parametrize("seed", range(12))
def test_flat_walk_over_logical_shards(ssa, tmp_path, seed):
    rng = np.random.default_rng(4000 + seed + SALT)
    d = int(rng.choice([64, 100, 256, 1024, 4096]))
    storage = str(rng.choice(["f32", "f16", "bf16"]))
    nsh = int(rng.integers(1, 6))
    pool = unit_rows(7000 if d <= 1024 else 2500, d, 4100 + seed)
    multi = ssa.IndexFlatIP(d, storage=storage, devices=[0] * nsh)
    one = ssa.IndexFlatIP(d, storage=storage)
    at = 0
    for step in range(10):
        grow = min(int(rng.choice([0, 1, 16, 16, 250, 3000])), len(pool) - at)
        if grow:
            multi.add(pool[at:at + grow]); one.add(pool[at:at + grow]); at += grow
        assert multi.ntotal == one.ntotal == at
        nq = int(rng.choice([1, 3, 9, 70, 300]))
        k = int(rng.choice([1, 10, 100, 300]))
        Q = unit_rows(nq, d, 4200 + 31 * seed + step)
        case = f"seed={seed} step={step} d={d} {storage} shards={nsh} n={at} nq={nq} k={k}"
        plain = one.search(Q, k)
        if rng.random() < 0.3 and at > k:
            floor = float(plain[0][0, min(k, at) // 2])
            a, b = multi.search(Q, k, min_score=floor), one.search(Q, k, min_score=floor)
        else:
            a, b = multi.search(Q, k), plain
        np.testing.assert_array_equal(a[1], b[1], err_msg=case)
        np.testing.assert_array_equal(a[0], b[0], err_msg=case)
        if at:
            ids = rng.integers(0, at, 9)
            np.testing.assert_array_equal(multi.reconstruct_batch(ids), one.reconstruct_batch(ids), err_msg=case)
        if step in (4, 9) and at:
            p = tmp_path / f"w{seed}_{step}.faiss"
            ssa.write_index(multi, p)
            back = ssa.read_index(p, storage=storage, devices=[0] * int(rng.integers(1, 5)))
            assert back.ntotal == at
            c = back.search(Q, k)
            np.testing.assert_array_equal(c[1], plain[1], err_msg=case)
            back.close()
    multi.close(); one.close()


@pytest.mark.parametrize("seed", range(8))
def test_ivf_walk_over_logical_shards(ssa, tmp_path, monkeypatch, seed):
    from oracle import flat_ip
    rng = np.random.default_rng(4500 + seed + SALT)
    d = int(rng.choice([64, 256, 512, 1024]))
    nlist = int(rng.choice([3, 16, 50]))
    storage = str(rng.choice(["f32", "f16"]))
    nsh = int(rng.integers(1, 5))
    monkeypatch.setenv("SSMI_NO_SHADOW", "1")           # exact scans on both sides: the two forms may pick different
    monkeypatch.setenv("SSMI_IVF_LIST_MFMA", "0")       # scan kernels for a batch (a shard holds fewer lists), which differ in summation order only
    cen = unit_rows(nlist, d, 4600 + seed)
    pool = cen[rng.integers(0, nlist, 4000)] + 0.6 * unit_rows(4000, d, 4700 + seed)
    pool = (pool / np.linalg.norm(pool, axis=1, keepdims=True)).astype(np.float32)
    multi = ssa.IndexIVFFlat(None, d, nlist, ssa.METRIC_INNER_PRODUCT, storage=storage, devices=[0] * nsh)
    one = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage=storage), d, nlist, ssa.METRIC_INNER_PRODUCT, storage=storage)
    multi.set_centroids(cen); one.set_centroids(cen)
    Xs = pool.astype(np.float16).astype(np.float32) if storage == "f16" else pool
    at = 0
    for step in range(8):
        grow = min(int(rng.choice([1, 16, 16, 40, 400, 1500])), len(pool) - at)
        if grow:
            multi.add(pool[at:at + grow]); one.add(pool[at:at + grow]); at += grow
        assert multi.ntotal == one.ntotal == at
        np.testing.assert_array_equal(multi.list_sizes(), one.list_sizes())
        nq = int(rng.choice([1, 3, 9, 40, 130]))
        k = int(rng.choice([1, 10, 100]))
        multi.nprobe = one.nprobe = int(rng.choice([1, 3, nlist]))
        Q = unit_rows(nq, d, 4800 + 31 * seed + step)
        case = f"seed={seed} step={step} d={d} nlist={nlist} {storage} shards={nsh} n={at} nq={nq} k={k} nprobe={one.nprobe}"
        (D, I), (Dr, Ir) = multi.search(Q, k), one.search(Q, k)
        ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xs[:at], Q, rtol=1e-5, tie_rtol=1e-5, atol=2e-6)
        assert ok, f"{case}: {msg}"
    p = tmp_path / f"ivfw{seed}.faiss"
    ssa.write_index(multi, p)
    back = ssa.read_index(p, storage=storage, devices=[0] * int(rng.integers(1, 5)))
    assert back.ntotal == at and back.nprobe == one.nprobe
    D, I = back.search(Q, k)
    ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xs[:at], Q, rtol=1e-5, tie_rtol=1e-5, atol=2e-6)
    assert ok, msg
    for ix in (multi, one, back):
        ix.close()

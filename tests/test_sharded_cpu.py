"""world_size-2 gloo test of the row-sharded search plumbing (offset exchange,
all-gather layout, merge call) on CPU.  The per-rank search and the merge are
injected test doubles built on the oracle; the HIP versions are covered by
tests/test_flat_gpu.py::test_topk_merge_of_shards_equals_unsharded."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import unit_rows


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleLocalIndex:
    def __init__(self, d):
        self.d, self.X = d, np.zeros((0, d), np.float32)

    @property
    def ntotal(self):
        return len(self.X)

    def add(self, x):
        self.X = np.concatenate([self.X, np.asarray(x, np.float32)])

    def search(self, q, k, id_offset=0, out=None):
        from oracle import flat_ip
        D, I = flat_ip.search(self.X, q.numpy(), k)
        I = np.where(I >= 0, I + id_offset, -1)
        if out is not None:
            out[0].copy_(torch.from_numpy(D)); out[1].copy_(torch.from_numpy(I))
            return out
        return torch.from_numpy(D), torch.from_numpy(I)


def oracle_merge(Dg, Ig):
    from oracle import flat_ip
    w, nq, k = Dg.shape
    assert Dg.dtype == torch.float32 and Ig.dtype == torch.int64
    D = Dg.permute(1, 0, 2).reshape(nq, w * k).numpy()
    I = Ig.permute(1, 0, 2).reshape(nq, w * k).numpy()
    outD = np.full((nq, k), flat_ip.NEG_MAX, np.float32); outI = np.full((nq, k), -1, np.int64)
    for qi in range(nq):
        v = I[qi] >= 0
        order = np.lexsort((I[qi][v], -D[qi][v].astype(np.float64)))[:k]
        outD[qi, :len(order)] = D[qi][v][order]; outI[qi, :len(order)] = I[qi][v][order]
    return torch.from_numpy(outD), torch.from_numpy(outI)


def _worker(rank, world, port, n, d, k, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from semantic_search_amd.sharded import ShardedFlatIP, shard_bounds
    X = unit_rows(n, d, 21)
    Q = torch.from_numpy(unit_rows(3, d, 22))
    b = shard_bounds(n, world)
    idx = ShardedFlatIP(d, _local_index=OracleLocalIndex(d), _merge=oracle_merge)
    idx.add_local(X[b[rank]:b[rank + 1]])
    idx.finalize()
    assert idx.ntotal == n and idx.row_offset == b[rank]
    D, I = idx.search(Q, k)
    if rank == 0:
        np.save(out + "_D.npy", D.numpy()); np.save(out + "_I.npy", I.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_search_equals_unsharded(tmp_path):
    from oracle import flat_ip
    n, d, k = 1001, 64, 12
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(2, _free_port(), n, d, k, out), nprocs=2, join=True)
    X, Q = unit_rows(n, d, 21), unit_rows(3, d, 22)
    Dr, Ir = flat_ip.search(X, Q, k)
    ok, msg, _ = flat_ip.check_against(np.load(out + "_D.npy"), np.load(out + "_I.npy"), Dr, Ir, X, Q, 1e-5, 1e-5)
    assert ok, msg


def test_shard_bounds():
    from semantic_search_amd.sharded import shard_bounds
    assert shard_bounds(10, 3) == [0, 4, 7, 10]
    assert shard_bounds(2, 4) == [0, 1, 2, 2, 2]
    assert shard_bounds(1000000, 8)[-1] == 1000000


# --------------------------------------------------------------------------- list-sharded IVF
class OracleLocalIVF:
    """Oracle-backed stand-in for faiss_compat.IndexIVFFlat (the surface ShardedIVFFlat drives)."""

    def __init__(self, d, nlist):
        self.d, self.nlist, self.nprobe, self.C = d, nlist, 1, None
        self.X = np.zeros((0, d), np.float32)
        self.labels = np.zeros((0,), np.int64)
        self.ids = np.zeros((0,), np.int64)
        self.assigned = 0  # vectors this rank ran through the coarse quantizer

    @property
    def is_trained(self):
        return self.C is not None

    @property
    def ntotal(self):
        return len(self.X)

    def train(self, x):
        from oracle import ivf_flat
        self.C, _ = ivf_flat.kmeans(np.asarray(x, np.float32), self.nlist, niter=4, seed=3)

    def get_centroids(self):
        return self.C.copy()

    def set_centroids(self, c):
        self.C = np.asarray(c, np.float32).copy()

    def assign(self, x):
        from oracle import ivf_flat
        x = x.numpy() if isinstance(x, torch.Tensor) else np.asarray(x, np.float32)
        self.assigned += len(x)
        return ivf_flat.assign(x, self.C).astype(np.int32)

    def add_preassigned(self, x, labels, ids):
        x = x.numpy() if isinstance(x, torch.Tensor) else np.asarray(x, np.float32)
        self.X = np.concatenate([self.X, x.astype(np.float32)])
        self.labels = np.concatenate([self.labels, np.asarray(labels, np.int64)])
        self.ids = np.concatenate([self.ids, np.asarray(ids, np.int64)])

    def search(self, q, k, id_offset=0, out=None):
        from oracle import flat_ip, ivf_flat
        q = q.numpy()
        _, probes = flat_ip.search(self.C, q, min(self.nprobe, self.nlist))
        D = np.full((len(q), k), flat_ip.NEG_MAX, np.float32); I = np.full((len(q), k), -1, np.int64)
        for qi in range(len(q)):
            sel = np.nonzero(np.isin(self.labels, probes[qi]))[0]
            if len(sel):
                d, i = flat_ip.topk_total_order(q[qi:qi + 1] @ self.X[sel].T, k, self.ids[sel])
                D[qi], I[qi] = d[0], i[0]
        if out is not None:
            out[0].copy_(torch.from_numpy(D)); out[1].copy_(torch.from_numpy(I))
            return out
        return torch.from_numpy(D), torch.from_numpy(I)


def _ivf_worker(rank, world, port, n, d, nlist, nprobe, k, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from semantic_search_amd.sharded import ShardedIVFFlat, shard_bounds
    X = unit_rows(n, d, 41)
    Q = torch.from_numpy(unit_rows(4, d, 42))
    idx = ShardedIVFFlat(d, nlist, _local_index=OracleLocalIVF(d, nlist), _merge=oracle_merge)
    idx.train(X[:600] if rank == 0 else None)              # rank 0 trains, centroids are broadcast
    idx.nprobe = nprobe
    # first half through add() (every rank passes the same vectors), second half through add_local()
    # (each rank passes only its own rows; rank 1 passes more than rank 0)
    half = n // 2
    idx.add(X[:half])
    cut = half + (n - half) // 3
    idx.add_local(X[half:cut] if rank == 0 else X[cut:])
    assert idx.ntotal == n
    loc = idx.local
    assert (loc.labels % world == rank).all()               # a rank holds only the lists it owns
    b = shard_bounds(half, world)
    expect_assigned = (b[rank + 1] - b[rank]) + ((cut - half) if rank == 0 else (n - cut))
    assert loc.assigned == expect_assigned                  # every vector was assigned once, on one rank
    D, I = idx.search(Q, k)
    np.save(out + "_members_%d.npy" % rank, np.stack([loc.ids, loc.labels]))
    if rank == 0:
        np.save(out + "_C.npy", loc.C); np.save(out + "_D.npy", D.numpy()); np.save(out + "_I.npy", I.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_list_sharded_ivf_equals_unsharded(tmp_path):
    """world_size-2 gloo run of ShardedIVFFlat: broadcast centroids, routed add (assign a slice, all-to-all
    to the list owners), search = local scans + one all-gather + merge; equals the unsharded oracle IVF."""
    from oracle import flat_ip, ivf_flat
    n, d, nlist, nprobe, k = 2000, 48, 16, 5, 10
    out = str(tmp_path / "ivf")
    mp.spawn(_ivf_worker, args=(2, _free_port(), n, d, nlist, nprobe, k, out), nprocs=2, join=True)
    X, Q = unit_rows(n, d, 41), unit_rows(4, d, 42)
    C = np.load(out + "_C.npy")
    members = np.concatenate([np.load(out + "_members_%d.npy" % r) for r in range(2)], axis=1)
    ids, labels = members
    assert sorted(ids.tolist()) == list(range(n))           # every vector stored exactly once, ids are global row numbers
    lab_by_id = np.empty(n, np.int64); lab_by_id[ids] = labels
    np.testing.assert_array_equal(lab_by_id, ivf_flat.assign(X, C))
    Dr, Ir, _ = ivf_flat.search(X, C, lab_by_id, Q, k, nprobe)
    np.testing.assert_array_equal(np.load(out + "_I.npy"), Ir)
    np.testing.assert_allclose(np.load(out + "_D.npy"), Dr, rtol=1e-6)

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

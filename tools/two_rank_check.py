"""Two REAL ranks (the HIP indexes, not test doubles) of the sharded classes on ONE GPU, gloo between them:
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/two_rank_check.py
Row-sharded Flat (fp32 and fp16 rows; batch 1, 5, 300) and list-sharded IVF (rank 0 trains, centroids broadcast,
reference-shaped add and routed add_local) against one unsharded index over the same vectors on rank 0.  The pytest
suite covers world_size 2 with oracle doubles on CPU (tests/test_sharded_cpu.py) and the real path on a one-rank RCCL
group (tests/test_sharded_gpu.py); a pytest process that has initialised the GPU must not start child processes on this
pool, so this check is a tool run from a launcher that never touches the GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import unit_rows
from oracle import flat_ip, ivf_flat
import semantic_search_amd as ssa
from semantic_search_amd.sharded import ShardedFlatIP, ShardedIVFFlat, shard_bounds


def r16(a):
    return a.astype(np.float16).astype(np.float32)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    fails = 0
    # ---- Flat, row-sharded
    for storage, n, d in (("f32", 5003, 256), ("f16", 9001, 1024), ("f32", 3, 64)):
        X = unit_rows(n, d, 41)
        b = shard_bounds(n, world)
        sh = ShardedFlatIP(d, storage=storage, device=0)
        if b[rank + 1] > b[rank]:
            sh.add_local(X[b[rank]:b[rank + 1]])
        sh.finalize()
        assert sh.ntotal == n and sh.row_offset == b[rank], (sh.ntotal, sh.row_offset)
        for B, k in ((1, 10), (5, 100), (300, 10), (2, 2000 if False else 700)):
            Q = unit_rows(B, d, 42 + B)
            D, I = sh.search(torch.from_numpy(Q).to(dev), k)
            D, I = D.cpu().numpy(), I.cpu().numpy()
            if rank == 0:
                whole = ssa.IndexFlatIP(d, storage=storage, device=0); whole.add(X)
                D0, I0 = whole.search(Q, k)
                mf = whole.last_timing()["last_path"] == 1
                Xs = r16(X) if (storage == "f16" or mf) else X
                Qs = r16(Q) if mf else Q
                tol = 1e-5 if not (mf and storage == "f32") else 1e-3
                ok, msg, _ = flat_ip.check_against(D, I, D0, I0, Xs, Qs, rtol=tol, tie_rtol=tol, atol=2e-6 if tol < 1e-4 else 6 * 2.0 ** -11 * np.sqrt(2.0 / d))
                print(f"flat {storage} n={n} d={d} B={B} k={k}: {'ok' if ok else 'MISMATCH ' + msg}", flush=True)
                fails += not ok
    # ---- IVF, list-sharded
    for storage, n, d, nlist in (("f32", 6000, 128, 16), ("f16", 8000, 512, 40)):
        X = unit_rows(n, d, 51)
        Xtrain = unit_rows(3000, d, 52)
        sh = ShardedIVFFlat(d, nlist, storage=storage, device=0)
        sh.train(Xtrain)
        C = sh.local.get_centroids()
        Call = [torch.zeros_like(torch.from_numpy(C)) for _ in range(world)]
        dist.all_gather(Call, torch.from_numpy(C))
        assert all(torch.equal(c, Call[0]) for c in Call), "centroids differ between ranks"
        half = n // 2
        sh.add(X[:half])                                   # reference-shaped: every rank passes the same vectors
        b = shard_bounds(n - half, world)
        sh.add_local(X[half + b[rank]:half + b[rank + 1]])  # routed: every rank passes its own
        assert sh.ntotal == n, sh.ntotal
        sizes = torch.from_numpy(sh.local.list_sizes().astype(np.int64))
        tot = sizes.clone(); dist.all_reduce(tot)
        own = np.arange(nlist) % world == rank
        assert (sizes.numpy()[~own] == 0).all(), "a rank holds members of a list it does not own"
        for nprobe in (1, 4, nlist):
            sh.nprobe = nprobe
            for B, k in ((1, 10), (6, 50), (40, 10)):
                Q = unit_rows(B, d, 60 + B)
                D, I = sh.search(torch.from_numpy(Q).to(dev), k)
                D, I = D.cpu().numpy(), I.cpu().numpy()
                if rank == 0:
                    whole = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage=storage, device=0), d, nlist, ssa.METRIC_INNER_PRODUCT, storage=storage, device=0)
                    whole.set_centroids(C); whole.add(X); whole.nprobe = nprobe
                    assert int(tot.sum()) == n and (tot.numpy() == whole.list_sizes()).all(), "list sizes differ from the unsharded index"
                    D0, I0 = whole.search(Q, k)
                    Xs = r16(X) if storage == "f16" else X
                    mf = "mfma" in whole.last_kernel()
                    Qs = r16(Q) if mf else Q
                    ok, msg, _ = flat_ip.check_against(D, I, D0, I0, Xs, Qs, rtol=1e-5 if not mf else 1e-3, tie_rtol=1e-5 if not mf else 1e-3, atol=2e-6 if not mf else 1e-4)
                    print(f"ivf {storage} n={n} d={d} nlist={nlist} nprobe={nprobe} B={B} k={k}: {'ok' if ok else 'MISMATCH ' + msg}", flush=True)
                    fails += not ok
    t = torch.tensor([fails]); dist.all_reduce(t)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("two-rank check:", "PASS" if int(t) == 0 else f"{int(t)} FAILURES", flush=True)
    sys.exit(1 if int(t) else 0)


if __name__ == "__main__":
    main()

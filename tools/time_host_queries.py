"""Batch-1 search latency with host (numpy) queries and results against device-resident ones:
time_host_queries.py n d storage [ivf_nlist]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import semantic_search_amd as ssa
n, d, storage = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
nlist = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if nlist:
    index = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage=storage), d, nlist, ssa.METRIC_INNER_PRODUCT, storage=storage)
    g0 = torch.Generator(device="cuda:0").manual_seed(3)
    xt = torch.randn((64 * nlist, d), generator=g0, device="cuda:0"); xt /= xt.norm(dim=1, keepdim=True)
    index.train(xt); del xt
    index.nprobe = 32
if not nlist:
    index = ssa.IndexFlatIP(d, storage=storage)
    index.add_random_unit(n, seed=1)
if nlist:
    index.add_random_unit(n, seed=1)
g = torch.Generator(device="cuda:0").manual_seed(2)
Q = torch.randn((64, d), generator=g, device="cuda:0"); Q /= Q.norm(dim=1, keepdim=True)
Qh = Q.cpu().numpy()
out = (torch.empty((1, 10), device="cuda:0"), torch.empty((1, 10), dtype=torch.int64, device="cuda:0"))
for B in (1, 8):
    for name, fn in (("device in/out", lambda i: index.search(Q[i:i + B], 10, out=out) if B == 1 else index.search(Q[i:i + B], 10)),
                     ("numpy in/out", lambda i: index.search(Qh[i:i + B], 10))):
        for i in range(10): fn(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(200): fn(i % 56)
        torch.cuda.synchronize()
        print("B=%d %-14s %.1f us per search" % (B, name, (time.perf_counter() - t0) / 200 * 1e6), flush=True)

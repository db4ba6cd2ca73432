"""The whole batch curve of the Flat search on one box: time_batch_sweep.py n d storage k [iters]
Per batch size: the kernel the dispatcher took, mean hipEvent time of the scan launches of one search, the fraction of
8 TB/s (the bytes the index's timing record counts for the scan launches of that search) and of 2.5 PF (2*nq*n*d)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import semantic_search_amd as ssa
n, d, storage, k = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
ix = ssa.IndexFlatIP(d, storage=storage); ix.add_random_unit(n, 1234)
g = torch.Generator().manual_seed(1)
print("n=%d d=%d storage=%s k=%d" % (n, d, storage, k))
print("%6s %-28s %9s %9s %8s %8s %10s" % ("nq", "kernel", "ms", "us/query", "of HBM", "of MFMA", "QPS"))
for nq in [1, 2, 3, 4, 5, 6, 8, 9, 12, 16, 24, 32, 48, 64, 96, 128, 129, 192, 256, 257, 384, 512, 768, 1024, 2048, 4096]:
    Q = torch.randn((2, nq, d), generator=g); Q = (Q / Q.norm(dim=-1, keepdim=True)).cuda()
    for it in range(2):
        ix.search(Q[it % 2], k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(iters):
        ix.search(Q[it % 2], k)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters                      # whole search on the stream: scan launches + merges
    t = ix.last_timing()
    print("%6d %-28s %9.3f %9.2f %8.3f %8.3f %10.0f" % (nq, ix.last_kernel(), ms, ms * 1e3 / nq, t["last_scan_bytes"] / (ms * 1e-3) / 8e12,
                                                    2.0 * nq * n * d / (ms * 1e-3) / 2.5e15, nq / (ms * 1e-3)), flush=True)

"""Mean hipEvent time of the batched Flat contraction: time_mfma.py n d nq k [iters]  (SSMI_LIB selects a build)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import semantic_search_amd as ssa
n, d, nq, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
ix = ssa.IndexFlatIP(d, storage="f16"); ix.add_random_unit(n, 1234)
g = torch.Generator().manual_seed(1); Q = torch.randn((2, nq, d), generator=g); Q = (Q / Q.norm(dim=-1, keepdim=True)).cuda()
for it in range(3):
    ix.search(Q[it % 2], k)
torch.cuda.synchronize()
for it in range(iters):
    ix.search(Q[it % 2], k)
torch.cuda.synchronize()
ms = ix.scan_times_ms(iters)
fl = 2.0 * nq * n * d
print("%s n=%d nq=%d k=%d: %s mean %.3f ms min %.3f  -> %.1f TF/s (%.3f of 2.5 PF)" % (
    os.environ.get("SSMI_LIB", "default"), n, nq, k, ix.last_kernel(), ms.mean(), ms.min(), fl / ms.mean() / 1e9, fl / ms.mean() / 1e9 / 2500))

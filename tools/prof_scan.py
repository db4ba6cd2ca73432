"""A few searches of one batch size on the exact streaming scan (K1), for rocprofv3 --pmc passes:
prof_scan.py n d storage nq k   (SSMI_MFMA_MIN_B is raised so that the batch stays on the scan)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SSMI_MFMA_MIN_B"] = "1000000"
os.environ["SSMI_NO_SHADOW"] = "1"
import torch
import semantic_search_amd as ssa
n, d, storage, nq, k = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
ix = ssa.IndexFlatIP(d, storage=storage); ix.add_random_unit(n, 1234)
g = torch.Generator().manual_seed(1); Q = torch.randn((nq, d), generator=g); Q = (Q / Q.norm(dim=1, keepdim=True)).cuda()
for it in range(5):
    D, I = ix.search(Q, k)
torch.cuda.synchronize()
print(ix.last_kernel(), ix.last_timing())

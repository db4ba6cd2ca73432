import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semantic_search_amd as ssa
n, d, nq, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ix = ssa.IndexFlatIP(d, storage="f16"); ix.add_random_unit(n, 1234)
g = torch.Generator().manual_seed(1); Q = torch.randn((nq, d), generator=g); Q = (Q / Q.norm(dim=1, keepdim=True)).cuda()
for it in range(3):
    D, I = ix.search(Q, k)
torch.cuda.synchronize()
print(ix.last_timing())

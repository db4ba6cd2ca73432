import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import semantic_search_amd as ssa
from oracle import flat_ip
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import unit_rows
n, d, nq, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
X = unit_rows(n, d, 1234); Q = unit_rows(nq, d, 4321)
ix = ssa.IndexFlatIP(d); ix.add(X)
D, I = ix.search(Q, k)
Dr, Ir = flat_ip.search(X, Q, k)
np.set_printoptions(linewidth=200)
print("D ", D[0][:12]); print("Dr", Dr[0][:12]); print("I ", I[0][:12]); print("Ir", Ir[0][:12])
print(ix.last_timing())
R = ix.reconstruct_n(0, 3)
print("recon equal:", np.array_equal(R, X[:3]), R[0][:4], X[0][:4])
import torch
Dt, It = ix.search(torch.from_numpy(Q).cuda(), k)
print("torch path D", Dt[0][:5].cpu().numpy(), It[0][:5].cpu().numpy())

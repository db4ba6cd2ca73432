"""Profiling driver (rocprofv3 target): IVF-Flat batch-1 searches over n random unit rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semantic_search_amd as ssa
n, d, nlist, nprobe = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1234)
def rows(m):
    x = torch.randn((m, d), generator=g, device=dev, dtype=torch.float32)
    return x / x.norm(dim=1, keepdim=True)
ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage="f16"), d, nlist, ssa.METRIC_INNER_PRODUCT, storage="f16")
ivf.train(rows(min(n, 64 * nlist)))
for s in range(0, n, 65536):
    ivf.add(rows(min(65536, n - s)))
ivf.nprobe = nprobe
if os.environ.get("PROF_TIMING", "1") == "0":
    ivf.set_timing(False)
Q = rows(64)
for i in range(40):
    D, I = ivf.search(Q[i:i + 1], 10)
torch.cuda.synchronize()
print(ivf.last_timing(), "expected probed rows per query", nprobe * n // nlist)

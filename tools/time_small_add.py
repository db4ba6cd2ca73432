import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semantic_search_amd as ssa
d, nlist = 4096, 4096
g = torch.Generator(device="cuda:0").manual_seed(1)
def rows(m):
    x = torch.randn((m, d), generator=g, device="cuda:0"); return x / x.norm(dim=1, keepdim=True)
for storage in ("f32", "f16"):
    ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage=storage), d, nlist, ssa.METRIC_INNER_PRODUCT, storage=storage)
    ivf.train(rows(64 * nlist))
    ivf.add(rows(200000))
    for n in (1, 16, 256):
        xs = [rows(n) for _ in range(100)]
        xh = [x.cpu().numpy() for x in xs]
        for x in xs[:10]: ivf.add(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for x in xs: ivf.add(x)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for x in xh: ivf.add(x)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(storage, "add of %d vectors: %.0f us (device rows), %.0f us (host rows)" % (n, (t1 - t0) * 1e4, (t2 - t1) * 1e4), flush=True)

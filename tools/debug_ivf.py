import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import semantic_search_amd as ssa
n, d, nlist, nprobe = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
storage = sys.argv[5] if len(sys.argv) > 5 else "f16"
with_flat = (sys.argv[6] != "noflat") if len(sys.argv) > 6 else True
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1234)
def rows(m):
    x = torch.randn((m, d), generator=g, device=dev, dtype=torch.float32)
    return x / x.norm(dim=1, keepdim=True)
ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage=storage), d, nlist, ssa.METRIC_INNER_PRODUCT, storage=storage)
flat = ssa.IndexFlatIP(d, storage=storage) if with_flat else None
ntrain = min(n, 256 * nlist)
xt = rows(ntrain)
torch.cuda.synchronize(); t0 = time.perf_counter()
ivf.train(xt)
torch.cuda.synchronize(); t_train = time.perf_counter() - t0
print(f"train: {ntrain} pts x {nlist} lists: {t_train:.2f} s", flush=True)
del xt
g = torch.Generator(device=dev).manual_seed(99)
t_add = 0.0
chunk = 65536
for s in range(0, n, chunk):
    x = rows(min(chunk, n - s))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ivf.add(x)
    t_add += time.perf_counter() - t0
    if flat is not None: flat.add(x)
print(f"add: {n} vectors in {t_add:.2f} s ({n / t_add:.0f} vec/s)", flush=True)
ls = ivf.list_sizes(); print("list sizes: min %d median %d max %d" % (ls.min(), np.median(ls), ls.max()))
ivf.nprobe = nprobe
gq = torch.Generator(device=dev).manual_seed(4321)
Q = torch.randn((1000, d), generator=gq, device=dev); Q = Q / Q.norm(dim=1, keepdim=True)
# batch-1 latency
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100):
        D, I = ivf.search(Q[i:i + 1], 10)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"IVF batch-1: {100 / dt:.0f} QPS ({dt * 10:.3f} ms/query); last scan kernel {ivf.last_timing()['last_scan_ms']:.3f} ms")
for B in (64, 1000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    D, I = ivf.search(Q[:B], 10)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"IVF batch-{B}: {B / dt:.0f} QPS")
Di, Ii = ivf.search(Q, 10)
if flat is not None:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Df, If = flat.search(Q, 10)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"Flat batch-1000: {1000 / dt:.0f} QPS")
    Ii, If = Ii.cpu().numpy(), If.cpu().numpy()
    print("recall@10 vs Flat: %.4f" % np.mean([len(set(a) & set(b)) / 10 for a, b in zip(Ii, If)]))
    sizes = ls
    Dc, Ic = None, None

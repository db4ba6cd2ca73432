"""Per-chunk timing of IVF add (assign + append): time_ivf_add.py n d nlist [iid|clustered]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semantic_search_amd as ssa
n, d, nlist = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
data = sys.argv[4] if len(sys.argv) > 4 else "iid"
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1234)
centres = torch.randn((16384, d), generator=g, device=dev); centres /= centres.norm(dim=1, keepdim=True)
def rows(m):
    x = torch.randn((m, d), generator=g, device=dev) * (0.6 / d ** 0.5 if data == "clustered" else 1.0)
    if data == "clustered":
        x += centres[torch.randint(0, 16384, (m,), generator=g, device=dev)]
    return x / x.norm(dim=1, keepdim=True)
ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage="f16"), d, nlist, ssa.METRIC_INNER_PRODUCT, storage="f16")
flat = ssa.IndexFlatIP(d, storage="f16") if os.environ.get("WITH_FLAT") else None
xt = rows(min(n, 256 * nlist))
torch.cuda.synchronize(); t0 = time.perf_counter()
ivf.train(xt)
print("train %d rows: %.3f s" % (xt.shape[0], time.perf_counter() - t0), flush=True)
del xt
ts = []
for s in range(0, n, 65536):
    x = rows(min(65536, n - s))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ivf.add(x)
    ts.append(time.perf_counter() - t0)
    if flat is not None:
        flat.add(x)
    if len(ts) % 20 == 0:
        print("chunks %3d-%3d: %.2f ms per chunk" % (len(ts) - 20, len(ts), sum(ts[-20:]) / 20 * 1e3), flush=True)
print(data, "total add %.3f s" % sum(ts), "list sizes", int(ivf.list_sizes().min()), int(ivf.list_sizes().max()))

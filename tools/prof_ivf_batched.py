"""Batched IVF search (list-major MFMA scan): kernel time from the library's events and the bytes the scan must
stream (sum over lists of rows x ceil(probes / group) x row bytes), for clustered rows."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import semantic_search_amd as ssa
n, d, nlist, nprobe = (int(a) for a in sys.argv[1:5])
batches = [int(b) for b in sys.argv[5:]] or [1000, 4096]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1234)
ncl = 4 * nlist
centres = torch.randn((ncl, d), generator=g, device=dev); centres /= centres.norm(dim=1, keepdim=True)
def rows(m):
    x = torch.randn((m, d), generator=g, device=dev) * (0.6 / d ** 0.5) + centres[torch.randint(0, ncl, (m,), generator=g, device=dev)]
    return x / x.norm(dim=1, keepdim=True)
ST = os.environ.get("PROF_STORAGE", "f16")
ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage=ST), d, nlist, ssa.METRIC_INNER_PRODUCT, storage=ST)
ivf.train(rows(min(n, 64 * nlist)))
for s in range(0, n, 65536):
    ivf.add(rows(min(65536, n - s)))
ivf.nprobe = nprobe
sizes = ivf.list_sizes().astype(np.int64)
cq = ssa.IndexFlatIP(d, storage="f16"); cq.add(torch.from_numpy(ivf.get_centroids()).to(dev))
for B in batches:
    Q = rows(B)
    ivf.search(Q, 10); torch.cuda.synchronize()
    t0 = time.perf_counter(); ivf.search(Q, 10); torch.cuda.synchronize(); wall = time.perf_counter() - t0
    t = ivf.last_timing()
    bt = -t["last_grid"]
    _, Ic = cq.search(Q, nprobe)
    probes = np.bincount(Ic.cpu().numpy().ravel(), minlength=nlist)
    passes = -(-probes // bt)
    byts = float((sizes * passes).sum()) * d * 2  # fp16 bytes (fp32 lists are scanned through their fp16 shadow)
    print(f"B={B} group={bt}: wall {wall*1e3:.2f} ms ({B/wall:.0f} QPS), scan kernel {t['last_scan_ms']:.2f} ms, merge {t['last_merge_ms']:.3f} ms, "
          f"streams {byts/1e9:.1f} GB (union of probed lists {float(sizes[probes>0].sum())*d*2/1e9:.1f} GB) -> {byts/t['last_scan_ms']/1e9:.2f} TB/s in the scan kernel")

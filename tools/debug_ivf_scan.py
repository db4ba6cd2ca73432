"""IVF batch-1 scan kernel time vs nprobe (fixed overhead vs bandwidth)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import semantic_search_amd as ssa
n, d, nlist = int(sys.argv[1]), 4096, int(sys.argv[2])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1234)
def rows(m):
    x = torch.randn((m, d), generator=g, device=dev, dtype=torch.float32)
    return x / x.norm(dim=1, keepdim=True)
ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage="f16"), d, nlist, ssa.METRIC_INNER_PRODUCT, storage="f16")
ivf.train(rows(min(n, 64 * nlist)))
for s in range(0, n, 65536):
    ivf.add(rows(min(65536, n - s)))
Q = rows(64)
for nprobe in (8, 32, 128, 512):
    ivf.nprobe = nprobe
    ms, rws = [], []
    for i in range(24):
        ivf.search(Q[i:i + 1], 10)
        t = ivf.last_timing()
        if i >= 4: ms.append(t["last_scan_ms"]); rws.append(t["last_scan_bytes"] / 8192)
    ms, rws = np.mean(ms), np.mean(rws)
    print(f"nprobe={nprobe}: rows {rws:.0f}  scan {ms * 1e3:.1f} us  {rws * 8192 / ms / 1e6:.0f} GB/s")

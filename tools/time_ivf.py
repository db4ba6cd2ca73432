"""Per-query IVF scan timing: time_ivf.py n d nlist nprobe [iters]   (SSMI_LIB selects a build; with
libssmi_ivfstamp.so the last three searches print the per-wave wall-clock stamps)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import semantic_search_amd as ssa
n, d, nlist, nprobe = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 100
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
Q = rows(256)
for i in range(10):
    ivf.search(Q[i:i + 1], 10)
ms, by = [], []
for i in range(iters):
    ivf.search(Q[i % 256:i % 256 + 1], 10)
    t = ivf.last_timing()
    ms.append(t["last_scan_ms"]); by.append(t["last_scan_bytes"])
ms, by = np.array(ms), np.array(by)
ivf.set_timing(False)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(iters):
    ivf.search(Q[i % 256:i % 256 + 1], 10)
torch.cuda.synchronize()
e2e = (time.perf_counter() - t0) / iters
print("%s: %s scan %.1f us for %.1f MB -> %.2f TB/s (%.3f of 8 TB/s); end-to-end %.1f us per search" % (
    os.path.basename(os.environ.get("SSMI_LIB", "default")), ivf.last_kernel(), ms.mean() * 1e3, by.mean() / 1e6,
    by.mean() / ms.mean() / 1e9, by.mean() / ms.mean() / 1e9 / 8.0, e2e * 1e6))
if "stamp" in os.environ.get("SSMI_LIB", ""):
    ivf.set_timing(True)
    os.environ["SSMI_IVF_STAMP_PRINT"] = "1"
    for i in range(3):
        ivf.search(Q[i:i + 1], 10)
    torch.cuda.synchronize()

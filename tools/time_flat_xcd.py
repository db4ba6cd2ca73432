"""K1 finish times per XCD: time_flat_xcd.py n d storage [iters]  (run with SSMI_LIB=.../libssmi_scanstamp.so
SSMI_SCAN_STAMP_PRINT=1: every search prints when the waves of each XCD finished their rows)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semantic_search_amd as ssa
n, d, storage = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
index = ssa.IndexFlatIP(d, storage=storage)
index.add_random_unit(n, seed=1)
g = torch.Generator(device="cuda:0").manual_seed(2)
Q = torch.randn((64, d), generator=g, device="cuda:0")
Q /= Q.norm(dim=1, keepdim=True)
out = (torch.empty((1, 10), device="cuda:0"), torch.empty((1, 10), dtype=torch.int64, device="cuda:0"))
stamp = os.environ.pop("SSMI_SCAN_STAMP_PRINT", None)
for i in range(20):
    index.search(Q[i:i + 1], 10, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50):
    index.search(Q[i % 64:i % 64 + 1], 10, out=out)
torch.cuda.synchronize()
print("%.3f ms per search, kernel %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3, index.last_timing()["last_scan_ms"]))
if stamp:
    os.environ["SSMI_SCAN_STAMP_PRINT"] = stamp
    for i in range(iters):
        index.search(Q[i:i + 1], 10, out=out)
    torch.cuda.synchronize()

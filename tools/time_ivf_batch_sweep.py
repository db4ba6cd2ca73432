"""The batch curve of the IVF search on one box: time_ivf_batch_sweep.py n d nlist nprobe k [iters]
(clustered rows as in bench.py's IVF workload).  Per batch size: kernel of the list scan, whole-search time on the
stream, microseconds per query, QPS.  SSMI_SWEEP_AB=1: every batch size with the scan forced per query
(SSMI_IVF_LISTMAJOR=0) and list-major (=1) next to the dispatcher's choice."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semantic_search_amd as ssa
n, d, nlist, nprobe, k = (int(a) for a in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1234)
centres = torch.randn((16384, d), generator=torch.Generator(device=dev).manual_seed(999), device=dev)
centres /= centres.norm(dim=1, keepdim=True)
def rows(m):
    x = torch.randn((m, d), generator=g, device=dev) * (0.6 / d ** 0.5) + centres[torch.randint(0, 16384, (m,), generator=g, device=dev)]
    return x / x.norm(dim=1, keepdim=True)
ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage="f16"), d, nlist, ssa.METRIC_INNER_PRODUCT, storage="f16")
ivf.train(rows(min(n, 256 * nlist)))
for s in range(0, n, 65536):
    ivf.add(rows(min(65536, n - s)))
ivf.nprobe = nprobe
ivf.set_timing(False)
Qall = rows(2 * 4096)
print("n=%d d=%d nlist=%d nprobe=%d k=%d" % (n, d, nlist, nprobe, k))
print("%6s %-34s %9s %9s %10s" % ("nq", "scan kernel", "ms", "us/query", "QPS"))
ab = os.environ.get("SSMI_SWEEP_AB") == "1"
sizes = [1, 2, 3, 4, 5, 8, 9, 12, 15, 16, 17, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1000, 1024, 2048, 4096]
for nq, mode in [(a, m) for a in sizes for m in (("-", "0", "1") if ab else ("-",))]:
    if mode == "-":
        os.environ.pop("SSMI_IVF_LISTMAJOR", None)
    else:
        os.environ["SSMI_IVF_LISTMAJOR"] = mode
    Q = [Qall[:nq], Qall[4096:4096 + nq]]
    for it in range(2):
        ivf.search(Q[it % 2], k)
    torch.cuda.synchronize()
    its = iters * (20 if nq <= 64 else 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(its):
        ivf.search(Q[it % 2], k)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / its
    print("%6d %-34s %9.3f %9.2f %10.0f" % (nq, ivf.last_kernel() + ("" if mode == "-" else " [%s]" % mode), ms, ms * 1e3 / nq, nq / (ms * 1e-3)), flush=True)

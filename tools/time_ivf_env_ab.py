"""Batched IVF search of one shape with environment switches flipped in ONE process (interleaved rounds), results compared:
time_ivf_env_ab.py n d nlist nprobe nq k iters VAR=a,b [...]   ('-' = variable unset; clustered rows as in bench.py's IVF workload)"""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semantic_search_amd as ssa
n, d, nlist, nprobe, nq, k, iters = (int(a) for a in sys.argv[1:8])
variants = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[8:]]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1234)
centres = torch.randn((16384, d), generator=torch.Generator(device=dev).manual_seed(999), device=dev)
centres /= centres.norm(dim=1, keepdim=True)
def rows(m):
    x = torch.randn((m, d), generator=g, device=dev) * (0.6 / d ** 0.5) + centres[torch.randint(0, 16384, (m,), generator=g, device=dev)]
    return x / x.norm(dim=1, keepdim=True)
ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage="f16"), d, nlist, ssa.METRIC_INNER_PRODUCT, storage="f16")
ivf.train(rows(min(n, 64 * nlist)))
for s in range(0, n, 65536):
    ivf.add(rows(min(65536, n - s)))
ivf.nprobe = nprobe
ivf.set_timing(False)
Q = rows(2 * nq).reshape(2, nq, d)
ref = None
for rnd in range(2):
    for combo in itertools.product(*[v for _, v in variants]):
        for (name, _), val in zip(variants, combo):
            if val == "-":
                os.environ.pop(name, None)
            else:
                os.environ[name] = val
        for it in range(2):
            ivf.search(Q[it % 2], k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(iters):
            D, I = ivf.search(Q[(it + 1) % 2], k)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        if ref is None:
            ref = (D.clone(), I.clone())
        assert torch.equal(ref[0], D) and torch.equal(ref[1], I), "variant changes the result"
        print("ivf n=%d nlist=%d nprobe=%d nq=%d k=%d round %d %s: %s %.3f ms per search, %.0f QPS" % (
            n, nlist, nprobe, nq, k, rnd, " ".join("%s=%s" % (nm, v) for (nm, _), v in zip(variants, combo)), ivf.last_kernel(), ms, nq / ms * 1e3), flush=True)

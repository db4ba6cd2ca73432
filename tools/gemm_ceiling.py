"""What the vendor GEMM (hipBLASLt through torch.matmul) sustains on the shape of the batched Flat search, and what a
library-only search (chunked GEMM + torch.topk + merge) costs end to end: gemm_ceiling.py [n] [nq] [d] [k].
A calibration of the 2.5 PFLOP/s roof for K2; not part of the product path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
d = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn((n, d), generator=g, device=dev, dtype=torch.float16)
Q = torch.randn((nq, d), generator=g, device=dev, dtype=torch.float16)
flops = 2.0 * nq * n * d

def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

# 1. the GEMM alone, whole database in one call (writes the nq x n fp16 score matrix)
out = torch.empty((nq, n), device=dev, dtype=torch.float16)
ms = timed(lambda: torch.matmul(Q, X.t(), out=out))
print("GEMM only, one call (%d x %d x %d, fp16 in/out): %.2f ms = %.0f TFLOP/s (%.3f of 2.5 PF)" % (nq, n, d, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500), flush=True)
del out
# 2. the GEMM alone in chunks of 65536 rows (score tile stays in the Infinity Cache / L2)
chunk = 65536
buf = torch.empty((nq, chunk), device=dev, dtype=torch.float16)
def gemm_chunks():
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        torch.matmul(Q, X[s:e].t(), out=buf[:, : e - s])
ms = timed(gemm_chunks)
print("GEMM only, %d-row chunks: %.2f ms = %.0f TFLOP/s (%.3f of 2.5 PF)" % (chunk, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500), flush=True)
# 3. library search: chunked GEMM + topk per chunk + final topk over the chunk winners
def lib_search():
    Ds, Is = [], []
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        sc = torch.matmul(Q, X[s:e].t(), out=buf[:, : e - s])
        dd, ii = torch.topk(sc, min(k, e - s), dim=1)
        Ds.append(dd); Is.append(ii + s)
    D = torch.cat(Ds, dim=1); I = torch.cat(Is, dim=1)
    dd, jj = torch.topk(D, k, dim=1)
    return dd, torch.gather(I, 1, jj)
ms = timed(lib_search, iters=3)
print("library search (chunked GEMM + torch.topk + merge), k=%d: %.2f ms = %.0f TFLOP/s equivalent (%.3f of 2.5 PF), %.0f queries/s" % (
    k, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500, nq / ms * 1e3), flush=True)
# 4. this repository's K2 on the same operands
import semantic_search_amd as ssa
ix = ssa.IndexFlatIP(d, storage="f16")
for s in range(0, n, 262144):
    ix.add(X[s:s + 262144])
Qf = Q.float()
ix.search(Qf, k); torch.cuda.synchronize()
ms = timed(lambda: ix.search(Qf, k), iters=5)
print("libssmi %s (fused top-k, no score matrix), k=%d: %.2f ms = %.0f TFLOP/s (%.3f of 2.5 PF), %.0f queries/s" % (
    ix.last_kernel(), k, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500, nq / ms * 1e3), flush=True)

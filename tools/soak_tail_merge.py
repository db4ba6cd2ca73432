"""Soak test of the merge tails (tail_merge.h): many batch-1 / batch-3 searches in a row, every result compared
with the first result of the same query.  A flaw in the publish / ticket / read protocol across XCDs (stale lists)
would show up as a rare mismatch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semantic_search_amd as ssa
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
def rows(m, d):
    x = torch.randn((m, d), generator=g, device=dev)
    return x / x.norm(dim=1, keepdim=True)
d = 1024
flat = ssa.IndexFlatIP(d); flat.add(rows(300_000, d))
ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage="f16"), d, 256, ssa.METRIC_INNER_PRODUCT, storage="f16")
ivf.train(rows(20_000, d)); ivf.add(rows(400_000, d)); ivf.nprobe = 16
for name, ix, nq, k in (("flat b1", flat, 1, 10), ("flat b3 k100", flat, 3, 100), ("ivf b1", ivf, 1, 10), ("ivf b3 k100", ivf, 3, 100)):
    ix.set_timing(False)
    Q = rows(16 * nq, d).view(16, nq, d)
    ref = [tuple(t.clone() for t in ix.search(Q[i], k)) for i in range(16)]
    bad = 0
    for it in range(reps):
        D, I = ix.search(Q[it % 16], k)
        if it % 64 == 63 or it == reps - 1:  # compare in bursts so that the searches run back to back
            torch.cuda.synchronize()
        if not (torch.equal(I, ref[it % 16][1]) and torch.equal(D, ref[it % 16][0])):
            bad += 1
    torch.cuda.synchronize()
    print(f"{name}: {reps} searches, {bad} mismatches", flush=True)

"""Flat search of one shape with environment switches flipped in ONE process on one box (interleaved rounds), results
compared between the variants:  time_env_ab.py n d storage nq k iters VAR=a,b [VAR2=c,d ...]   ('-' = variable unset)"""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semantic_search_amd as ssa
n, d, storage, nq, k, iters = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
variants = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[7:]]
ix = ssa.IndexFlatIP(d, storage=storage); ix.add_random_unit(n, 1234)
el = 2 if storage == "f16" else 4
g = torch.Generator().manual_seed(1); Q = torch.randn((2, nq, d), generator=g); Q = (Q / Q.norm(dim=-1, keepdim=True)).cuda()
ref = None
for rnd in range(2):
    for combo in itertools.product(*[v for _, v in variants]):
        for (name, _), val in zip(variants, combo):
            if val == "-":
                os.environ.pop(name, None)
            else:
                os.environ[name] = val
        for it in range(2):
            ix.search(Q[it % 2], k)
        torch.cuda.synchronize()
        for it in range(iters):
            D, I = ix.search(Q[(it + 1) % 2], k)
        torch.cuda.synchronize()
        ms = ix.scan_times_ms(iters)
        if ref is None:
            ref = (D.clone(), I.clone())
        if os.environ.get("SSMI_AB_LOOSE"):  # variants that sum in another order: same ids, scores to 1e-5
            assert torch.equal(ref[1], I) and torch.allclose(ref[0], D, rtol=1e-5, atol=1e-6), "variant changes the result"
        else:
            assert torch.equal(ref[0], D) and torch.equal(ref[1], I), "variant changes the result"
        print("n=%d %s nq=%d k=%d round %d %s: %s mean %.3f ms min %.3f -> %.2f TB/s (%.3f of 8), %.0f TFLOP/s (%.3f of 2500)" % (
            n, storage, nq, k, rnd, " ".join("%s=%s" % (nm, v) for (nm, _), v in zip(variants, combo)), ix.last_kernel(), ms.mean(), ms.min(),
            n * d * el / ms.mean() / 1e9, n * d * el / ms.mean() / 8e9, 2.0 * nq * n * d / ms.mean() / 1e9, 2.0 * nq * n * d / ms.mean() / 1e9 / 2500), flush=True)

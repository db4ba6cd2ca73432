"""k > n on the batched (MFMA) Flat search: valid results per query, repeated with fresh indexes, after free HBM has been
poisoned (0xFF bytes: NaN as fp16/fp32, -1 as ints, the largest key), so that a read of memory the library never
wrote shows.   debug_kgtn.py reps n nq k [poison GiB]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import semantic_search_amd as ssa
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n, nq, k = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (300, 20, 1024)
gib = int(sys.argv[5]) if len(sys.argv) > 5 else 64
poison = torch.empty(gib << 30, dtype=torch.uint8, device="cuda").fill_(255)
torch.cuda.synchronize(); del poison; torch.cuda.empty_cache()
rng = np.random.default_rng(7)
bad = 0
d = 4096
for rep in range(reps):
    X = rng.standard_normal((n, d)).astype(np.float32); X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = rng.standard_normal((nq, d)).astype(np.float32); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    ix = ssa.IndexFlatIP(d, storage="f16"); ix.add(X)
    D, I = ix.search(Q, k)
    valid = (I >= 0).sum(1); n_exp = min(n, k)
    rec = ix.reconstruct_batch(np.arange(n))
    rows_ok = np.array_equal(rec, X.astype(np.float16).astype(np.float32))
    if not (valid == n_exp).all() or not rows_ok or np.isnan(D).any():
        bad += 1
        print("rep %d: valid per query %s rows_ok %s nan %s kernel %s" % (rep, valid.tolist(), rows_ok, np.isnan(D).any(), ix.last_kernel()), flush=True)
        for q in np.nonzero(valid != n_exp)[0][:3]:
            ids = I[q][I[q] >= 0]
            sc = (X.astype(np.float16).astype(np.float32) @ Q[q].astype(np.float16).astype(np.float32))
            miss = sorted(set(range(n)) - set(ids.tolist()))
            print("  q%d: %d ids (%d distinct); last valid D %.5f; scores of missing ids: min %.5f max %.5f; of present ids: min %.5f" % (
                q, len(ids), len(set(ids.tolist())), D[q][len(ids) - 1] if len(ids) else float("nan"), sc[miss].min(), sc[miss].max(),
                sc[ids].min() if len(ids) else float("nan")))
print("bad repetitions:", bad)

"""Replays one case of tests/test_fuzz_gpu.py::test_ivf_sweep (salt, seed) and prints the vectors whose list is not
the best one: row, piece of the add it arrived in, best / assigned score.  GPU box only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
salt, seed = int(sys.argv[1]), int(sys.argv[2])
SALT = 100000 * salt
from conftest import unit_rows
rng = np.random.default_rng(2000 + seed + SALT)
d = int(rng.choice([24, 64, 100, 256, 512, 1024, 2048, 4096]))
nlist = int(rng.choice([1, 3, 16, 40, 64, 200]))
n = int(rng.choice([nlist, 257, 1000, 3071, 6000])); n = max(n, nlist)
nq = int(rng.choice([1, 2, 5, 8, 9, 17, 64, 130, 300]))
k = int(rng.choice([1, 10, 64, 65, 100, 256, 300]))
nprobe = int(rng.choice([1, 2, 5, nlist, nlist + 3]))
storage = str(rng.choice(["f32", "f16"]))
if rng.random() < 0.4:
    os.environ["SSMI_SEG_SHIFT"] = str(int(rng.choice([8, 10])))
if storage == "f32":
    os.environ["SSMI_NO_SHADOW"] = "1"
print(f"n={n} d={d} nlist={nlist} storage={storage} SEG_SHIFT={os.environ.get('SSMI_SEG_SHIFT')}", flush=True)
import semantic_search_amd as ssa
cen = unit_rows(max(nlist, 8), d, 9000 + seed)
C = cen[:nlist]
X = cen[rng.integers(0, len(cen), n)] + 0.6 * unit_rows(n, d, 9100 + seed)
X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
index = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage=storage), d, nlist, ssa.METRIC_INNER_PRODUCT, storage=storage)
index.set_centroids(C)
pieces = []
cut = int(rng.integers(0, n + 1))
if cut:
    index.add(X[:cut]); pieces.append((0, cut))
while cut < n:
    step = int(rng.choice([1, 3, 16, 16, 250]))
    index.add(X[cut:cut + step]); pieces.append((cut, min(n, cut + step)))
    cut += step
labels = index.assignments()
sc = X.astype(np.float64) @ C.astype(np.float64).T
best = sc.max(1); got = sc[np.arange(n), labels]
bad = np.nonzero(best - got > 2e-3 * np.abs(best) + 2e-6)[0]
print("pieces", pieces)
for r in bad:
    p = [q for q in pieces if q[0] <= r < q[1]][0]
    print(f"row {r} piece {p} (size {p[1]-p[0]}, offset {r-p[0]}) scores {sc[r]} label {labels[r]}")
print(len(bad), "bad rows")

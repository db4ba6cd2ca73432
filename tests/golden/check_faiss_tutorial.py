"""Independent re-derivation of tests/golden/faiss_tutorial_flat.json: float64 brute-force L2 k-NN over the tutorial's
data must give the ids (and, for the sanity block, the distances) FAISS's documentation prints.  python tests/golden/check_faiss_tutorial.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import faiss_tutorial as ft

xb, xq = ft.data()
X = xb.astype(np.float64)
for name, q, I_pub, D_pub in ft.blocks(xb, xq):
    d2 = ((X[None, :, :] - q.astype(np.float64)[:, None, :]) ** 2).sum(-1)
    I = np.argsort(d2, axis=1, kind="stable")[:, :4]
    assert (I == I_pub).all(), (name, I, I_pub)
    if D_pub is not None:
        np.testing.assert_allclose(np.take_along_axis(d2, I, 1), D_pub, rtol=1e-5, atol=1e-5)
    print(name, "ids match the published output")

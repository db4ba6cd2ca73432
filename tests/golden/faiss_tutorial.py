"""The FAISS tutorial problem (tutorial/python/1-Flat.py) and its exact mapping onto inner-product search.

tests/golden/faiss_tutorial_flat.json holds the output FAISS's documentation prints for it.  L2 search maps onto
the repository's inner-product path exactly:  ||x - q||^2 = ||q||^2 - 2 <[q, 1], [x, -||x||^2 / 2]>,  so the row
that maximises the augmented inner product is the L2 nearest neighbour and D_L2 = ||q||^2 - 2 D_IP."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def published():
    return json.load(open(os.path.join(HERE, "faiss_tutorial_flat.json")))


def data():
    d, nb, nq = 64, 100000, 10000
    rs = np.random.RandomState(1234)            # np.random.seed(1234) of the tutorial (legacy generator)
    xb = rs.random_sample((nb, d)).astype("float32")
    xb[:, 0] += np.arange(nb) / 1000.
    xq = rs.random_sample((nq, d)).astype("float32")
    xq[:, 0] += np.arange(nq) / 1000.
    return xb, xq


def as_inner_product(xb, q):
    """(X', Q', qnorm2): the augmented fp32 problem of one block of queries, translated by the block's mean query."""
    c = q.astype(np.float64).mean(0)
    X = xb.astype(np.float64) - c
    Q = q.astype(np.float64) - c
    Xa = np.concatenate([X, -0.5 * (X ** 2).sum(1, keepdims=True)], axis=1).astype(np.float32)
    Qa = np.concatenate([Q, np.ones((len(Q), 1))], axis=1).astype(np.float32)
    return np.ascontiguousarray(Xa), np.ascontiguousarray(Qa), (Q ** 2).sum(1)


def blocks(xb, xq):
    g = published()
    return [("sanity", xb[:5], np.array(g["sanity_I"]), np.array(g["sanity_D"])),
            ("first5", xq[:5], np.array(g["first5_I"]), None),
            ("last5", xq[-5:], np.array(g["last5_I"]), None)]

"""CPU tests of the oracle itself.  Parity is unpinned by the reference (no tests,
fixtures or FAISS in reach: SURVEY.md 8c), so the oracle is pinned against (a) a
float64 brute force on small cases, (b) its independent C restatement, (c) the
golden vectors in tests/golden/ that tests/golden/make_golden.py generated and
(d) the one external known answer in reach: the output FAISS's own documentation
prints for its Flat tutorial (tests/golden/faiss_tutorial_flat.json), mapped exactly
onto the inner-product search of this path."""
import json
from pathlib import Path

import numpy as np
import pytest

from conftest import unit_rows
from oracle import c_oracle, flat_ip

GOLDEN = Path(__file__).resolve().parent / "golden"


def brute(X, Q, k):
    s = Q.astype(np.float64) @ X.astype(np.float64).T
    I = np.full((len(Q), k), -1, np.int64)
    D = np.full((len(Q), k), flat_ip.NEG_MAX, np.float32)
    for qi in range(len(Q)):
        order = sorted(range(X.shape[0]), key=lambda j: (-s[qi, j], j))[:k]
        I[qi, :len(order)] = order
        D[qi, :len(order)] = s[qi, order]
    return D, I


@pytest.mark.parametrize("n,d,nq,k", [(200, 16, 5, 7), (64, 8, 3, 64), (10, 4, 2, 20), (1000, 128, 4, 10)])
def test_numpy_oracle_vs_float64_bruteforce(n, d, nq, k):
    X, Q = unit_rows(n, d, 1), unit_rows(nq, d, 2)
    D, I = flat_ip.search(X, Q, k, block=97)
    Db, Ib = brute(X, Q, k)
    ok, msg, _ = flat_ip.check_against(D, I, Db, Ib, X, Q, rtol=1e-5, tie_rtol=1e-5)
    assert ok, msg


def test_c_oracle_matches_numpy_oracle():
    X, Q = unit_rows(5000, 256, 3), unit_rows(6, 256, 4)
    D, I = flat_ip.search(X, Q, 10)
    Dc, Ic = c_oracle.search(X, Q, 10)
    ok, msg, _ = flat_ip.check_against(Dc, Ic, D, I, X, Q, rtol=1e-5, tie_rtol=1e-5)
    assert ok, msg


def test_c_ivf_oracle_matches_numpy_ivf_oracle():
    """oracle/ivf_flat.c (the CPU IVF baseline of bench.py) against oracle/ivf_flat.py on the same centroids,
    assignment and queries: probed lists, ids and scores; nprobe = nlist equals exact search; short results pad."""
    from oracle import ivf_flat
    n, d, nlist = 4000, 128, 32
    X, Q = unit_rows(n, d, 5), unit_rows(7, d, 6)
    C = unit_rows(nlist, d, 7)
    labels = ivf_flat.assign(X, C)
    g = c_oracle.GroupedIVF(X, C, labels)
    for nprobe, k in ((1, 10), (5, 10), (5, 300), (nlist, 10)):
        Dc, Ic, Pc = g.search(Q, k, nprobe)
        D, I, P = ivf_flat.search(X, C, labels, Q, k, nprobe)
        np.testing.assert_array_equal(Pc, P)
        ok, msg, _ = flat_ip.check_against(Dc, Ic, D, I, X, Q, rtol=1e-5, tie_rtol=1e-5)
        assert ok, (nprobe, k, msg)
    D, I = flat_ip.search(X, Q, 10)
    ok, msg, _ = flat_ip.check_against(Dc, Ic, D, I, X, Q, rtol=1e-5, tie_rtol=1e-5)
    assert ok, msg


def test_padding_and_tie_order():
    X = unit_rows(6, 32, 5)
    X[4] = X[1]
    D, I = flat_ip.search(X, X[1], 8)
    assert list(I[0, :2]) == [1, 4] and (I[0, 6:] == -1).all() and (D[0, 6:] == flat_ip.NEG_MAX).all()
    Dc, Ic = c_oracle.search(X, X[1], 8)
    assert list(Ic[0]) == list(I[0])
    D0, I0 = flat_ip.search(np.zeros((0, 32), np.float32), X[:2], 3)
    assert (I0 == -1).all()


def test_check_against_rejects_wrong_ids_and_accepts_ties():
    X, Q = unit_rows(500, 64, 6), unit_rows(2, 64, 7)
    D, I = flat_ip.search(X, Q, 5)
    bad = I.copy(); bad[0, 2] = (I[0, 2] + 1) % 500 if (I[0, 2] + 1) % 500 not in I[0] else (I[0, 2] + 7) % 500
    ok, msg, _ = flat_ip.check_against(D, bad, D, I, X, Q, 1e-5, 1e-5)
    assert not ok
    X2 = X.copy(); X2[400] = X2[I[0, 2]]  # an exact tie: either id is acceptable at that rank
    D2, I2 = flat_ip.search(X2, Q, 5)
    swapped = I2.copy()
    pos = list(I2[0]).index(min(I[0, 2], 400))
    if I2[0, pos + 1] == max(I[0, 2], 400):
        swapped[0, pos], swapped[0, pos + 1] = I2[0, pos + 1], I2[0, pos]
        ok, msg, _ = flat_ip.check_against(D2, swapped, D2, I2, X2, Q, 1e-5, 1e-5)
        assert ok, msg


def test_golden_flat_vectors():
    """BASELINE config 1 shape (10k x 4096, 100 queries, k=10) and edge cases, regenerated from seeds."""
    meta = json.loads((GOLDEN / "flat_cases.json").read_text())
    data = np.load(GOLDEN / "flat_expected.npz")
    for case in meta["cases"]:
        X = unit_rows(case["n"], case["d"], case["seed_db"])
        if case.get("storage") == "f16":
            X = X.astype(np.float16).astype(np.float32)
        Q = unit_rows(case["nq"], case["d"], case["seed_q"])
        D, I = flat_ip.search(X, Q, case["k"])
        ok, msg, _ = flat_ip.check_against(D, I, data[case["name"] + "_D"], data[case["name"] + "_I"], X, Q, 1e-5, 1e-5)
        assert ok, f"{case['name']}: {msg}"


def test_oracle_reproduces_the_output_faiss_publishes_for_its_flat_tutorial():
    """FAISS wiki 'Getting started' / tutorial/python/1-Flat.py: exact search over 100000 x 64 vectors (seed 1234),
    k = 4, printed for the first five database vectors and the first / last five queries.  L2 there, inner product here:
    ||x-q||^2 = ||q||^2 - 2 <[q,1],[x,-||x||^2/2]> turns one into the other exactly (tests/golden/faiss_tutorial.py), so
    both oracles must return FAISS's published ids, in FAISS's order, and the published distances."""
    import sys
    sys.path.insert(0, str(GOLDEN))
    import faiss_tutorial as ft
    xb, xq = ft.data()
    for name, q, I_pub, D_pub in ft.blocks(xb, xq):
        Xa, Qa, qn = ft.as_inner_product(xb, q)
        for impl in (flat_ip, c_oracle):
            D, I = impl.search(Xa, Qa, 4)
            np.testing.assert_array_equal(I, I_pub, err_msg=f"{name} {impl.__name__}")
            if D_pub is not None:
                np.testing.assert_allclose(qn[:, None] - 2.0 * D.astype(np.float64), D_pub, rtol=2e-5, atol=2e-5, err_msg=name)

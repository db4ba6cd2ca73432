"""Pins the oracle (and, on a GPU, the HIP path) against the real FAISS whenever `faiss` can be imported.

The reference delegates this whole path to `faiss-cpu>=1.7.4` (requirements.txt:4), which is not installed in the
build image and cannot be fetched: there the tests below SKIP and parity stays "unpinned" as
DESIGN.md section 5 records.  On a machine that has FAISS they turn the documented-semantics restatement into a
checked one: same call shapes as core/index_manager.py:363-378 and core/index_factory.py:138-146."""
import numpy as np
import pytest

from conftest import unit_rows
from oracle import flat_ip, ivf_flat

faiss = pytest.importorskip("faiss")


def test_oracle_flat_matches_faiss():
    X, Q = unit_rows(10_000, 4096, 1234), unit_rows(100, 4096, 4321)   # BASELINE configs[0]
    index = faiss.IndexFlatIP(4096)
    index.add(X)
    D, I = index.search(Q, 10)
    Do, Io = flat_ip.search(X, Q, 10)
    ok, msg, _ = flat_ip.check_against(Do, Io, D, I, X, Q)
    assert ok, msg
    D, I = index.search(Q[:3], 20_000)                                 # k > ntotal: -1 / -FLT_MAX padding
    assert (I[:, 10_000:] == -1).all() and (D[:, 10_000:] == flat_ip.NEG_MAX).all()


def test_oracle_ivf_matches_faiss_given_its_centroids():
    d, n, nlist = 256, 20_000, 64
    X, Q = unit_rows(n, d, 7), unit_rows(50, d, 8)
    index = faiss.IndexIVFFlat(faiss.IndexFlatIP(d), d, nlist, faiss.METRIC_INNER_PRODUCT)
    index.train(X)
    index.add(X)
    C = index.quantizer.reconstruct_n(0, nlist)
    labels = ivf_flat.assign(X, C)
    for nprobe in (1, 10, nlist):
        index.nprobe = nprobe
        D, I = index.search(Q, 10)
        Do, Io, _ = ivf_flat.search(X, C, labels, Q, 10, nprobe)
        ok, msg, _ = flat_ip.check_against(Do, Io, D, I, X, Q)
        assert ok, f"nprobe={nprobe}: {msg}"


@pytest.mark.gpu
def test_hip_flat_matches_faiss(ssa):
    X, Q = unit_rows(10_000, 4096, 1234), unit_rows(100, 4096, 4321)
    ref = faiss.IndexFlatIP(4096)
    ref.add(X)
    D, I = ref.search(Q[:1], 10)
    ours = ssa.IndexFlatIP(4096)
    ours.add(X)
    Dg, Ig = ours.search(Q[:1], 10)
    ok, msg, _ = flat_ip.check_against(Dg, Ig, D, I, X, Q[:1])
    assert ok, msg

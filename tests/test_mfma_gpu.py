"""GPU parity of the large-batch MFMA path (fp16 storage, nq >= 9) vs the oracle.

The contraction multiplies fp16 rows by fp16-rounded queries and accumulates in
fp32, so against the oracle run on the SAME rounded operands the only difference
is summation order (tolerance 1e-5).  Against fp32 queries the north-star rule
applies (scores within 1e-3 relative, ids equal except at 1e-3-relative ties).
"""
import numpy as np
import pytest

from conftest import unit_rows
from oracle import flat_ip

pytestmark = pytest.mark.gpu


def r16(a):
    return a.astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("n,d,nq,k", [
    (10000, 4096, 100, 10),   # BASELINE config 1 shape on the MFMA path
    (3000, 4096, 300, 100),   # two query tiles, config-3 k
    (1037, 4096, 9, 10),      # smallest MFMA batch, ragged tail tile
    (5000, 256, 64, 10),
    (700, 64, 33, 5),
    (2000, 1024, 16, 500),
    (300, 4096, 20, 1024),    # k > n: padding
    (70000, 512, 12, 10),     # many row tiles per slice -> threshold path exercised
    (30000, 2048, 50, 10),    # Matryoshka truncations on the register-streaming kernel: 64-query form ...
    (8000, 1024, 120, 64),    # ... and 128-query form
])
def test_mfma_matches_oracle_on_rounded_operands(ssa, n, d, nq, k):
    X, Q = unit_rows(n, d, 1234), unit_rows(nq, d, 4321)
    index = ssa.IndexFlatIP(d, storage="f16")
    index.add(X)
    D, I = index.search(Q, k)
    assert index.last_timing()["last_path"] == 1
    Xh, Qh = r16(X), r16(Q)
    Dr, Ir = flat_ip.search(Xh, Qh, k)
    ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xh, Qh, rtol=1e-5, tie_rtol=1e-5)
    assert ok, msg
    # north-star rule against un-rounded fp32 queries
    Dr2, Ir2 = flat_ip.search(Xh, Q, k)
    ok, msg, nm = flat_ip.check_against(D, I, Dr2, Ir2, Xh, Q, rtol=1e-3, tie_rtol=1e-3, atol=1e-5)
    assert ok, msg
    index.close()


def test_mfma_and_scan_paths_agree(ssa, monkeypatch):
    X, Q = unit_rows(20000, 4096, 5), unit_rows(40, 4096, 6)
    index = ssa.IndexFlatIP(4096, storage="f16")
    index.add(X)
    D1, I1 = index.search(Q, 10)
    assert index.last_timing()["last_path"] == 1
    monkeypatch.setenv("SSMI_MFMA_MIN_B", "1000000")
    D0, I0 = index.search(Q, 10)
    assert index.last_timing()["last_path"] == 0
    Xh = r16(X)
    ok, msg, nm = flat_ip.check_against(D1, I1, D0, I0, Xh, Q, rtol=1e-3, tie_rtol=1e-3, atol=1e-5)
    assert ok, msg
    recall = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(I0, I1)])
    assert recall > 0.99


def test_mfma_duplicate_rows_tie_order(ssa):
    X = unit_rows(1000, 256, 7)
    X[500] = X[3]; X[900] = X[3]
    Q = np.repeat(X[3][None], 10, axis=0)
    index = ssa.IndexFlatIP(256, storage="f16")
    index.add(X)
    D, I = index.search(Q, 4)
    assert index.last_timing()["last_path"] == 1
    assert (I[:, :3] == np.array([3, 500, 900])).all()

"""GPU parity of K1m (csrc/scan_mm.h: the streaming scan with its multiply-adds on the fp32 matrix instruction, four
rows per wave pass) against the CPU oracle, through the C-ABI.  Same tolerance as the register-FMA scan it stands in
for (tests/test_flat_gpu.py): the arithmetic is an fp32 fmaf chain either way, only the order of the sums differs."""
import numpy as np
import pytest

from conftest import unit_rows
from oracle import flat_ip

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _rounded(X, storage):
    if storage == "f16":
        return X.astype(np.float16).astype(np.float32)
    if storage == "bf16":
        u = X.view(np.uint32).astype(np.uint64)
        return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)
    return X


@pytest.mark.parametrize("n,d,nq,k,storage,kernel", [
    (8000, 256, 1, 10, "f32", "flat_scan_mm_kernel"),    # one query on 1 KiB rows runs as a quad (three copies of it)
    (8000, 512, 2, 10, "f32", "flat_scan_mm_kernel"),    # 2 KiB rows
    (6000, 512, 1, 10, "f16", "flat_scan_mm_kernel"),
    (5000, 384, 5, 10, "f32", "flat_scan_mm_kernel"),    # 6 chunks of 256 bytes: one load per half stage
    (4099, 256, 8, 64, "f32", "flat_scan_mm_kernel"),    # last group holds three rows; k = 64 fills the per-wave list
    (5, 512, 3, 10, "f32", "flat_scan_mm_kernel"),       # fewer rows than one group, fewer than k
    (7000, 1024, 4, 100, "f16", "flat_scan_mm_kernel"),  # k = 100: two list registers per lane
    (9001, 256, 3, 10, "bf16", "flat_scan_mm_kernel"),   # 512-byte rows
    (6000, 256, 20, 10, "f32", "flat_scan_mm_kernel"),   # three query tiles (the shadow is off below)
    (5000, 768, 8, 10, "f32", "flat_scan_mm_kernel"),      # 3 KiB rows: the four-row form, two loads per half stage
    (4000, 1536, 3, 100, "f32", "flat_scan_mm_kernel"),    # k = 100 on fp32 rows under 8 KiB
    (4000, 2048, 3, 10, "bf16", "flat_scan_mm_kernel"),
    (4000, 1536, 1, 10, "f32", "flat_scan_mm_kernel"),     # one query on a row length K1's register-query form does not take
    (3000, 3072, 1, 100, "f32", "flat_scan_mm_kernel"),
])
def test_scan_mm_is_what_the_plan_picks(ssa, monkeypatch, n, d, nq, k, storage, kernel):
    """No kernel switches: the shapes plan_flat_scan gives to K1m by itself (csrc/plan.hip)."""
    monkeypatch.setenv("SSMI_MFMA_MIN_B", "100000")
    monkeypatch.setenv("SSMI_NO_SHADOW", "1")
    X, Q = unit_rows(n, d, 1234), unit_rows(nq, d, 4321)
    index = ssa.IndexFlatIP(d, storage=storage)
    index.add(X)
    D, I = index.search(Q, k)
    assert index.last_kernel() == kernel, index.last_kernel()
    Xr = _rounded(X, storage)
    Dr, Ir = flat_ip.search(Xr, Q, k)
    ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xr, Q, rtol=RTOL, tie_rtol=RTOL)
    assert ok, msg
    index.close()


@pytest.mark.parametrize("n,d,nq,k,storage", [
    (10000, 4096, 8, 10, "f32"),     # the shape the kernel was written for: 5-8 queries on fp32 rows
    (10000, 4096, 5, 10, "f32"),     # three of the eight query columns are copies of the last query
    (4099, 4096, 7, 64, "f32"),      # last row group holds three rows; k = 64 fills the per-wave list
    (3, 4096, 8, 10, "f32"),         # fewer rows than one group, fewer rows than k
    (1, 2048, 6, 1, "f32"),
    (6001, 1024, 8, 10, "f32"),      # 16 chunks per row: one turn of the two-stage loop per row group
    (5002, 512, 8, 10, "f32"),       # 8 chunks: the four-load stage
    (20000, 4096, 24, 10, "f32"),    # three query tiles (shadow off below)
    (10000, 4096, 4, 10, "f16"),     # 3-4 queries on 16-bit rows
    (10000, 4096, 3, 100, "f16"),    # k = 100: two list registers per lane
    (4097, 2048, 4, 10, "bf16"),
    (3001, 1024, 3, 10, "f16"),      # 8 chunks of 128 elements
])
def test_scan_mm_matches_oracle(ssa, monkeypatch, n, d, nq, k, storage):
    monkeypatch.setenv("SSMI_SCAN_MM", "1")
    monkeypatch.setenv("SSMI_NO_SHADOW", "1")
    monkeypatch.setenv("SSMI_MFMA_MIN_B", "100000")   # 16-bit rows: keep the batch on the exact scan
    X, Q = unit_rows(n, d, 1234), unit_rows(nq, d, 4321)
    index = ssa.IndexFlatIP(d, storage=storage)
    index.add(X)
    D, I = index.search(Q, k)
    assert index.last_kernel() == "flat_scan_mm_kernel", index.last_kernel()
    Xr = _rounded(X, storage)
    Dr, Ir = flat_ip.search(Xr, Q, k)
    ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xr, Q, rtol=RTOL, tie_rtol=RTOL)
    assert ok, msg
    index.close()


@pytest.mark.parametrize("d,kernel", [(1024, "flat_scan_mm_kernel"), (512, "flat_scan_mm_kernel")])
def test_scan_mm_over_segments_with_a_score_floor(ssa, monkeypatch, d, kernel):
    """Rows spread over many small segments (a row group never straddles two: segments hold a multiple of 256 rows),
    and a min_score floor applied in the epilogue."""
    monkeypatch.setenv("SSMI_NO_SHADOW", "1")
    monkeypatch.setenv("SSMI_SEG_SHIFT", "8")
    n, nq, k = 3000, 8, 20
    X, Q = unit_rows(n, d, 77), unit_rows(nq, d, 78)
    index = ssa.IndexFlatIP(d)
    for s in range(0, n, 700):
        index.add(X[s:s + 700])
    D, I = index.search(Q, k)
    assert index.last_kernel() == kernel
    Dr, Ir = flat_ip.search(X, Q, k)
    ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, X, Q, rtol=RTOL, tie_rtol=RTOL)
    assert ok, msg
    floor = float(np.sort(Dr[:, 4])[nq // 2])
    Dm, Im = index.search(Q, k, min_score=floor)
    for q in range(nq):
        keep = Dr[q] >= floor * (1 + 1e-6) if floor > 0 else Dr[q] >= floor * (1 - 1e-6)
        m = int(keep.sum())
        assert (Im[q, :m] >= 0).all() and np.allclose(Dm[q, :m], Dr[q, :m], rtol=RTOL)
        assert (Im[q, m + 1:] == -1).all()
    index.close()

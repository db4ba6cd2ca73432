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


# ---- the 256-query kernel (nq > 128) with many row tiles per slice: LDS ring running on across row tiles,
# ---- thresholds shared between the slices of a query tile, folds and the survivor path of the filter
@pytest.mark.parametrize("n,d,nq,k,seg_shift", [
    (140000, 256, 300, 10, None),   # 5 row tiles per slice, 8 phases per tile: ring runs on across tiles
    (140000, 256, 300, 100, None),  # same with two-register lists
    (50000, 192, 300, 10, None),    # 6 phases per tile: not whole turns of the ring, fill after the K loop
    (140003, 256, 600, 10, 10),     # 1024-row segments: every fourth tile boundary is a segment boundary; ragged tail
    (70000, 1024, 1100, 10, None),  # five query tiles
])
def test_wide_kernel_many_row_tiles(ssa, monkeypatch, n, d, nq, k, seg_shift):
    if seg_shift is not None:
        monkeypatch.setenv("SSMI_SEG_SHIFT", str(seg_shift))
    X, Q = unit_rows(n, d, 77), unit_rows(nq, d, 78)
    index = ssa.IndexFlatIP(d, storage="f16")
    index.add(X)
    D, I = index.search(Q, k)
    assert index.last_timing()["last_path"] == 1
    Xh, Qh = r16(X), r16(Q)
    Dr, Ir = flat_ip.search(Xh, Qh, k)
    ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xh, Qh, rtol=1e-5, tie_rtol=1e-5)
    assert ok, msg
    # the same search with every slice keeping its thresholds to itself
    monkeypatch.setenv("SSMI_MFMA_SHARE_THR", "0")
    index2 = ssa.IndexFlatIP(d, storage="f16")
    index2.add(X)
    D2, I2 = index2.search(Q, k)
    ok, msg, _ = flat_ip.check_against(D2, I2, Dr, Ir, Xh, Qh, rtol=1e-5, tie_rtol=1e-5)
    assert ok, msg
    # ... and with the gthr exchange only, without the bound from the slices' published tops
    monkeypatch.setenv("SSMI_MFMA_SHARE_THR", "1")
    monkeypatch.setenv("SSMI_MFMA_SHARE_TOPS", "0")
    D3, I3 = index.search(Q, k)
    ok, msg, _ = flat_ip.check_against(D3, I3, Dr, Ir, Xh, Qh, rtol=1e-5, tie_rtol=1e-5)
    assert ok, msg
    index.close(); index2.close()


@pytest.mark.parametrize("nq,k", [(300, 10), (300, 100), (64, 10), (128, 100), (1100, 50)])
def test_cross_slice_bound_keeps_ties_and_every_winner(ssa, nq, k):
    """The slices publish the best scores of their lists and filter with the k-th largest published score
    (mfma_common.h: tops_bound).  Copies of one row spread over all slices tie exactly AT that bound: the result
    must still be the k lowest ids among them, and a query whose winners all sit in one slice must lose none."""
    n, d = 150000, 512
    X = unit_rows(n, d, 91)
    Q = unit_rows(nq, d, 92)
    copies = np.arange(7, n, n // 173)[:173]          # 173 copies of Q[0], a few per slice
    X[copies] = Q[0]
    near = Q[1][None, :] + 0.05 * np.random.default_rng(3).standard_normal((k + 5, d)).astype(np.float32)
    X[5000:5000 + k + 5] = near / np.linalg.norm(near, axis=1, keepdims=True)   # winners of Q[1]: one slice only
    index = ssa.IndexFlatIP(d, storage="f16")
    index.add(X)
    Xh, Qh = r16(X), r16(Q)
    Dr, Ir = flat_ip.search(Xh, Qh, k)
    for _ in range(2):
        D, I = index.search(Q, k)
        ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xh, Qh, rtol=1e-5, tie_rtol=1e-5)
        assert ok, msg
        np.testing.assert_array_equal(I[0], copies[:k])
        assert set(I[1]) <= set(range(5000, 5000 + k + 5))
    index.close()


@pytest.mark.parametrize("k", [10, 100])
def test_wide_kernel_best_rows_in_one_slice(ssa, k):
    """All good rows sit in the first slice: the other slices inherit a threshold above every score they
    hold and hand empty or short lists to the merge."""
    n, d, nq = 140000, 256, 300
    rng = np.random.default_rng(5)
    u = unit_rows(1, d, 9)[0]
    X = unit_rows(n, d, 10)
    good = u[None, :] + 0.15 * rng.standard_normal((600, d)).astype(np.float32)
    X[:600] = good / np.linalg.norm(good, axis=1, keepdims=True)
    Q = u[None, :] + 0.05 * rng.standard_normal((nq, d)).astype(np.float32)
    Q = (Q / np.linalg.norm(Q, axis=1, keepdims=True)).astype(np.float32)
    index = ssa.IndexFlatIP(d, storage="f16")
    index.add(X)
    for _ in range(2):  # thresholds travel between workgroups: the result must not depend on their timing
        D, I = index.search(Q, k)
        assert index.last_timing()["last_path"] == 1
        Xh, Qh = r16(X), r16(Q)
        Dr, Ir = flat_ip.search(Xh, Qh, k)
        ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xh, Qh, rtol=1e-5, tie_rtol=1e-5)
        assert ok, msg
        assert (I[:, :min(k, 100)] < 600).all()
    index.close()


@pytest.mark.parametrize("n,d,nq", [(4096, 1024, 300), (5000, 512, 1000), (70001, 256, 260), (300, 4096, 513)])
def test_top1_batches_take_the_argmax_epilogue(ssa, n, d, nq):
    """k == 1 on the 256-query MFMA kernel (IVF centroid assignment, K5): the running-maximum epilogue must give the
    oracle's top-1 on the fp16-rounded operands - several slices, a ragged last row tile, more query tiles than CUs -
    and the lowest id among exactly equal rows."""
    import numpy as np
    from conftest import unit_rows
    from oracle import flat_ip
    X = unit_rows(n, d, 11)
    X[n // 2] = X[7]; X[n - 1] = X[7]                       # three identical rows: ties go to the lowest id
    Q = unit_rows(nq, d, 12)
    Q[5] = X[7]
    index = ssa.IndexFlatIP(d, storage="f16")
    index.add(X)
    D, I = index.search(Q, 1)
    assert index.last_kernel() == "flat_mfma_pp_kernel" and index.last_timing()["last_path"] == 1
    Xh, Qh = X.astype(np.float16).astype(np.float32), Q.astype(np.float16).astype(np.float32)
    Dr, Ir = flat_ip.search(Xh, Qh, 1)
    ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, Xh, Qh, rtol=1e-5, tie_rtol=1e-5)
    assert ok, msg
    assert I[5, 0] == 7
    D2, I2 = index.search(Q, 2)                             # and agrees with the first column of the general path
    np.testing.assert_array_equal(I2[:, 0], I[:, 0]); np.testing.assert_array_equal(D2[:, 0], D[:, 0])
    index.close()


# ---- the workgroups of an XCD meeting behind every row tile, and XCD patches of query tiles x slices (round 3): neither
# ---- may change a result.  Forced on here (the plan switches them on from 128 row tiles per slice only); the wait of a
# ---- meeting is bounded: with a bound of one poll every wait runs out (what happens when the workgroups are not all resident,
# ---- because another kernel holds CUs), each workgroup stops meeting after two of them, and the launch finishes all the same
@pytest.mark.parametrize("n,d,nq,k,patch,sync_from,polls", [
    (300000, 256, 4096, 10, "8", "0", "64"),     # 16 query tiles x 16 slices, 74 row tiles per slice: the bench's mapping
    (300000, 256, 4096, 100, "4", "17", "64"),   # k = 100 lists, meetings from tile 17 on
    (200000, 128, 2048, 10, "2", "0", "64"),     # 8 query tiles x 32 slices
    (150001, 256, 2100, 10, "0", "0", "64"),     # ragged last query tile and last slice shorter than the others (it pays its arrivals)
    (300000, 256, 4096, 10, "8", "0", "1"),      # every wait runs out at once
])
def test_xcd_meetings_and_patches_change_no_result(ssa, monkeypatch, n, d, nq, k, patch, sync_from, polls):
    X, Q = unit_rows(n, d, 21), unit_rows(nq, d, 22)
    index = ssa.IndexFlatIP(d, storage="f16")
    index.add(X)
    monkeypatch.setenv("SSMI_MFMA_SYNC", "0")
    monkeypatch.setenv("SSMI_MFMA_PATCH", "0")
    D0, I0 = index.search(Q, k)
    assert index.last_kernel() == "flat_mfma_pp_kernel"
    monkeypatch.setenv("SSMI_MFMA_SYNC", "1")
    monkeypatch.setenv("SSMI_MFMA_PATCH", patch)
    monkeypatch.setenv("SSMI_MFMA_SYNC_FROM", sync_from)
    monkeypatch.setenv("SSMI_MFMA_SYNC_POLLS", polls)
    for _ in range(2):                                     # twice: the arrival counters are zeroed per launch
        D1, I1 = index.search(Q, k)
        np.testing.assert_array_equal(I0, I1)
        np.testing.assert_array_equal(D0, D1)
    sel = np.arange(0, nq, max(1, nq // 64))               # and the oracle on a sample of the queries
    Xh, Qh = r16(X), r16(Q[sel])
    Dr, Ir = flat_ip.search(Xh, Qh, k)
    ok, msg, _ = flat_ip.check_against(D1[sel], I1[sel], Dr, Ir, Xh, Qh, rtol=1e-5, tie_rtol=1e-5)
    assert ok, msg
    index.close()


def test_kernel_clock_of_the_last_contraction(ssa):
    """ssmi_index_get_kernel_clock: the 256-query kernel stamps its own start and end (shader clock and the 100 MHz clock);
    the figure must be a plausible GPU clock, the elapsed time must agree with the hipEvent time of that launch, and a search
    that ran another kernel must say so."""
    d = 512
    index = ssa.IndexFlatIP(d, storage="f16")
    index.add(unit_rows(200000, d, 31))
    with pytest.raises(RuntimeError):
        index.last_kernel_clock()                      # no search yet
    index.search(unit_rows(600, d, 32), 10)
    assert index.last_kernel() == "flat_mfma_pp_kernel"
    c = index.last_kernel_clock()
    assert 500.0 < c["mhz"] < 3000.0 and c["cycles"] > 0
    ms = index.last_timing()["last_scan_ms"]
    assert 0.3 * ms < c["ms"] <= 1.1 * ms, (c, ms)     # workgroup 0 starts with the launch and ends with its own slice
    index.search(unit_rows(2, d, 33), 10)               # the streaming scan
    with pytest.raises(RuntimeError):
        index.last_kernel_clock()
    index.close()

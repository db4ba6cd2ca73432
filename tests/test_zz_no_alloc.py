"""ssmi_index_search_static (allocation-free Flat search, include/ssmi.h) against ssmi_index_search.

Both entry points run the same streaming scan + merge tail; the static one keeps the kernels' scratch
(partial k-lists, ticket counters) in a fixed per-GPU device array instead of the handle's workspaces.
Bar: bit-exact (ids AND scores) against the existing entry point, which the other GPU tests pin to the oracle;
additionally checked against the oracle here.  Two calls in a row share the array: the second one is smaller
and must not see the first one's leftovers.
"""
import ctypes

import numpy as np
import pytest

from conftest import SALT, unit_rows
from oracle import flat_ip

pytestmark = pytest.mark.gpu


def _pair(ssa, index, Qd, k, id_offset=0):
    import torch
    nq = Qd.shape[0]
    D0, I0 = index.search(Qd, k, id_offset=id_offset)
    D1 = torch.full((nq, k), 7.0, dtype=torch.float32, device=Qd.device)
    I1 = torch.full((nq, k), 7, dtype=torch.int64, device=Qd.device)
    index.search_static(Qd, k, out=(D1, I1), id_offset=id_offset)
    torch.cuda.synchronize()
    return D0.cpu().numpy(), I0.cpu().numpy(), D1.cpu().numpy(), I1.cpu().numpy()


@pytest.mark.parametrize("storage,d,n", [("f32", 4096, 20000), ("f16", 4096, 20000), ("f32", 1024, 5000), ("f32", 72, 3000)])
def test_static_search_equals_search_twice_in_a_row(ssa, storage, d, n):
    import torch
    X = unit_rows(n, d, 1234)
    Xs = X if storage == "f32" else X.astype(np.float16).astype(np.float32)
    index = ssa.IndexFlatIP(d, storage=storage)
    index.add(X)
    Q = unit_rows(4, d, 4321)
    Qd = torch.from_numpy(Q).cuda()
    # first call: 4 queries, k=100 (fills 4 x slices x 100 keys of the array)
    D0, I0, D1, I1 = _pair(ssa, index, Qd, 100, id_offset=5)
    assert index.last_timing()["last_path"] == 0
    np.testing.assert_array_equal(I1, I0)
    np.testing.assert_array_equal(D1, D0)
    Dr, Ir = flat_ip.search(Xs, Q, 100)
    ok, msg, _ = flat_ip.check_against(D1, I1 - 5, Dr, Ir, Xs, Q, rtol=1e-5, tie_rtol=1e-5)
    assert ok, msg
    # second call right behind it: smaller input (1 query, k=10), same array
    D0, I0, D1, I1 = _pair(ssa, index, Qd[2:3].contiguous(), 10)
    np.testing.assert_array_equal(I1, I0)
    np.testing.assert_array_equal(D1, D0)
    ok, msg, _ = flat_ip.check_against(D1, I1, Dr[2:3, :10], Ir[2:3, :10], Xs, Q[2:3], rtol=1e-5, tie_rtol=1e-5)
    assert ok, msg
    index.close()


def test_static_search_smaller_index_after_larger(ssa):
    """A second handle with fewer rows (fewer row slices) right after a larger one on the same GPU."""
    import torch
    d = 4096
    big, small = ssa.IndexFlatIP(d), ssa.IndexFlatIP(d)
    Xb, Xs = unit_rows(30000, d, 1), unit_rows(40, d, 2)
    big.add(Xb); small.add(Xs)
    Qd = torch.from_numpy(unit_rows(8, d, 3)).cuda()
    for index, nq, k in ((big, 8, 64), (small, 2, 64), (big, 1, 10), (small, 1, 1)):
        D0, I0, D1, I1 = _pair(ssa, index, Qd[:nq].contiguous(), k)
        np.testing.assert_array_equal(I1, I0)
        np.testing.assert_array_equal(D1, D0)
    # k beyond ntotal pads with -1 / -FLT_MAX exactly like the allocating entry point
    assert (I1 >= 0).all() and I1.shape == (1, 1)
    D0, I0, D1, I1 = _pair(ssa, small, Qd[:2].contiguous(), 64)
    assert (I1[:, 40:] == -1).all() and (D1[:, 40:] == flat_ip.NEG_MAX).all()


def test_static_search_empty_index_and_limits(ssa):
    import torch
    index = ssa.IndexFlatIP(256)
    Qd = torch.from_numpy(unit_rows(3, 256, 1)).cuda()
    D = torch.zeros((3, 5), dtype=torch.float32, device="cuda")
    I = torch.zeros((3, 5), dtype=torch.int64, device="cuda")
    index.search_static(Qd, 5, out=(D, I))
    torch.cuda.synchronize()
    assert (I.cpu().numpy() == -1).all() and (D.cpu().numpy() == flat_ip.NEG_MAX).all()
    index.add(unit_rows(200000, 256, 2))
    # beyond the static maximum: a clear error, nothing launched
    nq, k = 64, 1024
    Qb = torch.from_numpy(unit_rows(nq, 256, 3)).cuda()
    Db = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
    Ib = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
    with pytest.raises(RuntimeError, match="static scratch"):
        index.search_static(Qb, k, out=(Db, Ib))
    # padded dimensions need a staging copy: refused
    odd = ssa.IndexFlatIP(100)
    odd.add(unit_rows(50, 100, 4))
    with pytest.raises(RuntimeError, match="multiple of 8"):
        odd.search_static(torch.from_numpy(unit_rows(1, 100, 5)).cuda(), 5,
                          out=(torch.zeros((1, 5), device="cuda"), torch.zeros((1, 5), dtype=torch.int64, device="cuda")))
    out = ctypes.c_size_t()
    assert ssa._lib.load().ssmi_static_scratch_bytes(ctypes.byref(out)) == 0 and out.value == 8 << 20


def test_static_search_makes_no_device_allocation(ssa):
    """hipMalloc moves the driver's free-memory figure; the static entry point must leave it where it was."""
    import torch
    d = 4096
    index = ssa.IndexFlatIP(d)
    index.add(unit_rows(4096, d, 1))
    Qd = torch.from_numpy(unit_rows(4, d, 2)).cuda()
    D = torch.zeros((4, 50), dtype=torch.float32, device="cuda")
    I = torch.zeros((4, 50), dtype=torch.int64, device="cuda")
    index.search_static(Qd, 50, out=(D, I))  # first launch of a kernel loads its code object (driver memory)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        index.search_static(Qd, 50, out=(D, I))
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0


@pytest.mark.parametrize("seed", range(24))
def test_static_search_sweep(ssa, monkeypatch, seed):
    """Seeded shapes: both entry points on the streaming scan (the batched one is kept off the matrix cores) must agree
    bit for bit, call after call on the shared array; a request beyond the array is refused with a message."""
    import torch
    monkeypatch.setenv("SSMI_MFMA_MIN_B", "1000000")
    rng = np.random.default_rng(9000 + seed + SALT)
    d = int(rng.choice([8, 64, 256, 1000, 1024, 4096]))
    storage = str(rng.choice(["f32", "f16"]))
    n = int(rng.choice([1, 70, 257, 3000, 20000]))
    index = ssa.IndexFlatIP(d, storage=storage, fp16_shadow=False)
    index.add(unit_rows(n, d, 9100 + seed))
    for call in range(3):
        nq = int(rng.choice([1, 2, 3, 8, 9, 40]))
        k = int(rng.choice([1, 10, 64, 65, 300, 1024]))
        Qd = torch.from_numpy(unit_rows(nq, d, 9200 + seed * 7 + call)).cuda()
        try:
            D0, I0, D1, I1 = _pair(ssa, index, Qd, k, id_offset=int(rng.choice([0, 77])))
        except RuntimeError as e:
            assert "static scratch" in str(e), e     # nq x row slices x k keys beyond the 8 MiB array
            continue
        np.testing.assert_array_equal(I1, I0, err_msg=f"seed={seed} call={call} n={n} d={d} {storage} nq={nq} k={k}")
        np.testing.assert_array_equal(D1, D0, err_msg=f"seed={seed} call={call} n={n} d={d} {storage} nq={nq} k={k}")
    index.close()

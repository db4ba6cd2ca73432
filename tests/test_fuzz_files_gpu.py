"""Damaged index files: read_index (the stand-in of faiss.read_index, core/index_manager.py:185) must answer a
truncated, bit-flipped or size-inflated vectors.faiss - and a torn or foreign vectors.faiss.journal - with a
RuntimeError (what the reference sees from FAISS) or with a usable index, never with a crash, a hang or an
allocation of whatever a corrupt size field claims.  Seeded mutations of small Flat and IVF files."""
import shutil
import struct

import numpy as np
import pytest

from conftest import SALT, unit_rows

pytestmark = pytest.mark.gpu


def _make_files(ssa, tmp_path):
    d = 40
    X = unit_rows(900, d, 5)
    flat = ssa.IndexFlatIP(d); flat.add(X[:700])
    pf = tmp_path / "flat.faiss"; ssa.write_index(flat, pf)
    ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d), d, 9, ssa.METRIC_INNER_PRODUCT)
    ivf.set_centroids(unit_rows(9, d, 6)); ivf.add(X[:600])
    pi = tmp_path / "ivf.faiss"; ssa.write_index(ivf, pi)
    ivf.add(X[600:800]); ssa.write_index(ivf, pi, incremental=True)
    ivf.add(X[800:900]); ssa.write_index(ivf, pi, incremental=True)
    assert (tmp_path / "ivf.faiss.journal").exists()
    return d, X, pf, pi


def _usable(ssa, index, d):
    n = index.ntotal
    assert 0 <= n <= 10_000
    D, I = index.search(unit_rows(3, d, 1), 5)
    assert I.shape == (3, 5) and ((I >= -1) & (I < max(n, 1) + 10_000)).all()
    index.close()


@pytest.mark.parametrize("seed", range(8))
def test_damaged_files_raise_or_load(ssa, tmp_path, seed):
    rng = np.random.default_rng(6000 + seed + SALT)
    d, X, pf, pi = _make_files(ssa, tmp_path)
    outcomes = {"raised": 0, "loaded": 0}
    for trial in range(40):
        src = pf if rng.random() < 0.4 else pi
        victim = tmp_path / f"v{trial}.faiss"
        shutil.copy(src, victim)
        if src == pi and rng.random() < 0.7:
            shutil.copy(str(pi) + ".journal", str(victim) + ".journal")
        target = victim if (src == pf or rng.random() < 0.6 or not (tmp_path / f"v{trial}.faiss.journal").exists()) \
            else tmp_path / f"v{trial}.faiss.journal"
        raw = bytearray(target.read_bytes())
        kind = rng.choice(["truncate", "flip", "inflate", "garbage_tail", "zero_run"])
        if kind == "truncate":
            raw = raw[:int(rng.integers(0, len(raw)))]
        elif kind == "flip":
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(0, min(len(raw), 400)))     # headers, counts and the first list sizes
                raw[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == "inflate":   # an 8-byte field somewhere in the header region claims an absurd size
            pos = int(rng.integers(4, min(len(raw) - 8, 300)))
            raw[pos:pos + 8] = struct.pack("<Q", [1 << 40, 1 << 62, (1 << 64) - 1, 10_000_000_000][int(rng.integers(0, 4))])
        elif kind == "garbage_tail":
            raw += bytes(rng.integers(0, 256, int(rng.integers(1, 5000)), dtype=np.uint8))
        else:
            pos = int(rng.integers(0, max(1, len(raw) - 64)))
            raw[pos:pos + 64] = bytes(64)
        target.write_bytes(bytes(raw))
        try:
            index = ssa.read_index(victim)
        except RuntimeError as e:
            assert str(e), "an error without a message"
            outcomes["raised"] += 1
            continue
        _usable(ssa, index, d)
        outcomes["loaded"] += 1
    assert outcomes["raised"] > 0 and outcomes["raised"] + outcomes["loaded"] == 40, outcomes


def test_torn_and_foreign_journals_leave_the_base_index(ssa, tmp_path):
    d, X, pf, pi = _make_files(ssa, tmp_path)
    whole = ssa.read_index(pi)
    assert whole.ntotal == 900
    jp = str(pi) + ".journal"
    raw = open(jp, "rb").read()
    # a tail torn at any point keeps the complete records before it
    seen = set()
    for cut in list(range(0, 40)) + list(np.linspace(40, len(raw) - 1, 25).astype(int)):
        open(jp, "wb").write(raw[:cut])
        ix = ssa.read_index(pi)
        assert ix.ntotal in (600, 800, 900), (cut, ix.ntotal)
        seen.add(ix.ntotal)
        ix.close()
    assert seen == {600, 800}
    # the journal of another base file (other size) is ignored
    open(jp, "wb").write(raw)
    other = tmp_path / "other.faiss"
    ix2 = ssa.IndexIVFFlat(ssa.IndexFlatIP(d), d, 9, ssa.METRIC_INNER_PRODUCT)
    ix2.set_centroids(unit_rows(9, d, 6)); ix2.add(X[:50]); ssa.write_index(ix2, other)
    shutil.copy(jp, str(other) + ".journal")
    assert ssa.read_index(other).ntotal == 50


def test_nprobe_field_of_an_ivf_file(ssa, tmp_path):
    """IwFl header: fourcc, d:i32, ntotal:i64, two i64, is_trained:u8, metric:i32 = 37 bytes, then nlist:u64, nprobe:u64.
    Found by the salted sweep (SSMI_FUZZ_SALT=3, seed 4): a value whose low 32 bits are zero loaded as nprobe 0 and the
    first search failed in the coarse step.  Above nlist is legal (FAISS clamps at search time), zero reads as 1, a value
    beyond an int is damage."""
    d, X, pf, pi = _make_files(ssa, tmp_path)
    (tmp_path / "ivf.faiss.journal").unlink()
    raw = bytearray(pi.read_bytes())
    assert struct.unpack_from("<Q", raw, 37)[0] == 9
    Q = unit_rows(3, d, 1)
    exact = np.argsort(-(Q.astype(np.float64) @ X[:600].astype(np.float64).T), axis=1, kind="stable")[:, :5]
    for value, outcome in [(0, 1), (9 + 5, 14), (1 << 40, None), ((1 << 64) - 1, None), (1 << 31, None), ((1 << 31) - 1, (1 << 31) - 1)]:
        v = tmp_path / f"np{value}.faiss"
        struct.pack_into("<Q", raw, 45, value)
        v.write_bytes(bytes(raw))
        if outcome is None:
            with pytest.raises(RuntimeError, match="corrupt IVF header"):
                ssa.read_index(v)
            continue
        ix = ssa.read_index(v)
        assert ix.nprobe == outcome and ix.ntotal == 600
        D, I = ix.search(Q, 5)
        if outcome >= 9:      # every list probed: the exact result
            np.testing.assert_array_equal(I, exact)
        ix.close()


def test_append_after_a_torn_or_stale_journal_keeps_every_vector(ssa, tmp_path):
    """Incremental checkpoints behind a damaged journal (found by review, round 2): a record appended behind a torn
    tail was dropped or misread by the next read, and one appended to the journal of an older base file was ignored.
    Now the append starts at the last byte known good (torn tail cut off) or rewrites a foreign journal's header."""
    d, X, pf, pi = _make_files(ssa, tmp_path)        # base 600 rows + records of 200 and 100
    jp = str(pi) + ".journal"
    raw = open(jp, "rb").read()
    extra = unit_rows(50, d, 77)
    rec1_end = 32 + 12 + 200 * (4 + 8 + d * 4)        # header: fourcc, d:i32, nlist:u64, base ntotal:i64, base bytes:u64
    assert len(raw) == rec1_end + 12 + 100 * (4 + 8 + d * 4)
    for cut in (rec1_end + 5, rec1_end + 12 + 37, len(raw) - 1, 32 + 3, 11):
        open(jp, "wb").write(raw[:cut])
        ix = ssa.read_index(pi)
        kept = ix.ntotal
        assert kept == (800 if cut >= rec1_end else 600), (cut, kept)
        ix.add(extra)
        ssa.write_index(ix, pi, incremental=True)
        ix.close()
        back = ssa.read_index(pi)
        assert back.ntotal == kept + 50, (cut, back.ntotal)
        Dq, Iq = back.search(extra[:4], 1)
        np.testing.assert_array_equal(Iq[:, 0], np.arange(kept, kept + 4))
        back.add(extra); ssa.write_index(back, pi, incremental=True); back.close()   # and the reloaded handle goes on
        again = ssa.read_index(pi)
        assert again.ntotal == kept + 100
        again.close()
        # restore the starting point: base of 600 + the two records
        ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d), d, 9, ssa.METRIC_INNER_PRODUCT)
        ivf.set_centroids(unit_rows(9, d, 6)); ivf.add(X[:600]); ssa.write_index(ivf, pi)
        open(jp, "wb").write(raw)
        assert ssa.read_index(pi).ntotal == 900
    # a stale journal of an OLDER base (crash between the base rewrite and the removal of its journal)
    ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d), d, 9, ssa.METRIC_INNER_PRODUCT)
    ivf.set_centroids(unit_rows(9, d, 6)); ivf.add(X[:700]); ssa.write_index(ivf, pi); ivf.close()
    open(jp, "wb").write(raw)                        # header names the 600-row base
    ix = ssa.read_index(pi)
    assert ix.ntotal == 700
    ix.add(extra); ssa.write_index(ix, pi, incremental=True); ix.close()
    back = ssa.read_index(pi)
    assert back.ntotal == 750
    np.testing.assert_array_equal(back.search(extra[:4], 1)[1][:, 0], np.arange(700, 704))
    back.close()
    # the journal disappears under a live handle that already appended to it: the checkpoint rewrites the base
    ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d), d, 9, ssa.METRIC_INNER_PRODUCT)
    ivf.set_centroids(unit_rows(9, d, 6)); ivf.add(X[:600]); ssa.write_index(ivf, pi)
    ivf.add(X[600:700]); ssa.write_index(ivf, pi, incremental=True)
    (tmp_path / "ivf.faiss.journal").unlink()
    ivf.add(X[700:750]); ssa.write_index(ivf, pi, incremental=True)
    assert ssa.read_index(pi).ntotal == 750

"""GPU tests: write_index/read_index (FAISS file layout) and the SemanticIndex surface."""
import struct

import numpy as np
import pytest

from conftest import unit_rows
from oracle import flat_ip

pytestmark = pytest.mark.gpu


def faiss_header(d, ntotal, trained=True):
    return struct.pack("<iqqqBi", d, ntotal, 1 << 20, 1 << 20, 1 if trained else 0, 0)


def test_flat_write_read_roundtrip_and_layout(ssa, tmp_path):
    X = unit_rows(300, 72, 1)
    index = ssa.IndexFlatIP(72)
    index.add(X)
    p = tmp_path / "vectors.faiss"
    ssa.write_index(index, p)
    raw = p.read_bytes()
    # byte-for-byte the documented IndexFlatIP layout
    assert raw == b"IxFI" + faiss_header(72, 300) + struct.pack("<Q", 300 * 72) + X.tobytes()
    back = ssa.read_index(p)
    assert type(back).__name__ == "IndexFlatIP" and back.ntotal == 300 and back.d == 72
    np.testing.assert_array_equal(back.reconstruct_n(0, 300), X)
    Q = unit_rows(3, 72, 2)
    D0, I0 = index.search(Q, 10); D1, I1 = back.search(Q, 10)
    np.testing.assert_array_equal(I0, I1); np.testing.assert_array_equal(D0, D1)
    # an fp16-storage index round-trips exactly through the fp32 file
    h = ssa.IndexFlatIP(72, storage="f16"); h.add(X)
    ssa.write_index(h, tmp_path / "h.faiss")
    hb = ssa.read_index(tmp_path / "h.faiss", storage="f16")
    np.testing.assert_array_equal(hb.reconstruct_n(0, 300), X.astype(np.float16).astype(np.float32))


def test_read_independently_written_ivf_file(ssa, tmp_path):
    """A file assembled here with struct/numpy per the published FAISS layout (arbitrary ids, an
    empty list, 'full' size table) must load and search like the oracle says."""
    d, nlist, n = 64, 5, 200
    X, C = unit_rows(n, d, 3), unit_rows(nlist, d, 4)
    labels = np.argmax(X @ C.T, axis=1)
    labels[labels == 2] = 3                       # list 2 stays empty
    ids = np.arange(n, dtype=np.int64)
    buf = b"IwFl" + faiss_header(d, n) + struct.pack("<QQ", nlist, 2)
    buf += b"IxFI" + faiss_header(d, nlist) + struct.pack("<Q", nlist * d) + C.tobytes()
    buf += struct.pack("<BQ", 0, 0)
    buf += b"ilar" + struct.pack("<QQ", nlist, d * 4) + b"full" + struct.pack("<Q", nlist)
    buf += b"".join(struct.pack("<Q", int((labels == l).sum())) for l in range(nlist))
    for l in range(nlist):
        m = np.nonzero(labels == l)[0]
        if len(m):
            buf += X[m].tobytes() + ids[m].tobytes()
    p = tmp_path / "ivf.faiss"
    p.write_bytes(buf)
    index = ssa.read_index(p)
    assert type(index).__name__ == "IndexIVFFlat" and index.ntotal == n and index.nlist == nlist and index.nprobe == 2
    assert index.is_trained
    np.testing.assert_array_equal(index.assignments(), labels)
    np.testing.assert_array_equal(index.get_centroids(), C)
    from oracle import ivf_flat
    Q = unit_rows(4, d, 5)
    D, I = index.search(Q, 10)
    Dr, Ir, _ = ivf_flat.search(X, C, labels, Q, 10, 2)
    ok, msg, _ = flat_ip.check_against(D, I, Dr, Ir, X, Q, 1e-5, 1e-5)
    assert ok, msg
    # and our writer reproduces a file that reads back identically
    ssa.write_index(index, tmp_path / "again.faiss")
    again = ssa.read_index(tmp_path / "again.faiss")
    np.testing.assert_array_equal(again.assignments(), labels)
    D2, I2 = again.search(Q, 10)
    np.testing.assert_array_equal(I, I2)
    with pytest.raises(RuntimeError):
        ssa.read_index(tmp_path / "missing.faiss")
    (tmp_path / "bad.faiss").write_bytes(b"IHNf" + b"\0" * 64)
    with pytest.raises(RuntimeError, match="unsupported index type"):
        ssa.read_index(tmp_path / "bad.faiss")
    # corrupt or truncated files fail with a message (faiss.read_index raises; nothing may abort the process):
    # a list size of 2^60, sizes that do not add up to ntotal, a file cut inside a list, a flat payload count of 2^62
    sizes_at = buf.index(b"full") + 4 + 8
    for name, data in (
            ("huge_list", buf[:sizes_at] + struct.pack("<Q", 1 << 60) + buf[sizes_at + 8:]),
            ("bad_sum", buf[:sizes_at] + struct.pack("<Q", int((labels == 0).sum()) + 1) + buf[sizes_at + 8:]),
            ("cut", buf[:-100]),
            ("cut_in_sizes", buf[:sizes_at + 12]),
            ("flat_huge", b"IxFI" + faiss_header(d, 1 << 50) + struct.pack("<Q", (1 << 50) * d) + X.tobytes()),
            ("flat_cut", b"IxFI" + faiss_header(d, n) + struct.pack("<Q", n * d) + X.tobytes()[:-4])):
        (tmp_path / name).write_bytes(data)
        with pytest.raises(RuntimeError, match="read_index"):
            ssa.read_index(tmp_path / name)


def test_factory_rebuild_and_compact_on_device(ssa):
    F, C, T = ssa.IndexFactory, ssa.IndexConfig, ssa.IndexType
    X = unit_rows(1500, 128, 7)
    flat = F.create_index(C(T.FLAT, 128))
    flat.add(X)
    calls = []
    ivf, ok = F.rebuild_index(flat, C(T.IVF_FLAT, 128, nlist=12, nprobe=12), lambda c, t, s: calls.append((c, t)))
    assert ok and F.detect_index_type(ivf) is T.IVF_FLAT and ivf.ntotal == 1500 and ivf.nprobe == 12
    assert calls[0] == (0, 1500) and calls[-1] == (1500, 1500)
    Q = unit_rows(3, 128, 8)
    D, I = ivf.search(Q, 5); Dr, Ir = flat.search(Q, 5)
    np.testing.assert_array_equal(I, Ir)          # nprobe == nlist is exact
    same, ok = F.rebuild_index(ivf, C(T.FLAT, 128))
    assert not ok and same is ivf                 # IVF without a direct map cannot be reconstructed (as with FAISS)
    keep = [5, 3, 1499, 700]
    small, mapping, ok = F.compact_index(flat, keep)
    assert ok and mapping == {3: 0, 5: 1, 700: 2, 1499: 3} and small.ntotal == 4
    np.testing.assert_array_equal(small.reconstruct_n(0, 4), X[[3, 5, 700, 1499]])
    empty, mapping, ok = F.compact_index(flat, [])
    assert ok and mapping == {} and empty.ntotal == 0
    fresh, ok = F.rebuild_index(F.create_index(C(T.FLAT, 128)), C(T.IVF_FLAT, 128))
    assert ok and fresh.ntotal == 0 and not fresh.is_trained


def test_device_resident_rebuild_compact_and_exclusion_scores(ssa, monkeypatch):
    """rows move from the old index to the new one on the GPU (chunked device gather ->
    train/add with device tensors), and the exclusion filter's candidate x negative products are one device
    launch; both against numpy on the host copies the test holds."""
    F, C, T = ssa.IndexFactory, ssa.IndexConfig, ssa.IndexType
    d = 256
    X = unit_rows(5000, d, 71)
    for storage in ("f32", "f16"):
        Xs = X if storage == "f32" else X.astype(np.float16).astype(np.float32)
        flat = ssa.IndexFlatIP(d, storage=storage)
        flat.add(X)
        ids = np.array([0, 4999, 17, 17, 2500], dtype=np.int64)
        dev = flat.reconstruct_batch_device(ids)
        assert dev.is_cuda and dev.dtype.is_floating_point and tuple(dev.shape) == (5, d)
        np.testing.assert_array_equal(dev.cpu().numpy(), Xs[ids])
        np.testing.assert_array_equal(dev.cpu().numpy(), flat.reconstruct_batch(ids))
        neg = unit_rows(3, d, 72)
        sims = flat.score_ids(ids, neg)
        np.testing.assert_allclose(sims, Xs[ids] @ neg.T, rtol=1e-5, atol=1e-6)
        with pytest.raises(RuntimeError):
            flat.score_ids([5000], neg)
        # rebuild in chunks of 700 rows (several chunks, a ragged last one), Flat -> IVF with training on device rows
        monkeypatch.setattr(F, "_CHUNK_ROWS", 700)
        ivf, ok = F.rebuild_index(flat, C(T.IVF_FLAT, d, nlist=16, nprobe=16, extra={"storage": storage}))
        assert ok and ivf.ntotal == 5000 and ivf.is_trained and ivf.storage == storage
        Q = unit_rows(4, d, 73)
        D0, I0 = flat.search(Q, 10); D1, I1 = ivf.search(Q, 10)
        np.testing.assert_array_equal(I1, I0)          # nprobe == nlist: exact, ids kept their positions
        np.testing.assert_allclose(D1, D0, rtol=1e-6)
        ivf.make_direct_map()
        np.testing.assert_allclose(ivf.score_ids(ids, neg), Xs[ids] @ neg.T, rtol=1e-5, atol=1e-6)   # IVF: through the direct map
        keep = list(range(0, 5000, 3))
        small, mapping, ok = F.compact_index(ivf, keep, C(T.FLAT, d, extra={"storage": storage}))
        assert ok and small.ntotal == len(keep) and mapping[3] == 1 and mapping[4998] == len(keep) - 1
        np.testing.assert_array_equal(small.reconstruct_n(0, len(keep)), Xs[keep])
        for ix in (flat, ivf, small):
            ix.close()


def test_semantic_index_lifecycle(ssa, tmp_path):
    from semantic_search_amd.index_manager import SemanticIndex
    d = 96
    X = unit_rows(1200, d, 9)
    si = SemanticIndex("photos", embedding_dim=d, check_files_exist=False, index_type="ivf-flat", base_path=tmp_path)
    assert si.get_index_type() == "ivf_flat"
    ids = []
    for s in range(0, 1200, 16):                  # reference batches: 16 images (core/config.py:59)
        b = X[s:s + 16]
        ids += si.add_vectors(b, [f"/img/{s + i}.jpg" for i in range(len(b))], [f"/th/{s + i}.jpg" for i in range(len(b))],
                              [(1.0, 10)] * len(b))
    assert ids == list(range(1200))               # ids continue through the training buffer phase
    assert si._faiss_index.is_trained and si._faiss_index.ntotal == 1200   # trained at >= max(10*nlist, 1000)
    si._faiss_index.nprobe = 100
    res = si.search(X[17], top_k=5)
    assert res[0].vector_id == 17 and res[0].file_path == "/img/17.jpg" and abs(res[0].score - 1.0) < 1e-5
    batch = si.search_batch(X[[3, 900]], top_k=3)
    assert [r[0].vector_id for r in batch] == [3, 900]
    # the nodes' min_score filter (nodes/search_nodes.py:152-159), applied on the engine's sorted rows
    full = si.search(X[17], top_k=50)
    cut = float(full[9].score)
    some = si.search(X[17], top_k=50, min_score=cut)
    assert [r.vector_id for r in some] == [r.vector_id for r in full if r.score >= cut] and 10 <= len(some) < 50
    assert len(si.search(X[17], top_k=50, min_score=2.0)) == 0
    assert [len(r) for r in si.search_batch(X[[3, 900]], top_k=5, min_score=0.999)] == [1, 1]
    stats = si.get_stats()
    assert stats["vector_count"] == 1200 and stats["index_type"] == "ivf_flat" and stats["index_params"]["nlist"] == 100
    si.save()
    assert si.faiss_path.exists() and si.config_path.exists()
    # reload from disk
    si2 = SemanticIndex("photos", embedding_dim=4096, check_files_exist=False, base_path=tmp_path)
    assert si2.embedding_dim == d and si2.get_index_type() == "ivf_flat" and si2._faiss_index.ntotal == 1200
    # rebuild to flat fails without a direct map (reference behaviour), works after make_direct_map
    assert si2.rebuild("flat")["success"] is False
    si2._faiss_index.make_direct_map()
    out = si2.rebuild("flat")
    assert out == {"success": True, "old_type": "ivf_flat", "new_type": "flat", "old_count": 1200, "new_count": 1200}
    # delete two files, compact, ids are renumbered
    assert si2.db.remove_image("/img/0.jpg") and si2.db.remove_image("/img/5.jpg")
    assert si2.needs_compaction() == (False, 2, pytest.approx(2 / 1200 * 100))
    out = si2.compact()
    assert out["success"] and out["compacted"] and out["new_count"] == 1198 and out["removed_count"] == 2
    res = si2.search(X[17], top_k=1)
    assert res[0].file_path == "/img/17.jpg" and res[0].vector_id == 15
    # exclusion search drops candidates close to a negative vector
    res = si2.search_with_exclusion(X[17], [X[17]], top_k=5, exclusion_threshold=0.9)
    assert all(r.file_path != "/img/17.jpg" for r in res)
    vid = si2.add_document_vector(X[0], "/doc/a.pdf#1", "/th/a.jpg")
    assert vid == 1198
    si2.close(); si.db.close()


def test_training_buffer_survives_save_and_reload(ssa, tmp_path):
    """Additive over the reference: vectors buffered for IVF training are persisted by
    save() and restored on load, instead of being lost with the process (core/index_manager.py:288-289)."""
    from semantic_search_amd.index_manager import SemanticIndex
    d = 64
    X = unit_rows(1100, d, 19)
    si = SemanticIndex("buf", embedding_dim=d, check_files_exist=False, index_type="ivf_flat", base_path=tmp_path)
    ids = si.add_vectors(X[:600], [f"/i/{i}" for i in range(600)], [""] * 600, [(0.0, 0)] * 600)
    assert ids == list(range(600)) and si._faiss_index.ntotal == 0 and si._training_buffer_size == 600
    si.save()
    assert si.training_buffer_path.exists()
    si.close()
    si2 = SemanticIndex("buf", embedding_dim=d, check_files_exist=False, index_type="ivf_flat", base_path=tmp_path)
    assert si2._training_buffer_size == 600 and not si2._faiss_index.is_trained
    ids = si2.add_vectors(X[600:], [f"/i/{i}" for i in range(600, 1100)], [""] * 500, [(0.0, 0)] * 500)
    assert ids == list(range(600, 1100))
    assert si2._faiss_index.is_trained and si2._faiss_index.ntotal == 1100 and si2._training_buffer_size == 0
    si2._faiss_index.nprobe = 100
    assert si2.search(X[5], top_k=1)[0].vector_id == 5        # a vector buffered before the restart is searchable
    si2.save()
    assert not si2.training_buffer_path.exists()
    si2.close()


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_incremental_write_is_byte_identical_to_full_write(ssa, tmp_path, storage):
    """Checkpoints of a growing Flat index append the new rows; the file must equal a full rewrite at every step,
    and a file that is not an earlier state of the index (another index, a truncated file) must be rewritten."""
    d = 72
    X = unit_rows(700, d, 3)
    ix = ssa.IndexFlatIP(d, storage=storage)
    inc, full = tmp_path / "inc.faiss", tmp_path / "full.faiss"
    done = 0
    for step in (300, 1, 0, 250, 149):
        ix.add(X[done:done + step]); done += step
        ssa.write_index(ix, inc, incremental=True)
        ssa.write_index(ix, full)
        assert inc.read_bytes() == full.read_bytes(), f"after {done} rows"
    back = ssa.read_index(inc, storage=storage)
    assert back.ntotal == 700
    np.testing.assert_array_equal(back.reconstruct_n(0, 700), ix.reconstruct_n(0, 700))
    # another index's file of the same shape: the last-row check forces a full rewrite
    other = ssa.IndexFlatIP(d, storage=storage)
    other.add(unit_rows(100, d, 4))
    ssa.write_index(other, inc)
    ssa.write_index(ix, inc, incremental=True)
    assert inc.read_bytes() == full.read_bytes()
    # a truncated file
    inc.write_bytes(full.read_bytes()[:-10])
    ssa.write_index(ix, inc, incremental=True)
    assert inc.read_bytes() == full.read_bytes()
    # an IVF index takes the full-write path
    ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d), d, 4, ssa.METRIC_INNER_PRODUCT)
    ivf.set_centroids(unit_rows(4, d, 5)); ivf.add(X[:50])
    a, b = tmp_path / "a.faiss", tmp_path / "b.faiss"
    ssa.write_index(ivf, a, incremental=True); ssa.write_index(ivf, b)
    assert a.read_bytes() == b.read_bytes()


def test_ivf_incremental_checkpoints_journal(ssa, tmp_path):
    """SURVEY.md 8f rank 4 for IVF: a checkpoint of a grown IVF index appends the new members to
    vectors.faiss.journal instead of rewriting the base file (the reference rewrites everything every 10 batches,
    core/index_manager.py:859-860 -> :241); read_index replays the journal; a full write folds it back in."""
    import os
    d, nlist = 96, 12
    X = unit_rows(4000, d, 31)
    C = unit_rows(nlist, d, 32)
    ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d), d, nlist, ssa.METRIC_INNER_PRODUCT)
    ivf.set_centroids(C)
    ivf.nprobe = 4
    p, j = tmp_path / "vectors.faiss", tmp_path / "vectors.faiss.journal"
    ivf.add(X[:3000])
    ssa.write_index(ivf, p, incremental=True)                 # first checkpoint: full write
    base = p.read_bytes()
    assert not j.exists()
    jsizes = []
    done = 3000
    for step in (16, 16, 160, 1, 0, 300):                     # the reference adds 16 vectors per batch
        ivf.add(X[done:done + step]); done += step
        ssa.write_index(ivf, p, incremental=True)
        assert p.read_bytes() == base                         # the base file is not touched
        jsizes.append(os.path.getsize(j) if j.exists() else 0)
    rec = lambda n: 12 + n * (4 + 8 + d * 4)
    assert jsizes == list(np.cumsum([32 + rec(16), rec(16), rec(160), rec(1), 0, rec(300)]))
    Q = unit_rows(5, d, 33)
    D0, I0 = ivf.search(Q, 10)
    back = ssa.read_index(p)
    assert back.ntotal == done == ivf.ntotal and back.nprobe == 4
    np.testing.assert_array_equal(back.assignments(), ivf.assignments())
    D1, I1 = back.search(Q, 10)
    np.testing.assert_array_equal(I1, I0); np.testing.assert_array_equal(D1, D0)
    # the reloaded index goes on journaling where the file stands
    back.add(X[done:done + 50]); ivf.add(X[done:done + 50]); done += 50
    ssa.write_index(back, p, incremental=True)
    assert p.read_bytes() == base and os.path.getsize(j) == jsizes[-1] + rec(50)
    # a torn tail (crash while appending) costs only the torn record
    whole = j.read_bytes()
    j.write_bytes(whole[:-7])
    torn = ssa.read_index(p)
    assert torn.ntotal == done - 50
    j.write_bytes(whole)
    # a full write folds the journal in: same bytes as a full write of the index that never left memory
    full_a, full_b = tmp_path / "a.faiss", tmp_path / "b.faiss"
    again = ssa.read_index(p)
    ssa.write_index(again, full_a); ssa.write_index(ivf, full_b)
    assert full_a.read_bytes() == full_b.read_bytes()
    ssa.write_index(again, p)
    assert not j.exists() and p.read_bytes() == full_b.read_bytes()
    # a journal that belongs to another base file is ignored
    j.write_bytes(whole)
    assert ssa.read_index(p).ntotal == done
    # the journal is bounded: once it holds more than half the base (+64k rows) the checkpoint rewrites the base
    small = ssa.IndexIVFFlat(ssa.IndexFlatIP(d), d, nlist, ssa.METRIC_INNER_PRODUCT)
    small.set_centroids(C)
    small.add(X[:10])
    sp, sj = tmp_path / "s.faiss", tmp_path / "s.faiss.journal"
    ssa.write_index(small, sp, incremental=True)
    big = unit_rows(70000, d, 34)
    small.add(big)
    ssa.write_index(small, sp, incremental=True)
    assert not sj.exists() and ssa.read_index(sp).ntotal == 70010


def test_search_multi_index_merges_like_the_node(ssa, tmp_path):
    from semantic_search_amd.index_manager import SemanticIndex, search_multi_index
    d = 64
    X = unit_rows(600, d, 21)
    a = SemanticIndex("a", embedding_dim=d, check_files_exist=False, base_path=tmp_path)
    b = SemanticIndex("b", embedding_dim=d, check_files_exist=False, base_path=tmp_path)
    a.add_vectors(X[:300], [f"/a/{i}.jpg" for i in range(300)], [f"/ta/{i}.jpg" for i in range(300)], [(1.0, 1)] * 300)
    b.add_vectors(X[300:], [f"/b/{i}.jpg" for i in range(300)], [f"/tb/{i}.jpg" for i in range(300)], [(1.0, 1)] * 300)
    q = X[310]
    merged = search_multi_index([a, b, None], q, top_k=5, normalize_scores=False)
    want = SemanticIndex.merge_search_results([a.search(q, 50), b.search(q, 50)], top_k=5, normalize_scores=False)
    assert [(r.file_path, r.score) for r in merged] == [(r.file_path, r.score) for r in want]
    assert merged[0].file_path == "/b/10.jpg"
    assert len(search_multi_index([a, b], q, top_k=5, normalize_scores=False, min_score=0.99)) == 1
    a.close(); b.close()
    # by default hits whose file no longer exists are dropped, like the reference does (core/index_manager.py:389)
    real = tmp_path / "real.jpg"
    real.write_bytes(b"x")
    c = SemanticIndex("c", embedding_dim=d, base_path=tmp_path)
    c.add_vectors(X[:2], ["/gone/0.jpg", str(real)], ["", ""], [(0.0, 0)] * 2)
    assert [r.file_path for r in c.search(X[0], top_k=2)] == [str(real)]
    c.close()


def test_checkpoint_after_training_rewrites_the_base(ssa, tmp_path):
    """An IVF index checkpointed while untrained (the reference saves every 10 batches, also while it still buffers
    training vectors), trained later and checkpointed again must not leave the untrained base file with the members
    in a journal on top: the quantizer changed, so the incremental write becomes a full one."""
    d, nlist = 48, 6
    X = unit_rows(500, d, 77)
    ix = ssa.IndexIVFFlat(ssa.IndexFlatIP(d), d, nlist, ssa.METRIC_INNER_PRODUCT)
    p = tmp_path / "vectors.faiss"
    ssa.write_index(ix, p, incremental=True)              # untrained, empty
    back = ssa.read_index(p)
    assert not back.is_trained and back.ntotal == 0
    D, I = back.search(X[:3], 5)                          # FAISS: no list to probe -> no result, no exception
    assert (I == -1).all()
    ix.train(X); ix.add(X)
    ssa.write_index(ix, p, incremental=True)
    back = ssa.read_index(p)
    assert back.is_trained and back.ntotal == 500
    np.testing.assert_array_equal(back.get_centroids(), ix.get_centroids())
    ix.nprobe = back.nprobe = nlist
    D0, I0 = ix.search(X[:7], 10); D1, I1 = back.search(X[:7], 10)
    np.testing.assert_array_equal(I0, I1); np.testing.assert_array_equal(D0, D1)
    # replaced centroids invalidate the base as well
    ix2 = ssa.IndexIVFFlat(ssa.IndexFlatIP(d), d, nlist, ssa.METRIC_INNER_PRODUCT)
    ix2.set_centroids(unit_rows(nlist, d, 78)); ix2.add(X[:100])
    ssa.write_index(ix2, p)
    ix2.add(X[100:200]); ssa.write_index(ix2, p, incremental=True)
    assert ssa.read_index(p).ntotal == 200

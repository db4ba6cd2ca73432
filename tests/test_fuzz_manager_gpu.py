"""Stateful sweep over the SemanticIndex surface (the FAISS half of core/index_manager.py): random sequences of the
operations the reference's nodes issue - add_vectors in batches of 1..16 (and bulk), search / search_batch, deleting
files, save + reopen, compact, rebuild to the other index type - checked after every step against a plain numpy model
of what the index must hold (ids in insertion order, renumbered by compaction; hits of deleted files dropped after the
engine's top-k, as core/index_manager.py:376-396 does).  IVF indexes are searched with nprobe = nlist, where IVF-Flat is
exact, so one model serves both types; the buffering phase of an untrained IVF index returns nothing, as in the reference."""
import numpy as np
import pytest

from conftest import SALT, unit_rows

pytestmark = pytest.mark.gpu


def r16(a):
    return a.astype(np.float16).astype(np.float32)


class Model:
    def __init__(self, storage):
        self.storage = storage
        self.rows = []      # vectors in the engine, in id order (buffered ones included: they keep their ids)
        self.paths = []     # path of id i, None once its file was deleted
        self.serial = 0

    def stored(self):
        X = np.stack(self.rows) if self.rows else np.zeros((0, 1), np.float32)
        return r16(X) if self.storage == "f16" else X

    def expect(self, q, k, trained, min_score=0.0):
        if not trained or not self.rows:
            return [], []
        s = self.stored() @ q.astype(np.float32)
        order = np.lexsort((np.arange(len(s)), -s))[:k]          # engine's top-k: score desc, id asc
        keep = [i for i in order if self.paths[i] is not None and (min_score <= 0 or s[i] >= np.float32(min_score))]
        return keep, [float(s[i]) for i in keep]


def check_search(si, model, q, k, min_score=0.0):
    eng = si._faiss_index
    if hasattr(eng, "nprobe") and eng.is_trained:
        eng.nprobe = eng.nlist
    res = si.search(q, top_k=k, min_score=min_score)
    ids, scores = model.expect(q, k, eng.is_trained, min_score)
    got = [r.vector_id for r in res]
    if min_score > 0 and got != ids:   # a score within rounding of the floor may fall on either side
        s = dict(zip(*model.expect(q, k, eng.is_trained)))
        diff = set(got) ^ set(ids)
        assert all(abs(s[i] - min_score) < 1e-5 for i in diff), (got, ids)
        return
    if got != ids and len(got) == len(ids) and min_score <= 0:
        # two rows whose scores differ by less than the fp32 sums' rounding may come out in either order (or, at rank k,
        # either one): position by position the model's score of the returned id must equal that of the expected one
        full = model.stored() @ q.astype(np.float32)
        assert all(model.paths[g] is not None and abs(float(full[g]) - float(full[i])) <= 2e-6 for g, i in zip(got, ids)), \
            f"ids {got} != model {ids}"
        np.testing.assert_allclose([r.score for r in res], scores, rtol=1e-5, atol=1e-6)
        return
    assert got == ids, f"ids {got} != model {ids}"
    np.testing.assert_allclose([r.score for r in res], scores, rtol=1e-5, atol=1e-6)
    assert [r.file_path for r in res] == [model.paths[i] for i in ids]


@pytest.mark.parametrize("seed", range(32))
def test_semantic_index_random_walk(ssa, tmp_path, seed):
    from semantic_search_amd.index_manager import SemanticIndex
    rng = np.random.default_rng(8000 + seed + SALT)
    d = int(rng.choice([32, 96, 256]))
    storage = str(rng.choice(["f32", "f16"]))
    itype = str(rng.choice(["flat", "ivf_flat"]))
    pool = unit_rows(6000, d, 8100 + seed)
    model = Model(storage)
    log = []

    def open_index():
        return SemanticIndex("walk", embedding_dim=d, index_type=itype, base_path=tmp_path, storage=storage, check_files_exist=False)

    def add(si, n):
        vecs = pool[model.serial:model.serial + n]
        paths = [f"/img/{model.serial + i}.jpg" for i in range(n)]
        ids = si.add_vectors(vecs, paths, [p + ".th" for p in paths], [(1.0, 10)] * n)
        assert ids == list(range(len(model.rows), len(model.rows) + n)), f"ids {ids[:3]}.. after {len(model.rows)} rows"
        model.rows += list(vecs); model.paths += paths; model.serial += n

    si = open_index()
    try:
        for step in range(40):
            eng = si._faiss_index
            trained = eng.is_trained
            op = rng.choice(["add", "add", "add", "bulk", "search", "search", "batch", "floor", "delete", "reopen", "compact", "rebuild"])
            log.append(str(op))
            if op == "add":
                add(si, int(rng.choice([1, 2, 16, 16])))
            elif op == "bulk" and model.serial < 4500:
                add(si, int(rng.choice([300, 700, 1100])))
            elif op == "search" and model.rows:
                q = model.rows[int(rng.integers(len(model.rows)))] if rng.random() < 0.5 else unit_rows(1, d, int(rng.integers(1 << 30)))[0]
                check_search(si, model, q, int(rng.choice([1, 5, 10, 50])))
            elif op == "batch" and model.rows:
                if hasattr(eng, "nprobe") and trained:
                    eng.nprobe = eng.nlist
                Q = unit_rows(int(rng.choice([2, 7, 12])), d, int(rng.integers(1 << 30)))
                if storage == "f32" and len(Q) >= 9:
                    eng.fp16_shadow = False    # keep fp32 batches on the exact scan: the model is exact
                for q, res in zip(Q, si.search_batch(Q, top_k=8)):
                    ids, _ = model.expect(q, 8, trained)
                    if not (storage == "f16" and len(Q) >= 5):  # fp16 batches round the queries too: ids may swap at ties
                        assert [r.vector_id for r in res] == ids
                    else:
                        assert len(set(r.vector_id for r in res) & set(ids)) >= len(ids) - 1
            elif op == "floor" and model.rows and trained:
                q = model.rows[int(rng.integers(len(model.rows)))]
                check_search(si, model, q, 20, min_score=float(rng.choice([0.05, 0.2, 0.999])))
            elif op == "delete" and trained and any(p is not None for p in model.paths):
                alive = [i for i, p in enumerate(model.paths) if p is not None]
                i = int(rng.choice(alive))
                assert si.db.remove_image(model.paths[i])
                model.paths[i] = None
            elif op == "reopen":
                si.save()
                si.db.close()
                si = open_index()
                assert si._faiss_index.is_trained == trained
                assert si._faiss_index.ntotal == (len(model.rows) if trained else 0)
            elif op == "compact" and trained:
                if hasattr(eng, "make_direct_map"):
                    eng.make_direct_map()
                dead = sum(p is None for p in model.paths)
                out = si.compact()
                assert out["success"], out
                if dead:
                    assert out["compacted"] and out["new_count"] == len(model.rows) - dead
                    keep = [i for i, p in enumerate(model.paths) if p is not None]
                    model.rows = [model.rows[i] for i in keep]; model.paths = [model.paths[i] for i in keep]
                else:
                    assert not out["compacted"]
            elif op == "rebuild" and trained and model.rows:
                if hasattr(eng, "make_direct_map"):
                    eng.make_direct_map()
                target = "flat" if si.get_index_type() != "flat" else "ivf_flat"
                out = si.rebuild(target)
                assert out["success"] and out["new_count"] == len(model.rows), out
                assert si._faiss_index.is_trained
                # too few vectors to train nlist centroids: the reference falls back to Flat (core/index_factory.py:296-300)
                assert si.get_index_type() in (target, "flat")
            if model.rows and si._faiss_index.is_trained and step % 5 == 4:
                check_search(si, model, model.rows[-1], 3)
    except AssertionError as e:
        raise AssertionError(f"seed={seed} d={d} {storage} {itype} after {log}: {e}") from e
    finally:
        try:
            si.db.close()
        except Exception:
            pass

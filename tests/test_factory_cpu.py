"""Host logic of the index factory (reference core/index_factory.py surface), no GPU."""
import numpy as np
import pytest


class FakeIndex:
    """Duck-typed stand-in used only to exercise host logic without a device."""
    def __init__(self, d, n=0, trained=True):
        self.d, self.ntotal, self.is_trained = d, n, trained


class FakeIndexIVFFlat(FakeIndex):
    def __init__(self, d, nlist=100):
        super().__init__(d, 0, False)
        self.nlist, self.nprobe = nlist, 1

    def train(self, x):
        if len(x) < self.nlist:
            raise RuntimeError("not enough training points")
        self.is_trained = True


def test_index_type_from_string(ssa):
    T = ssa.IndexType
    assert T.from_string("flat") is T.FLAT and T.from_string("IVF-Flat") is T.IVF_FLAT
    assert T.from_string("ivf_pq") is T.IVF_PQ and T.from_string("HNSW") is T.HNSW
    assert T.from_string("ivf") is T.FLAT and T.from_string("") is T.FLAT  # unknown -> FLAT (index_factory.py:32-33)


def test_index_config_defaults_and_json_roundtrip(ssa):
    c = ssa.IndexConfig(index_type=ssa.IndexType.IVF_FLAT, dimension=2048)
    assert (c.nlist, c.nprobe, c.hnsw_m, c.hnsw_ef_construction, c.hnsw_ef_search, c.pq_m, c.pq_bits) == (100, 10, 32, 200, 64, 8, 8)
    d = c.to_dict()
    assert set(d) == {"index_type", "dimension", "nlist", "nprobe", "hnsw_m", "hnsw_ef_construction", "hnsw_ef_search",
                      "pq_m", "pq_bits", "is_trained", "training_size"}
    assert d["index_type"] == "ivf_flat" and ssa.IndexConfig.from_dict(d) == c
    e = ssa.IndexConfig.from_dict({})
    assert e.index_type is ssa.IndexType.FLAT and e.dimension == 4096 and e.nlist == 100


def test_descriptions_cover_every_type(ssa):
    for t in ssa.IndexType:
        desc = ssa.INDEX_TYPE_DESCRIPTIONS[t]
        assert {"name", "description", "needs_training", "supports_removal"} <= set(desc)
        assert desc["supports_removal"] is False
    assert ssa.INDEX_TYPE_DESCRIPTIONS[ssa.IndexType.IVF_FLAT]["needs_training"] is True
    assert ssa.INDEX_TYPE_DESCRIPTIONS[ssa.IndexType.FLAT]["needs_training"] is False


def test_min_training_size(ssa):
    F, C, T = ssa.IndexFactory, ssa.IndexConfig, ssa.IndexType
    assert F.get_min_training_size(C(T.FLAT, 64)) == 0
    assert F.get_min_training_size(C(T.IVF_FLAT, 64)) == 1000
    assert F.get_min_training_size(C(T.IVF_FLAT, 64, nlist=4096)) == 40960
    assert F.get_min_training_size(C(T.HNSW, 64)) == 0


def test_detect_type_and_stats_by_class_name(ssa):
    F, T = ssa.IndexFactory, ssa.IndexType
    assert F.detect_index_type(FakeIndex(8)) is T.FLAT
    assert F.detect_index_type(FakeIndexIVFFlat(8)) is T.IVF_FLAT
    assert F.detect_index_type(ssa.IndexFlatIP.__new__(ssa.IndexFlatIP)) is T.FLAT
    assert F.detect_index_type(ssa.IndexIVFFlat.__new__(ssa.IndexIVFFlat)) is T.IVF_FLAT
    s = F.get_index_stats(FakeIndexIVFFlat(8, nlist=7))
    assert s == {"total_vectors": 0, "dimension": 8, "index_type": "ivf_flat", "is_trained": False, "nlist": 7, "nprobe": 1}
    assert "nlist" not in F.get_index_stats(FakeIndex(8, 3))
    # the Flat class must not grow IVF attributes (hasattr drives stats/set_search_params)
    assert not any(hasattr(ssa.IndexFlatIP, a) for a in ("nlist", "nprobe", "hnsw"))
    assert all(hasattr(ssa.IndexIVFFlat, a) for a in ("nlist", "nprobe"))


def test_needs_training_train_and_search_params(ssa):
    F, C, T = ssa.IndexFactory, ssa.IndexConfig, ssa.IndexType
    assert F.needs_training(FakeIndex(8)) is False and F.needs_training(object()) is False
    ivf = FakeIndexIVFFlat(8, nlist=10)
    assert F.needs_training(ivf) is True
    assert F.train_index(ivf, np.zeros((3, 8))) is False      # exception swallowed -> False
    assert F.train_index(ivf, np.zeros((50, 8))) is True and F.train_index(ivf, np.zeros((1, 8))) is True
    F.set_search_params(ivf, C(T.IVF_FLAT, 8, nprobe=33))
    assert ivf.nprobe == 33
    F.set_search_params(FakeIndex(8), C(T.FLAT, 8))  # no attribute, no error


def test_out_of_scope_types_fail_loudly(ssa):
    F, C, T = ssa.IndexFactory, ssa.IndexConfig, ssa.IndexType
    for t in (T.HNSW, T.IVF_PQ):
        with pytest.raises(NotImplementedError):
            F.create_index(C(t, 64))
    with pytest.raises(ValueError):
        F.create_index(C("bogus", 64))


def test_approximate_types_map_to_flat_only_on_request(ssa, monkeypatch):
    """The opt-in of INTEGRATION.md section 3: no GPU is touched before the mapping is decided."""
    from semantic_search_amd import index_factory as m
    F, C, T = ssa.IndexFactory, ssa.IndexConfig, ssa.IndexType
    made = []
    monkeypatch.setattr(m.faiss, "IndexFlatIP", lambda d, **kw: made.append((d, kw)) or "flat-index")
    assert not m._approximate_as_flat(C(T.HNSW, 64))
    monkeypatch.setenv("SSMI_APPROX_AS_FLAT", "1")
    assert F.create_index(C(T.HNSW, 64)) == "flat-index" and F.create_index(C(T.IVF_PQ, 32)) == "flat-index"
    monkeypatch.delenv("SSMI_APPROX_AS_FLAT")
    cfg = C(T.HNSW, 16); cfg.extra["approx_as_flat"] = True
    assert F.create_index(cfg) == "flat-index"
    assert [d for d, _ in made] == [64, 32, 16]
    with pytest.raises(NotImplementedError):
        F.create_index(C(T.HNSW, 64))


def test_merge_search_results_normalises_dedups_and_sorts(ssa):
    """Mirror of SemanticIndex.merge_search_results (reference core/index_manager.py:635-703)."""
    from semantic_search_amd.index_manager import SearchResult, SearchResults, SemanticIndex
    a = SearchResults([SearchResult("x.jpg", "tx", 0.9, 1), SearchResult("y.jpg", "ty", 0.5, 2)], "text", "cats")
    b = SearchResults([SearchResult("y.jpg", "ty", 0.30, 7), SearchResult("z.jpg", "tz", 0.10, 8),
                       SearchResult("w.jpg", "tw", 0.20, 9)], "text", "dogs")
    m = SemanticIndex.merge_search_results([a, b], top_k=3)
    assert m.query_type == "multi_index" and m.query_info == "cats + dogs"
    # per-index min-max: a -> x 1.0, y 0.0; b -> y 1.0, w 0.5, z 0.0; y keeps its best (1.0, id 7)
    assert [(r.file_path, round(r.score, 6)) for r in m] == [("x.jpg", 1.0), ("y.jpg", 1.0), ("w.jpg", 0.5)]
    assert m[1].vector_id == 7
    raw = SemanticIndex.merge_search_results([a, b], top_k=10, normalize_scores=False)
    assert [r.file_path for r in raw] == ["x.jpg", "y.jpg", "w.jpg", "z.jpg"] and raw[1].score == 0.5
    assert len(SemanticIndex.merge_search_results([])) == 0
    one = SearchResults([SearchResult("q.jpg", "t", 0.4, 3)], "text", "")   # single score: range falls back to 1.0
    assert SemanticIndex.merge_search_results([one])[0].score == 0.0

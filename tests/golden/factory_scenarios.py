"""Scenarios that drive an index-factory module (IndexType / IndexConfig / IndexFactory / INDEX_TYPE_DESCRIPTIONS)
over a faiss-shaped engine and record everything the host logic decides: enum mapping, dict round trips, stats,
id mappings, progress-callback sequences, printed warnings, success flags, fall-backs, and the exact search result
of the index that comes out.

Run twice:
  * tests/golden/make_factory_golden.py (build container only) runs them over the REFERENCE's own
    core/index_factory.py with a numpy stand-in for `faiss` -> tests/golden/factory_golden.json;
  * tests/test_factory_golden_*.py replay them over semantic_search_amd.index_factory (+ the HIP engine for the
    scenarios that build indexes) and compare with that file.
This file is ours (no reference text); it only CALLS the factory surface, the way core/index_manager.py does
(:210, :284-308, :1326, :1406) and nodes/index_nodes.py does (:28, :97).
"""
import contextlib
import io
import re

import numpy as np


def rows(n, d, seed):
    rng = np.random.Generator(np.random.Philox(seed))
    x = rng.standard_normal((n, d), dtype=np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def _norm_print(text):
    """Printed lines, with the engine's own wording (after the reference's fixed prefix) cut off."""
    out = []
    for ln in text.splitlines():
        ln = re.sub(r"(Cannot reconstruct vectors): .*", r"\1: <engine message>", ln)
        ln = re.sub(r"(Index training failed): .*", r"\1: <engine message>", ln)
        out.append(ln)
    return out


class _Rec:
    def __init__(self):
        self.calls = []

    def __call__(self, cur, total, status):
        self.calls.append([int(cur), int(total), str(status)])


def _exact_topk(index, Q, k):
    """Search result of an index that must be exact (Flat, or IVF probing every list)."""
    if hasattr(index, "nlist"):
        index.nprobe = index.nlist
    D, I = index.search(Q, k)
    return np.asarray(I).tolist()


# ------------------------------------------------------------------------------------------------ host-only
def host_scenarios(F):
    T, C = F.IndexType, F.IndexConfig
    out = {}
    out["from_string"] = {s: T.from_string(s).value for s in
                          ["flat", "FLAT", "ivf_flat", "IVF-Flat", "ivf-flat", "hnsw", "HNSW", "ivf_pq", "IVF-PQ", "ivf", "", "bogus",
                           "Flat ", "ivfflat"]}
    out["enum_values"] = [m.value for m in T]
    cfg = C(index_type=T.IVF_FLAT, dimension=4096)
    out["config_defaults"] = cfg.to_dict()
    cfg2 = C(index_type=T.FLAT, dimension=256, nlist=7, nprobe=3, hnsw_m=5, hnsw_ef_construction=11, hnsw_ef_search=13,
             pq_m=4, pq_bits=6, is_trained=True, training_size=99)
    out["config_round_trip"] = C.from_dict(cfg2.to_dict()).to_dict()
    out["config_from_partial"] = [C.from_dict(d).to_dict() for d in
                                  [{}, {"index_type": "IVF-Flat"}, {"index_type": "nonsense", "dimension": 8, "nlist": 3},
                                   {"dimension": 2048, "nprobe": 64, "training_size": 5, "unknown_key": 1}]]
    out["min_training_size"] = {f"{t.value}/{nl}": F.IndexFactory.get_min_training_size(C(index_type=t, dimension=8, nlist=nl))
                                for t in T for nl in (1, 50, 100, 101, 4096)}
    out["descriptions"] = {t.value: {"keys": sorted(F.INDEX_TYPE_DESCRIPTIONS[t].keys()),
                                     "needs_training": F.INDEX_TYPE_DESCRIPTIONS[t]["needs_training"],
                                     "supports_removal": F.INDEX_TYPE_DESCRIPTIONS[t]["supports_removal"]} for t in T}
    return out


# ------------------------------------------------------------------------------------------------ with an engine
def engine_scenarios(F, fa):
    """fa: the faiss-shaped module the factory under test creates its indexes from (IndexFlatIP, IndexIVFFlat)."""
    T, C, IF = F.IndexType, F.IndexConfig, F.IndexFactory
    out = {}
    d = 16
    X = rows(2500, d, 1)
    Q = rows(6, d, 2)

    def run(fn):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            res = fn()
        res["printed"] = _norm_print(buf.getvalue())
        return res

    # --- create / stats / params / training gate (core/index_manager.py:210, 284-308; index_nodes.py get_stats)
    def s_create():
        r = {}
        flat = IF.create_index(C(index_type=T.FLAT, dimension=d))
        ivf = IF.create_index(C(index_type=T.IVF_FLAT, dimension=d, nlist=5, nprobe=3))
        r["types"] = [IF.detect_index_type(flat).value, IF.detect_index_type(ivf).value]
        r["needs_training"] = [IF.needs_training(flat), IF.needs_training(ivf), IF.needs_training(object())]
        r["stats_empty"] = [IF.get_index_stats(flat), IF.get_index_stats(ivf)]
        r["train_flat_is_true"] = IF.train_index(flat, X[:10])
        r["train_too_few"] = IF.train_index(ivf, X[:3])                  # fewer points than clusters -> False, swallowed
        r["still_untrained"] = IF.needs_training(ivf)
        r["train_ok"] = IF.train_index(ivf, X[:400].astype(np.float64))  # any float dtype is cast
        r["train_again"] = IF.train_index(ivf, X[:1])                    # trained: True without looking at the data
        flat.add(X[:100]); ivf.add(X[:300])
        r["stats_filled"] = [IF.get_index_stats(flat), IF.get_index_stats(ivf)]
        IF.set_search_params(ivf, C(index_type=T.IVF_FLAT, dimension=d, nprobe=4))
        IF.set_search_params(flat, C(index_type=T.FLAT, dimension=d, nprobe=4))      # no nprobe attribute: untouched
        r["nprobe_after_set"] = ivf.nprobe
        r["flat_has_no_nprobe"] = not hasattr(flat, "nprobe") and not hasattr(flat, "nlist") and not hasattr(flat, "hnsw")
        try:
            IF.create_index(C(index_type="not-a-type", dimension=d))
            r["unknown_type"] = "no error"
        except ValueError as e:
            r["unknown_type"] = "ValueError: " + str(e)
        return r
    out["create_stats_train"] = run(s_create)

    # --- rebuild_index (core/index_manager.py:1326)
    def s_rebuild(new_type, n, nlist=5, src="flat", direct_map=True, seed=11):
        def go():
            np.random.seed(seed)
            if src == "flat":
                old = fa.IndexFlatIP(d)
            else:
                old = fa.IndexIVFFlat(fa.IndexFlatIP(d), d, 4, fa.METRIC_INNER_PRODUCT)
                old.train(X[:500])
                if direct_map:
                    old.make_direct_map()
            if n:
                old.add(X[:n])
            cfg = C(index_type=new_type, dimension=d, nlist=nlist, nprobe=2)
            rec = _Rec()
            new, ok = IF.rebuild_index(old, cfg, rec)
            r = {"ok": ok, "same_object": new is old, "callbacks": rec.calls, "config_type_after": cfg.index_type.value,
                 "new_type": IF.detect_index_type(new).value, "ntotal": new.ntotal, "stats": IF.get_index_stats(new) if ok else None}
            if ok and n:
                r["top3"] = _exact_topk(new, Q, 3)
            return r
        return run(go)
    out["rebuild_empty"] = s_rebuild(T.IVF_FLAT, 0)
    out["rebuild_flat_to_flat"] = s_rebuild(T.FLAT, 700)
    out["rebuild_flat_to_ivf"] = s_rebuild(T.IVF_FLAT, 700)
    out["rebuild_ivf_to_flat"] = s_rebuild(T.FLAT, 700, src="ivf")
    out["rebuild_train_fails_falls_back"] = s_rebuild(T.IVF_FLAT, 40, nlist=100)        # 40 points, 100 clusters
    out["rebuild_unreconstructable"] = s_rebuild(T.FLAT, 300, src="ivf", direct_map=False)
    out["rebuild_odd_count"] = s_rebuild(T.FLAT, 7)                                       # n//2, 3n//4 rounding

    def s_no_callback():
        old = fa.IndexFlatIP(d)
        old.add(X[:5])
        new, ok = IF.rebuild_index(old, C(index_type=T.FLAT, dimension=d))
        _, mapping, ok2 = IF.compact_index(old, [4, 0])
        return {"ok": [ok, ok2], "ntotal": new.ntotal, "mapping": sorted([int(a), int(b)] for a, b in mapping.items())}
    out["no_callback"] = run(s_no_callback)

    # --- compact_index (core/index_manager.py:1406)
    def s_compact(n, valid, cfg_type=None, nlist=5, src="flat", direct_map=True, seed=12):
        def go():
            np.random.seed(seed)
            if src == "flat":
                old = fa.IndexFlatIP(d)
            else:
                old = fa.IndexIVFFlat(fa.IndexFlatIP(d), d, 4, fa.METRIC_INNER_PRODUCT)
                old.train(X[:500])
                if direct_map:
                    old.make_direct_map()
            if n:
                old.add(X[:n])
            cfg = None if cfg_type is None else C(index_type=cfg_type, dimension=d, nlist=nlist, nprobe=2)
            rec = _Rec()
            new, mapping, ok = IF.compact_index(old, list(valid), cfg, rec)
            r = {"ok": ok, "same_object": new is old, "callbacks": rec.calls,
                 "config_type_after": None if cfg is None else cfg.index_type.value,
                 "mapping": sorted([int(a), int(b)] for a, b in mapping.items()),
                 "new_type": IF.detect_index_type(new).value, "ntotal": new.ntotal}
            if ok and len(valid):
                r["top3"] = _exact_topk(new, Q, 3)
            return r
        return run(go)
    rng = np.random.default_rng(5)
    keep_a = rng.permutation(2500)[:2100].tolist()                      # unsorted on purpose; > 2 x 1000 vectors
    out["compact_flat_default_config"] = s_compact(2500, keep_a)
    out["compact_none_valid"] = s_compact(300, [])
    out["compact_none_valid_ivf_default"] = s_compact(300, [], src="ivf")
    out["compact_exactly_1000"] = s_compact(1200, list(range(100, 1100)))
    out["compact_flat_to_ivf"] = s_compact(900, list(range(0, 900, 2)), cfg_type=T.IVF_FLAT)
    out["compact_ivf_default_config"] = s_compact(900, list(range(5, 800)), src="ivf")     # detected type: ivf_flat, nlist 100
    out["compact_train_fails_falls_back"] = s_compact(300, list(range(0, 60)), cfg_type=T.IVF_FLAT, nlist=100)
    out["compact_unreconstructable"] = s_compact(300, list(range(0, 60)), src="ivf", direct_map=False)
    out["compact_id_out_of_range"] = s_compact(1500, list(range(0, 1400)) + [1500, 1501])    # fails after 1400 good rows
    out["compact_duplicates_kept"] = s_compact(50, [3, 3, 7, 1])
    return out


def sampling_scenario(F, fa):
    """More than 100000 vectors: rebuild/compact train on np.random.choice(n, 100000, replace=False)
    (core/index_factory.py:293-296, 401-403).  Recorded: how much of the global numpy RNG stream the call consumed
    (the next draw after it), and the outcome."""
    T, C, IF = F.IndexType, F.IndexConfig, F.IndexFactory
    d, n = 8, 100_300
    X = rows(n, d, 3)
    out = {}
    for name in ("rebuild", "compact"):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            np.random.seed(2024)
            old = fa.IndexFlatIP(d)
            old.add(X)
            cfg = C(index_type=T.IVF_FLAT, dimension=d, nlist=4, nprobe=4)
            rec = _Rec()
            if name == "rebuild":
                new, ok = IF.rebuild_index(old, cfg, rec)
            else:
                new, _, ok = IF.compact_index(old, list(range(n)), cfg, rec)
            nxt = int(np.random.randint(0, 1 << 30))
        out[name] = {"ok": ok, "ntotal": new.ntotal, "new_type": IF.detect_index_type(new).value,
                     "n_callbacks": len(rec.calls), "first_callbacks": rec.calls[:3], "last_callbacks": rec.calls[-3:],
                     "next_global_random_draw": nxt, "printed": _norm_print(buf.getvalue()),
                     "top1": _exact_topk(new, X[[0, 50_000, n - 1]], 1)}
    return out

"""SemanticIndex: the FAISS-touching half of the reference's core/index_manager.py,
running on the MI355X engine.

Scope: load/create, save, add_vectors (with the IVF
training buffer), add_document_vector, search, exclusion search on vectors,
rebuild, compact, needs_compaction, get_index_type, get_stats, close, and the
process-wide index cache.  Method names, arguments, return shapes and failure
behaviour follow the reference (cited per method as file:line under
/root/reference/core/index_manager.py).  Out of scope, by design: media ingestion
(index_folder & co), the embedding model, thumbnails; the SQLite schema of
core/database.py is reduced to the few id-bookkeeping calls this half needs
(`MetadataStore`).

Additive: `search_batch` (the reference discards all but row 0 of a search),
`storage=` / `device=` engine options, and a configurable base path (the
reference hard-codes H:/semantic_search, core/config.py:20).
"""
import json
import os
import sqlite3
import threading
from dataclasses import dataclass
from datetime import datetime
from pathlib import Path
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np

from . import faiss_compat as faiss
from .index_factory import IndexConfig, IndexFactory, IndexType

DEFAULT_TOP_K = 50          # core/config.py:63
MAX_TOP_K = 500             # core/config.py:64
DEFAULT_BATCH_SIZE = 16     # core/config.py:59
CHECKPOINT_INTERVAL = 10    # core/config.py:60


def get_base_path() -> Path:
    return Path(os.environ.get("SSMI_BASE_PATH", str(Path.home() / ".semantic_search")))


def get_index_path(index_name: str, base: Optional[Path] = None) -> Path:
    return (base or get_base_path()) / "indexes" / index_name          # core/config.py:143-150


@dataclass
class SearchResult:
    """core/index_manager.py:72-88"""
    file_path: str
    thumbnail_path: str
    score: float
    vector_id: int
    media_type: str = "image"

    def to_dict(self) -> Dict[str, Any]:
        return {"file_path": self.file_path, "thumbnail_path": self.thumbnail_path, "score": float(self.score),
                "vector_id": self.vector_id, "media_type": self.media_type}


@dataclass
class SearchResults:
    """core/index_manager.py:91-125"""
    results: List[SearchResult]
    query_type: str
    query_info: str

    def __len__(self) -> int:
        return len(self.results)

    def __iter__(self):
        return iter(self.results)

    def __getitem__(self, idx) -> SearchResult:
        return self.results[idx]

    def get_paths(self) -> List[str]:
        return [r.file_path for r in self.results]

    def get_scores(self) -> List[float]:
        return [r.score for r in self.results]

    def to_dict(self) -> Dict[str, Any]:
        return {"query_type": self.query_type, "query_info": self.query_info, "count": len(self.results),
                "results": [r.to_dict() for r in self.results]}


class MetadataStore:
    """vector_id <-> file bookkeeping: the subset of core/database.py this half calls
    (add_image :333, get_max_vector_id :325, get_images_by_vector_ids :381,
    get_deleted_vector_count :1373, get_all_vector_ids :1384, update_vector_ids :1423,
    clear_deleted_vectors :1426).  SQLite, one connection guarded by a lock."""

    def __init__(self, path: Path):
        self._conn = sqlite3.connect(str(path), check_same_thread=False)
        self._lock = threading.Lock()
        with self._lock, self._conn:
            self._conn.execute("CREATE TABLE IF NOT EXISTS images (file_path TEXT PRIMARY KEY, thumbnail_path TEXT, "
                               "vector_id INTEGER, media_type TEXT, file_mtime REAL, file_size INTEGER)")
            self._conn.execute("CREATE INDEX IF NOT EXISTS idx_vid ON images(vector_id)")
            self._conn.execute("CREATE TABLE IF NOT EXISTS deleted_vectors (vector_id INTEGER PRIMARY KEY)")

    def add_image(self, file_path, thumbnail_path, vector_id, media_type="image", file_mtime=0.0, file_size=0):
        with self._lock, self._conn:
            self._conn.execute("INSERT OR REPLACE INTO images VALUES (?,?,?,?,?,?)",
                               (str(file_path), str(thumbnail_path), int(vector_id), media_type, file_mtime, file_size))

    def remove_image(self, file_path) -> bool:
        with self._lock, self._conn:
            row = self._conn.execute("SELECT vector_id FROM images WHERE file_path=?", (str(file_path),)).fetchone()
            if row is None:
                return False
            self._conn.execute("DELETE FROM images WHERE file_path=?", (str(file_path),))
            self._conn.execute("INSERT OR IGNORE INTO deleted_vectors VALUES (?)", (row[0],))
            return True

    def get_max_vector_id(self) -> int:
        with self._lock:
            row = self._conn.execute("SELECT MAX(vector_id) FROM images").fetchone()
        return -1 if row[0] is None else int(row[0])

    def get_images_by_vector_ids(self, ids: List[int]) -> Dict[int, Tuple[str, str, str]]:
        out: Dict[int, Tuple[str, str, str]] = {}
        with self._lock:
            for s in range(0, len(ids), 500):
                chunk = [int(i) for i in ids[s:s + 500]]
                q = "SELECT vector_id, file_path, thumbnail_path, media_type FROM images WHERE vector_id IN (%s)" % ",".join("?" * len(chunk))
                for vid, fp, tp, mt in self._conn.execute(q, chunk):
                    out[int(vid)] = (fp, tp, mt)
        return out

    def get_deleted_vector_count(self) -> int:
        with self._lock:
            return int(self._conn.execute("SELECT COUNT(*) FROM deleted_vectors").fetchone()[0])

    def get_all_vector_ids(self) -> List[int]:
        with self._lock:
            return [int(r[0]) for r in self._conn.execute("SELECT vector_id FROM images ORDER BY vector_id")]

    def update_vector_ids(self, mapping: Dict[int, int]) -> None:
        # two passes through negative ids so that old and new id ranges may overlap (core/database.py:436-467)
        with self._lock, self._conn:
            for old, new in mapping.items():
                self._conn.execute("UPDATE images SET vector_id=? WHERE vector_id=?", (-(new + 1), old))
            self._conn.execute("UPDATE images SET vector_id = -vector_id - 1 WHERE vector_id < 0")

    def clear_deleted_vectors(self) -> None:
        with self._lock, self._conn:
            self._conn.execute("DELETE FROM deleted_vectors")

    def count(self) -> int:
        with self._lock:
            return int(self._conn.execute("SELECT COUNT(*) FROM images").fetchone()[0])

    def close(self) -> None:
        with self._lock:
            self._conn.close()


class SemanticIndex:
    """core/index_manager.py:128-1478 (FAISS half)."""

    def __init__(self, index_name: str, embedding_dim: int = 4096, index_type: str = "flat",
                 base_path: Optional[Path] = None, storage: str = "f32", device: Optional[int] = None,
                 check_files_exist: bool = True, devices: Optional[List[int]] = None):
        self.index_name = index_name
        self.embedding_dim = embedding_dim
        self.index_type_str = index_type
        self._engine_opts = {"storage": storage}
        if device is not None:
            self._engine_opts["device"] = device
        if devices is not None:
            # additive: ONE index over several GPUs inside this process (multi_device.py) - rows (Flat) or lists (IVF)
            # sharded, results merged on the first device; what the reference's single process can hold
            self._engine_opts.pop("device", None)
            self._engine_opts["devices"] = [int(v) for v in devices]
        # the reference drops hits whose file has disappeared since indexing (:389); False skips the stat per hit
        self._check_files_exist = check_files_exist
        self.index_path = get_index_path(index_name, Path(base_path) if base_path else None)
        self.faiss_path = self.index_path / "vectors.faiss"     # core/config.py:153-155
        self.config_path = self.index_path / "config.json"
        self.index_path.mkdir(parents=True, exist_ok=True)
        self.db = MetadataStore(self.index_path / "metadata.db")
        self._config = self._load_or_create_config()
        self._faiss_index = None
        self._index_config: Optional[IndexConfig] = None
        self._load_or_create_faiss()
        self._lock = threading.RLock()                          # :175
        self._training_buffer: List[np.ndarray] = []            # :178-179
        self._training_buffer_size = 0
        # Additive: in the reference the vectors buffered for IVF training live only in
        # memory (:288-289) although their metadata rows already exist, so they are lost when the process exits
        # before training; here save() persists them next to vectors.faiss and they are restored on load.
        self.training_buffer_path = self.index_path / "training_buffer.npy"
        if self.training_buffer_path.exists() and IndexFactory.needs_training(self._faiss_index):
            buf = np.load(self.training_buffer_path, allow_pickle=False)
            if buf.ndim == 2 and buf.shape[1] == self.embedding_dim and len(buf):
                self._training_buffer, self._training_buffer_size = [np.ascontiguousarray(buf, dtype=np.float32)], len(buf)

    # -- load / create / save -------------------------------------------------
    @property
    def _faiss_index(self):
        return self.__dict__.get("_engine")

    @_faiss_index.setter
    def _faiss_index(self, index) -> None:
        # every engine index bound to a SemanticIndex serves searches and is never timed: no per-search event
        # records (each costs the stream ~6 us)
        self.__dict__["_engine"] = index
        if index is not None and hasattr(index, "set_timing"):
            index.set_timing(False)

    def _load_or_create_faiss(self) -> None:
        """:181-214"""
        params = self._config.get("index_params", {})
        if self.faiss_path.exists():
            self._faiss_index = faiss.read_index(str(self.faiss_path), **self._engine_opts)
            self.embedding_dim = self._faiss_index.d
            self._index_config = IndexConfig(
                index_type=IndexFactory.detect_index_type(self._faiss_index), dimension=self.embedding_dim,
                nlist=params.get("nlist", 100), nprobe=params.get("nprobe", 10), hnsw_m=params.get("hnsw_m", 32),
                hnsw_ef_search=params.get("hnsw_ef_search", 64),
                is_trained=getattr(self._faiss_index, "is_trained", True), extra=dict(self._engine_opts))
        else:
            self._index_config = IndexConfig(index_type=IndexType.from_string(self.index_type_str),
                                             dimension=self.embedding_dim, extra=dict(self._engine_opts))
            self._faiss_index = IndexFactory.create_index(self._index_config)
            self._config["index_type"] = self._index_config.index_type.value
            self._config["index_params"] = self._index_config.to_dict()

    def _load_or_create_config(self) -> Dict[str, Any]:
        """:216-229"""
        if self.config_path.exists():
            return json.loads(self.config_path.read_text())
        config = {"index_name": self.index_name, "embedding_dim": self.embedding_dim,
                  "created_at": datetime.now().isoformat(), "model_used": None}
        self._save_config(config)
        return config

    def _save_config(self, config: Optional[Dict[str, Any]] = None) -> None:
        self.config_path.write_text(json.dumps(self._config if config is None else config, indent=2))

    def save(self) -> None:
        """:238-242"""
        with self._lock:
            # an index that only grew since the last checkpoint writes its new rows only: Flat appends them to the
            # file, IVF appends a record to vectors.faiss.journal (O(new rows) instead of O(N) per checkpoint)
            faiss.write_index(self._faiss_index, str(self.faiss_path), incremental=True)
            self._save_config()
            if self._training_buffer_size:
                np.save(self.training_buffer_path, np.vstack(self._training_buffer))
            elif self.training_buffer_path.exists():
                self.training_buffer_path.unlink()

    # -- add --------------------------------------------------------------------
    def add_vectors(self, vectors: np.ndarray, file_paths: List[str], thumbnail_paths: List[str],
                    file_infos: List[Tuple[float, int]], media_types: Optional[List[str]] = None) -> List[int]:
        """:244-346.  IVF indexes buffer vectors until get_min_training_size, then train and add
        them all; a failed training falls back to a Flat index holding the buffered vectors."""
        if len(vectors) == 0:
            return []
        media_types = media_types or ["image"] * len(vectors)
        vectors = np.ascontiguousarray(np.asarray(vectors).astype(np.float32))
        with self._lock:
            if IndexFactory.needs_training(self._faiss_index):
                self._training_buffer.append(vectors)
                self._training_buffer_size += len(vectors)
                if self._training_buffer_size >= IndexFactory.get_min_training_size(self._index_config):
                    all_vectors = np.vstack(self._training_buffer)
                    if IndexFactory.train_index(self._faiss_index, all_vectors):
                        self._faiss_index.add(all_vectors)
                        self._index_config.is_trained = True
                        self._index_config.training_size = len(all_vectors)
                        self._config["index_params"] = self._index_config.to_dict()
                    else:
                        self._index_config.index_type = IndexType.FLAT
                        self._faiss_index = IndexFactory.create_index(self._index_config)
                        self._faiss_index.add(all_vectors)
                    self._training_buffer, self._training_buffer_size = [], 0
            else:
                self._faiss_index.add(vectors)
            start_id = self._faiss_index.ntotal - len(vectors)
            if start_id < 0:  # still buffering: ids continue from the metadata store (:323-325)
                m = self.db.get_max_vector_id()
                start_id = m + 1 if m >= 0 else 0
            ids = []
            for i, (path, thumb, (mtime, size)) in enumerate(zip(file_paths, thumbnail_paths, file_infos)):
                self.db.add_image(path, thumb, start_id + i, media_types[i], mtime, size)
                ids.append(start_id + i)
            return ids

    def add_document_vector(self, vector: np.ndarray, file_path: str, thumbnail_path: str) -> int:
        """:1182-1189 (FAISS half): one vector, id = ntotal - 1, no training-buffer handling."""
        with self._lock:
            self._faiss_index.add(np.ascontiguousarray(np.asarray(vector, dtype=np.float32).reshape(1, -1)))
            vid = self._faiss_index.ntotal - 1
        self.db.add_image(file_path, thumbnail_path, vid, "document")
        return vid

    # -- search -----------------------------------------------------------------
    def _results_from(self, scores: np.ndarray, indices: np.ndarray, query_type: str) -> SearchResults:
        valid = indices >= 0                                    # :376-378
        scores, indices = scores[valid], indices[valid]
        records = self.db.get_images_by_vector_ids(indices.tolist())
        out = []
        for score, vid in zip(scores, indices):
            rec = records.get(int(vid))
            if rec is None:
                continue
            if self._check_files_exist and not Path(rec[0]).exists():   # :389
                continue
            out.append(SearchResult(rec[0], rec[1], float(score), int(vid), rec[2] or "image"))
        return SearchResults(out, query_type, "")

    @staticmethod
    def _floor(min_score: float) -> Optional[float]:
        """The search nodes' score filter (nodes/search_nodes.py:152-159: keep r.score >= min_score when
        min_score > 0) as the engine's score floor: applied inside the kernels (ssmi_index_search_min), so rows
        come back already cut and padded with -1, which _results_from drops like any short result."""
        return float(min_score) if (min_score is not None and min_score > 0.0) else None

    def search(self, query_vector: np.ndarray, top_k: int = DEFAULT_TOP_K, min_score: float = 0.0) -> SearchResults:
        """:348-398 (min_score is additive: SURVEY.md 8f rank 3)"""
        q = np.asarray(query_vector)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        q = np.ascontiguousarray(q.astype(np.float32))
        with self._lock:
            scores, indices = self._faiss_index.search(q, top_k, min_score=self._floor(min_score))
        return self._results_from(scores[0], indices[0], "vector")

    def search_batch(self, query_vectors: np.ndarray, top_k: int = DEFAULT_TOP_K, min_score: float = 0.0) -> List[SearchResults]:
        """Additive: every row of a (B, d) query matrix, one engine call (MFMA path for large B)."""
        q = np.ascontiguousarray(np.atleast_2d(np.asarray(query_vectors)).astype(np.float32))
        with self._lock:
            scores, indices = self._faiss_index.search(q, top_k, min_score=self._floor(min_score))
        return [self._results_from(s, i, "vector") for s, i in zip(scores, indices)]

    def search_with_exclusion(self, query_vector: np.ndarray, negative_vectors: List[np.ndarray], top_k: int = DEFAULT_TOP_K,
                              exclusion_threshold: float = 0.3) -> SearchResults:
        """Vector-level form of search_by_text_with_exclusion (:553-633): over-fetch
        min(3*top_k, 500) candidates, drop those whose stored vector has inner product
        >= threshold with any negative vector.  One device launch (ssmi_index_score_ids)
        replaces the per-candidate reconstruct() + np.dot calls (:606-621)."""
        candidates = self.search(query_vector, min(top_k * 3, MAX_TOP_K))
        if not candidates.results or not negative_vectors:
            return SearchResults(candidates.results[:top_k], "vector_with_exclusion", "")
        neg = np.stack([np.asarray(v, dtype=np.float32).reshape(-1) for v in negative_vectors])
        with self._lock:
            try:   # candidates x negatives inner products in one device launch (no vectors travel to the host)
                sims = self._faiss_index.score_ids([r.vector_id for r in candidates.results], neg)
            except RuntimeError:
                return SearchResults([], "vector_with_exclusion", "")   # :610-612 skips what it cannot reconstruct
        keep = ~(sims >= exclusion_threshold).any(axis=1)
        kept = [r for r, k in zip(candidates.results, keep) if k][:top_k]
        return SearchResults(kept, "vector_with_exclusion", "")

    @staticmethod
    def merge_search_results(results_list: List[SearchResults], top_k: int = DEFAULT_TOP_K,
                             normalize_scores: bool = True) -> SearchResults:
        """:635-703 (the consumer of nodes/search_nodes.py:869-881): merge the results of several
        indexes, optionally min-max normalising each index's scores to [0, 1], deduplicating by file
        path (highest score wins), best first."""
        if not results_list:
            return SearchResults(results=[], query_type="multi_index", query_info="")
        best: Dict[str, SearchResult] = {}
        infos = []
        for sr in results_list:
            if sr.query_info:
                infos.append(sr.query_info)
            if not sr.results:
                continue
            lo, span = 0.0, 1.0
            if normalize_scores:
                hi, lo = max(r.score for r in sr.results), min(r.score for r in sr.results)
                span = hi - lo if hi > lo else 1.0
            for r in sr.results:
                score = (r.score - lo) / span if normalize_scores else r.score
                if r.file_path not in best or score > best[r.file_path].score:
                    best[r.file_path] = SearchResult(file_path=r.file_path, thumbnail_path=r.thumbnail_path,
                                                     score=score, vector_id=r.vector_id, media_type=r.media_type)
        merged = sorted(best.values(), key=lambda x: x.score, reverse=True)[:top_k]
        return SearchResults(results=merged, query_type="multi_index",
                             query_info=" + ".join(infos) if infos else "multi-index search")

    # -- maintenance --------------------------------------------------------------
    def _new_config(self, new_type: IndexType) -> IndexConfig:
        p = self._config.get("index_params", {})
        return IndexConfig(index_type=new_type, dimension=self.embedding_dim, nlist=p.get("nlist", 100),
                           nprobe=p.get("nprobe", 10), hnsw_m=p.get("hnsw_m", 32),
                           hnsw_ef_search=p.get("hnsw_ef_search", 64), extra=dict(self._engine_opts))

    def _swap(self, new_index, new_config: IndexConfig) -> None:
        self._faiss_index, self._index_config = new_index, new_config
        self._config["index_type"] = new_config.index_type.value
        self._config["index_params"] = new_config.to_dict()
        self.save()

    def rebuild(self, new_index_type: Optional[str] = None,
                progress_callback: Optional[Callable[[int, int, str], None]] = None) -> Dict[str, Any]:
        """:1288-1350"""
        with self._lock:
            old_type = IndexFactory.detect_index_type(self._faiss_index)
            old_count = self._faiss_index.ntotal
            new_type = IndexType.from_string(new_index_type) if new_index_type else old_type
            new_config = self._new_config(new_type)
            new_index, ok = IndexFactory.rebuild_index(self._faiss_index, new_config, progress_callback)
            if not ok:
                return {"success": False, "error": "Rebuild failed - could not extract vectors from old index"}
            self._swap(new_index, new_config)
            return {"success": True, "old_type": old_type.value, "new_type": new_config.index_type.value,
                    "old_count": old_count, "new_count": new_index.ntotal}

    def compact(self, new_index_type: Optional[str] = None,
                progress_callback: Optional[Callable[[int, int, str], None]] = None) -> Dict[str, Any]:
        """:1352-1448"""
        with self._lock:
            old_count = self._faiss_index.ntotal
            if self.db.get_deleted_vector_count() == 0:
                return {"success": True, "compacted": False, "message": "No deleted vectors to compact",
                        "vector_count": old_count}
            valid = self.db.get_all_vector_ids()
            new_type = IndexType.from_string(new_index_type) if new_index_type else IndexFactory.detect_index_type(self._faiss_index)
            new_config = self._new_config(new_type)
            new_index, mapping, ok = IndexFactory.compact_index(self._faiss_index, valid, new_config, progress_callback)
            if not ok:
                return {"success": False, "error": "Compaction failed - could not extract vectors from index"}
            self.db.update_vector_ids(mapping)
            self.db.clear_deleted_vectors()
            self._swap(new_index, new_config)
            return {"success": True, "compacted": True, "old_count": old_count, "new_count": len(valid),
                    "removed_count": old_count - len(valid), "index_type": new_config.index_type.value}

    def needs_compaction(self) -> Tuple[bool, int, float]:
        """:1450-1468: recommended above 10 % wasted or 1000 deleted vectors."""
        deleted, total = self.db.get_deleted_vector_count(), self._faiss_index.ntotal
        if total == 0:
            return False, 0, 0.0
        wasted = deleted / total * 100
        return (wasted > 10.0 or deleted > 1000), deleted, wasted

    def get_index_type(self) -> str:
        return IndexFactory.detect_index_type(self._faiss_index).value

    def get_stats(self) -> Dict[str, Any]:
        """:1258-1286 (FAISS half)"""
        fs = IndexFactory.get_index_stats(self._faiss_index)
        needs, deleted, wasted = self.needs_compaction()
        return {"indexed_files": self.db.count(), "vector_count": self._faiss_index.ntotal,
                "embedding_dim": self.embedding_dim, "index_type": fs["index_type"], "is_trained": fs["is_trained"],
                "faiss_path": str(self.faiss_path),
                "faiss_size_mb": round(self.faiss_path.stat().st_size / 2 ** 20, 2) if self.faiss_path.exists() else 0,
                "model_used": self._config.get("model_used"), "created_at": self._config.get("created_at"),
                "index_params": fs, "deleted_vectors": deleted, "wasted_percentage": round(wasted, 1),
                "needs_compaction": needs}

    def close(self) -> None:
        """:1474-1477.  An IVF index that checkpointed incrementally leaves vectors.faiss.journal next to the base
        file; closing folds it back in with one full write, so the file at rest is a plain, complete FAISS file."""
        self.save()
        if Path(str(self.faiss_path) + ".journal").exists():
            with self._lock:
                faiss.write_index(self._faiss_index, str(self.faiss_path))
        self.db.close()
        self._faiss_index.close()


_index_cache: Dict[str, SemanticIndex] = {}     # :1481-1482
_index_lock = threading.Lock()


def search_multi_index(indexes: List["SemanticIndex"], query_vector: np.ndarray, top_k: int = DEFAULT_TOP_K,
                       normalize_scores: bool = True, min_score: float = 0.0) -> SearchResults:
    """Vector-level form of the SearchMultiIndex node (nodes/search_nodes.py:840-900): ONE encoded query searched in
    every index with max(top_k, 50) candidates each (:868), merged by SemanticIndex.merge_search_results (:882-886),
    then the min_score filter (:891-898).  The node re-encodes the text once per index (:871-878); here the caller
    encodes once and passes the vector."""
    per_index_k = max(top_k, 50)
    live = [ix for ix in indexes if ix is not None]
    merged = SemanticIndex.merge_search_results(_search_all(live, query_vector, per_index_k), top_k=top_k,
                                                normalize_scores=normalize_scores)
    if min_score > 0.0:
        merged = SearchResults([r for r in merged.results if r.score >= min_score], merged.query_type, merged.query_info)
    return merged


def _search_all(indexes: List["SemanticIndex"], query_vector: np.ndarray, k: int) -> List[SearchResults]:
    """One query against several indexes without a host round trip per index: the query goes to the device once,
    every index's search is enqueued on the current stream with device outputs, and all results come back in one
    synchronisation (the node searches its indexes one after the other, nodes/search_nodes.py:871-878).  Indexes
    on different GPUs, or an engine without device tensors, take the serial path."""
    import torch
    devs = {getattr(ix._faiss_index, "device", None) for ix in indexes}
    if len(indexes) < 2 or len(devs) != 1 or None in devs or not torch.cuda.is_available():
        return [ix.search(query_vector, k) for ix in indexes]
    q = torch.from_numpy(np.ascontiguousarray(np.asarray(query_vector, dtype=np.float32).reshape(1, -1))).to(
        torch.device("cuda", devs.pop()))
    outs = []
    for ix in indexes:
        with ix._lock:
            outs.append(ix._faiss_index.search(q, k))
    D = torch.stack([o[0][0] for o in outs]).cpu().numpy()     # one synchronising copy for all indexes
    I = torch.stack([o[1][0] for o in outs]).cpu().numpy()
    return [ix._results_from(D[j], I[j], "vector") for j, ix in enumerate(indexes)]


def get_or_create_index(index_name: str, embedding_dim: int = 4096, index_type: str = "flat", **kwargs) -> SemanticIndex:
    """:1485-1504"""
    with _index_lock:
        if index_name not in _index_cache:
            _index_cache[index_name] = SemanticIndex(index_name, embedding_dim, index_type, **kwargs)
        return _index_cache[index_name]


def drop_cached_index(index_name: str) -> None:
    with _index_lock:
        idx = _index_cache.pop(index_name, None)
    if idx is not None:
        idx.close()

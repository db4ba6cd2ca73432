"""Index factory: the reference's core/index_factory.py surface over the HIP engine.

Same names, arguments, defaults and failure behaviour as the reference module
(cited per member as file:line under /root/reference), with `faiss` replaced by
`semantic_search_amd.faiss_compat`.  Differences, all deliberate:
  - HNSW and IVF-PQ are outside the north-star scope (Flat + IVF-Flat).  The enum
    members and config fields stay (config.json compatibility); create_index
    raises NotImplementedError for them instead of shipping an untested path.
  - rebuild/compact read vectors back with ONE bulk gather (reconstruct_batch)
    instead of ntotal single-row calls (core/index_factory.py:273-274,371-372).
  - `storage` / `device` can be passed through IndexConfig.extra (additive).
"""
from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np

from . import faiss_compat as faiss


class IndexType(Enum):
    """core/index_factory.py:18-33"""
    FLAT = "flat"
    IVF_FLAT = "ivf_flat"
    HNSW = "hnsw"
    IVF_PQ = "ivf_pq"

    @classmethod
    def from_string(cls, s: str) -> "IndexType":
        """Case-insensitive, '-' == '_'; anything unknown means FLAT (core/index_factory.py:26-33)."""
        key = s.lower().replace("-", "_")
        return next((m for m in cls if m.value == key), cls.FLAT)


@dataclass
class IndexConfig:
    """core/index_factory.py:36-90 (field names and defaults are part of config.json)."""
    index_type: IndexType
    dimension: int
    nlist: int = 100
    nprobe: int = 10
    hnsw_m: int = 32
    hnsw_ef_construction: int = 200
    hnsw_ef_search: int = 64
    pq_m: int = 8
    pq_bits: int = 8
    is_trained: bool = False
    training_size: int = 0
    # additive, not serialised: engine options {"storage": "f32"|"f16"|"bf16", "device": int, "devices": [int, ...]}
    extra: Dict[str, Any] = field(default_factory=dict, repr=False, compare=False)

    _KEYS = ("nlist", "nprobe", "hnsw_m", "hnsw_ef_construction", "hnsw_ef_search", "pq_m", "pq_bits",
             "is_trained", "training_size")

    def to_dict(self) -> Dict[str, Any]:
        out: Dict[str, Any] = {"index_type": self.index_type.value, "dimension": self.dimension}
        out.update({k: getattr(self, k) for k in self._KEYS})
        return out

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "IndexConfig":
        defaults = cls(index_type=IndexType.FLAT, dimension=4096)
        kwargs = {k: d.get(k, getattr(defaults, k)) for k in cls._KEYS}
        return cls(index_type=IndexType.from_string(d.get("index_type", "flat")),
                   dimension=d.get("dimension", 4096), **kwargs)


# core/index_factory.py:94-119 (imported by nodes/index_nodes.py:28)
INDEX_TYPE_DESCRIPTIONS = {
    IndexType.FLAT: {
        "name": "Flat (Exact)",
        "description": "Exact search on the GPU (every vector is scored). Batches of up to 8 queries use exact fp32 products; larger batches run on the matrix cores with fp16-rounded operands (scores within ~1e-4, near-ties may swap; set index.fp16_shadow = False to keep them exact). HBM-bandwidth bound; fine up to tens of millions of vectors per MI355X.",
        "needs_training": False,
        "supports_removal": False,
    },
    IndexType.IVF_FLAT: {
        "name": "IVF-Flat (Fast)",
        "description": "Approximate search using clustering: probes nprobe of nlist inverted lists. Needs training.",
        "needs_training": True,
        "supports_removal": False,
    },
    IndexType.HNSW: {
        "name": "HNSW (not built)",
        "description": "Graph index; outside the scope of the MI355X engine (use Flat or IVF-Flat).",
        "needs_training": False,
        "supports_removal": False,
    },
    IndexType.IVF_PQ: {
        "name": "IVF-PQ (not built)",
        "description": "Product-quantised index; outside the scope of the MI355X engine (use IVF-Flat with f16 storage).",
        "needs_training": True,
        "supports_removal": False,
    },
}

ProgressCallback = Optional[Callable[[int, int, str], None]]
_TRAIN_SAMPLE_CAP = 100000  # core/index_factory.py:293,401


def _approximate_as_flat(config) -> bool:
    """HNSW / IVF-PQ are outside this engine's scope and fail loudly by default; `IndexConfig.extra["approx_as_flat"]` or
    SSMI_APPROX_AS_FLAT=1 maps them to the exact Flat index instead."""
    import os
    return bool(config.extra.get("approx_as_flat")) or os.environ.get("SSMI_APPROX_AS_FLAT") == "1"


class IndexFactory:
    """core/index_factory.py:122-416"""

    @staticmethod
    def create_index(config: IndexConfig):
        """core/index_factory.py:125-163"""
        opts = {k: v for k, v in config.extra.items() if k in ("storage", "device", "devices")}
        if config.index_type == IndexType.FLAT:
            return faiss.IndexFlatIP(config.dimension, **opts)
        if config.index_type == IndexType.IVF_FLAT:
            # (one index over several GPUs keeps its coarse quantizer inside every shard: no separate object)
            quantizer = None if "devices" in opts else faiss.IndexFlatIP(config.dimension, **opts)
            index = faiss.IndexIVFFlat(quantizer, config.dimension, config.nlist, faiss.METRIC_INNER_PRODUCT, **opts)
            index.nprobe = config.nprobe
            return index
        if config.index_type in (IndexType.HNSW, IndexType.IVF_PQ) and _approximate_as_flat(config):
            # opt-in for a switched-over installation whose UI still offers "hnsw" (nodes/index_nodes.py:34-44): the exact
            # Flat index returns a superset-quality answer for either approximate type (recall 1.0); the object says what
            # it is (IndexFlatIP), so detect_index_type and config.json record "flat" from the next save on
            print(f"[SemanticSearch] index type {config.index_type.value!r} is served by the exact Flat index (SSMI_APPROX_AS_FLAT)")
            return faiss.IndexFlatIP(config.dimension, **opts)
        if config.index_type in (IndexType.HNSW, IndexType.IVF_PQ):
            raise NotImplementedError(
                f"index type {config.index_type.value!r} is outside the MI355X engine's scope (flat, ivf_flat)")
        raise ValueError(f"Unknown index type: {config.index_type}")

    @staticmethod
    def needs_training(index) -> bool:
        """core/index_factory.py:165-171"""
        return (not index.is_trained) if hasattr(index, "is_trained") else False

    @staticmethod
    def train_index(index, vectors: np.ndarray) -> bool:
        """core/index_factory.py:173-195: True if already trained; any exception means False."""
        if not IndexFactory.needs_training(index):
            return True
        try:
            index.train(np.ascontiguousarray(vectors.astype(np.float32)))
            return bool(index.is_trained)
        except Exception as e:  # the reference swallows everything here
            print(f"[SemanticSearch] Index training failed: {e}")
            return False

    @staticmethod
    def get_min_training_size(config: IndexConfig) -> int:
        """core/index_factory.py:197-203"""
        if config.index_type in (IndexType.IVF_FLAT, IndexType.IVF_PQ):
            return max(config.nlist * 10, 1000)
        return 0

    @staticmethod
    def detect_index_type(index) -> IndexType:
        """core/index_factory.py:205-217 (class-name sniffing, same precedence)."""
        name = type(index).__name__
        for needle, kind in (("IVFPQ", IndexType.IVF_PQ), ("IVFFlat", IndexType.IVF_FLAT), ("HNSW", IndexType.HNSW)):
            if needle in name:
                return kind
        return IndexType.FLAT

    @staticmethod
    def get_index_stats(index) -> Dict[str, Any]:
        """core/index_factory.py:219-238"""
        stats: Dict[str, Any] = {
            "total_vectors": index.ntotal,
            "dimension": index.d,
            "index_type": IndexFactory.detect_index_type(index).value,
            "is_trained": getattr(index, "is_trained", True),
        }
        for attr in ("nlist", "nprobe"):
            if hasattr(index, attr):
                stats[attr] = getattr(index, attr)
        if hasattr(index, "hnsw"):
            stats["hnsw_m"] = getattr(index.hnsw, "M", None)
            stats["hnsw_ef_search"] = index.hnsw.efSearch
        return stats

    @staticmethod
    def set_search_params(index, config: IndexConfig) -> None:
        """core/index_factory.py:314-320"""
        if hasattr(index, "nprobe"):
            index.nprobe = config.nprobe
        if hasattr(index, "hnsw") and hasattr(index.hnsw, "efSearch"):
            index.hnsw.efSearch = config.hnsw_ef_search

    # -- shared tail of rebuild/compact -------------------------------------
    _CHUNK_ROWS = 65536  # rows moved per step (1 GiB of fp32 at d = 4096), all of it in HBM

    @staticmethod
    def _first_unreadable(old_index, ids: np.ndarray) -> int:
        """Position in `ids` of the first vector old_index.reconstruct would refuse (len(ids) if none): the reference
        walks the ids one reconstruct() at a time and stops at the first RuntimeError (core/index_factory.py:271-278,
        369-378); here the vectors move later, on the device, so the same answer is worked out up front - an IVF index
        without a direct map refuses every id, any index refuses ids outside [0, ntotal)."""
        if len(ids) == 0:
            return 0
        bad = np.nonzero((ids < 0) | (ids >= old_index.ntotal))[0]
        first_bad = int(bad[0]) if len(bad) else len(ids)
        if first_bad > 0:
            try:
                old_index.reconstruct_batch_device(ids[:1])
            except RuntimeError:
                return 0
        return first_bad

    @staticmethod
    def _train_and_fill(index, old_index, ids: np.ndarray, config: IndexConfig, report, n_done: int, status: Optional[str]):
        """Shared tail of rebuild_index / compact_index (core/index_factory.py:286-307, 394-411): train on all vectors
        or on a sample of 100000 drawn like the reference draws it (np.random.choice without replacement, used in the
        order drawn), fall back to Flat when training fails, add.  DEVICE-RESIDENT: rows are gathered on the GPU
        (ssmi_index_reconstruct_n_dev) and handed to train()/add() as device tensors chunk by chunk, so no (N, d) host
        array exists at any point."""
        n = len(ids)
        if IndexFactory.needs_training(index):
            report(n_done, n, status)
            sample = ids
            if n > _TRAIN_SAMPLE_CAP:
                sample = ids[np.random.choice(n, _TRAIN_SAMPLE_CAP, replace=False)]
            ok = False
            try:
                index.train(old_index.reconstruct_batch_device(sample))
                ok = bool(index.is_trained)
            except Exception as e:  # the reference's train_index swallows everything (core/index_factory.py:193-195)
                print(f"[SemanticSearch] Index training failed: {e}")
            if not ok:
                print("[SemanticSearch] Warning: Training failed, falling back to Flat index")
                config.index_type = IndexType.FLAT
                index = IndexFactory.create_index(config)
        return index

    @staticmethod
    def _fill(index, old_index, ids: np.ndarray) -> None:
        for s0 in range(0, len(ids), IndexFactory._CHUNK_ROWS):
            index.add(old_index.reconstruct_batch_device(ids[s0:s0 + IndexFactory._CHUNK_ROWS]))

    @staticmethod
    def rebuild_index(old_index, new_config: IndexConfig, progress_callback: ProgressCallback = None) -> Tuple[Any, bool]:
        """core/index_factory.py:240-312.  Returns (new_index, True) or (old_index, False); the progress callback sees
        the reference's sequence: (0, "Extracting..."), (n/2, "Creating new index..."), (n/2, "Training new index...")
        only when the index just created asks for training, (3n/4, "Adding..."), (n, "Rebuild complete")."""
        n = old_index.ntotal
        report = progress_callback or (lambda *a: None)
        if n == 0:
            return IndexFactory.create_index(new_config), True
        report(0, n, "Extracting vectors from old index...")
        ids = np.arange(n, dtype=np.int64)
        if IndexFactory._first_unreadable(old_index, ids) < n:
            print("[SemanticSearch] Warning: Cannot reconstruct vectors from this index type")
            return old_index, False
        report(n // 2, n, "Creating new index...")
        new_index = IndexFactory.create_index(new_config)
        new_index = IndexFactory._train_and_fill(new_index, old_index, ids, new_config, report, n // 2, "Training new index...")
        report(n * 3 // 4, n, "Adding vectors to new index...")
        IndexFactory._fill(new_index, old_index, ids)
        report(n, n, "Rebuild complete")
        return new_index, True

    @staticmethod
    def compact_index(old_index, valid_vector_ids: List[int], new_config: Optional[IndexConfig] = None,
                      progress_callback: ProgressCallback = None) -> Tuple[Any, Dict[int, int], bool]:
        """core/index_factory.py:322-416.  Keeps sorted(valid ids), renumbers them 0..n-1; the progress callback sees
        the reference's sequence, the "Extracted i/n vectors..." report after every 1000 vectors included."""
        n_valid, n_total = len(valid_vector_ids), old_index.ntotal
        report = progress_callback or (lambda *a: None)
        if n_valid == 0:
            if new_config is None:
                new_config = IndexConfig(index_type=IndexFactory.detect_index_type(old_index), dimension=old_index.d)
            return IndexFactory.create_index(new_config), {}, True
        report(0, n_valid, f"Extracting {n_valid} vectors (removing {n_total - n_valid} deleted)...")
        keep = sorted(valid_vector_ids)
        id_mapping = {old: new for new, old in enumerate(keep)}
        ids = np.asarray(keep, dtype=np.int64)
        readable = IndexFactory._first_unreadable(old_index, ids)
        for done in range(1000, readable + 1, 1000):
            report(done, n_valid, f"Extracted {done}/{n_valid} vectors...")
        if readable < n_valid:
            why = (f"id {int(ids[readable])} out of range [0, {n_total})" if 0 < readable or not (0 <= ids[0] < n_total)
                   else "this index has no direct map")
            print(f"[SemanticSearch] Warning: Cannot reconstruct vectors: {why}")
            return old_index, {}, False
        report(n_valid, n_valid, "Creating compacted index...")
        if new_config is None:
            new_config = IndexConfig(index_type=IndexFactory.detect_index_type(old_index), dimension=old_index.d)
        new_index = IndexFactory.create_index(new_config)
        new_index = IndexFactory._train_and_fill(new_index, old_index, ids, new_config, report, n_valid, "Training compacted index...")
        IndexFactory._fill(new_index, old_index, ids)
        report(n_valid, n_valid, f"Compaction complete: {n_total} -> {n_valid} vectors")
        return new_index, id_mapping, True

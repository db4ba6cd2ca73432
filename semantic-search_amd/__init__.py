"""semantic_search_amd - MI355X-native Stage-1 vector retrieval.

Drop-in for the FAISS-touching half of EricRollei/Semantic-Search's
core/index_factory.py and core/index_manager.py: `faiss_compat` offers the
`faiss.Index` duck type (IndexFlatIP, IndexIVFFlat, read_index, write_index,
METRIC_INNER_PRODUCT) on top of hand-written gfx950 HIP kernels (libssmi.so, C-ABI
in include/ssmi.h); `index_factory` / `index_manager` mirror the reference's own
modules above that.
"""
from . import faiss_compat  # noqa: F401
from .faiss_compat import (  # noqa: F401
    IndexFlatIP,
    IndexIVFFlat,
    METRIC_INNER_PRODUCT,
    METRIC_L2,
    read_index,
    write_index,
)
from .index_factory import (  # noqa: F401
    INDEX_TYPE_DESCRIPTIONS,
    IndexConfig,
    IndexFactory,
    IndexType,
)

__all__ = [
    "faiss_compat", "IndexFlatIP", "IndexIVFFlat", "METRIC_INNER_PRODUCT", "METRIC_L2", "read_index",
    "write_index", "INDEX_TYPE_DESCRIPTIONS", "IndexConfig", "IndexFactory", "IndexType",
]

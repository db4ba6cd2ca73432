"""min_score -> the fp32 floor of ssmi_index_search_min: `score_f32 >= floor` must decide what the reference's
double-precision `r.score >= min_score` decides (nodes/search_nodes.py:152-159)."""
import numpy as np

from semantic_search_amd.faiss_compat import _floor32, _NO_FLOOR


def test_floor_is_the_smallest_float32_at_or_above_min_score():
    rng = np.random.default_rng(5)
    ms = np.concatenate([rng.uniform(-1, 1, 2000), rng.uniform(0.05, 0.1, 2000), [0.0, 0.1, 0.25, 1 / 3, -1 / 3, 1e-45, -1e-45, 1e-300]])
    for m in ms:
        f = np.float32(_floor32(float(m)))
        assert float(f) >= m
        below = np.nextafter(f, np.float32(-np.inf), dtype=np.float32)
        assert float(below) < m
        # the decision on the three neighbouring scores equals the reference's double comparison
        for s in (below, f, np.nextafter(f, np.float32(np.inf), dtype=np.float32)):
            assert (s >= f) == (float(s) >= m)


def test_floor_edges():
    assert _floor32(None) == _NO_FLOOR
    assert _floor32(-np.inf) == _NO_FLOOR
    assert _floor32(-1e39) == _NO_FLOOR
    assert _floor32(0.5) == 0.5
    assert _floor32(1e39) == float(np.finfo(np.float32).max)

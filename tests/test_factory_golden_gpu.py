"""IndexFactory over the HIP engine replays the scenarios the REFERENCE's core/index_factory.py was driven through
(tests/golden/factory_golden.json, made by tests/golden/make_factory_golden.py in the build container; only the JSON
travels): create / stats / training gate, rebuild_index and compact_index with their progress-callback sequences,
id mappings, printed warnings, fall-back to Flat when training fails, the un-reconstructable and out-of-range cases
(core/index_factory.py:125-416), and the exact search result of the index that comes out (bit-equal ids: the
reference ran over the numpy oracle, ours over the kernels)."""
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

import factory_scenarios as S  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = json.load(open(os.path.join(HERE, "golden", "factory_golden.json")))


@pytest.fixture(scope="module")
def replay(ssa):
    from semantic_search_amd import faiss_compat, index_factory
    return json.loads(json.dumps(S.engine_scenarios(index_factory, faiss_compat)))


@pytest.mark.parametrize("name", sorted(GOLDEN["engine"]))
def test_engine_scenario_equals_the_reference(replay, name):
    want, got = GOLDEN["engine"][name], replay[name]
    assert set(got) == set(want), name
    for key in sorted(want):
        assert got[key] == want[key], (name, key)


def test_training_sample_draw_equals_the_reference(ssa):
    """> 100000 vectors: one np.random.choice(n, 100000, replace=False) from the GLOBAL numpy stream, like the reference
    (core/index_factory.py:293-296, 401-403): the next draw after the call is the same number."""
    from semantic_search_amd import faiss_compat, index_factory
    got = json.loads(json.dumps(S.sampling_scenario(index_factory, faiss_compat)))
    assert got == GOLDEN["sampling"]

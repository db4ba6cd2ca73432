"""Host logic of semantic_search_amd.index_factory against the REFERENCE's own core/index_factory.py: the golden file
was produced by running tests/golden/factory_scenarios.py over the reference module (tests/golden/make_factory_golden.py,
build container only); here the same scenarios run over ours.  This half needs no index: enum mapping
(core/index_factory.py:26-33), the config.json schema (:59-90), minimum training sizes (:197-203), the metadata flags of
INDEX_TYPE_DESCRIPTIONS (:94-119)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

import factory_scenarios as S  # noqa: E402
from semantic_search_amd import index_factory as F  # noqa: E402

GOLDEN = json.load(open(os.path.join(HERE, "golden", "factory_golden.json")))


def test_host_logic_equals_the_reference_modules():
    ours = json.loads(json.dumps(S.host_scenarios(F)))
    assert set(ours) == set(GOLDEN["host"])
    for name in sorted(ours):
        assert ours[name] == GOLDEN["host"][name], name


def test_golden_covers_the_control_flow_it_claims():
    e = GOLDEN["engine"]
    assert e["rebuild_train_fails_falls_back"]["config_type_after"] == "flat"
    assert e["compact_id_out_of_range"]["ok"] is False and len(e["compact_id_out_of_range"]["callbacks"]) == 2
    assert e["rebuild_unreconstructable"]["same_object"] is True
    assert [c[2] for c in e["rebuild_flat_to_ivf"]["callbacks"]][2] == "Training new index..."
    assert GOLDEN["sampling"]["rebuild"]["next_global_random_draw"] == GOLDEN["sampling"]["compact"]["next_global_random_draw"]

"""Generates tests/golden/factory_golden.json by running tests/golden/factory_scenarios.py over the REFERENCE's own
core/index_factory.py (loaded as a single file: its package __init__ pulls in cv2, qwen_vl_utils and creates H:/
directories) with tests/golden/fake_faiss.py standing in for the `faiss` module it imports.

BUILD CONTAINER ONLY: /root/reference does not exist on the GPU box, and no reference text is written anywhere -
only the scenario outputs (JSON).  What this pins is the reference's HOST logic on this path (enum mapping, config
dict schema, min training sizes, stats dicts, rebuild / compact control flow: callback sequences, id mappings,
fall-backs, numpy RNG consumption); the engine behind it is the oracle, so the arithmetic stays "parity unpinned".

    python tests/golden/make_factory_golden.py
"""
import importlib.util
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, HERE]

REF = os.environ.get("SSMI_REFERENCE", "/root/reference")


def load_reference_factory():
    import fake_faiss
    sys.modules["faiss"] = fake_faiss
    try:
        spec = importlib.util.spec_from_file_location("ref_index_factory", os.path.join(REF, "core", "index_factory.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        del sys.modules["faiss"]
    return mod, fake_faiss


def main():
    import factory_scenarios as S
    F, fa = load_reference_factory()
    golden = {"generated_by": "tests/golden/make_factory_golden.py over <reference>/core/index_factory.py + tests/golden/fake_faiss.py",
              "host": S.host_scenarios(F), "engine": S.engine_scenarios(F, fa), "sampling": S.sampling_scenario(F, fa)}
    out = os.path.join(HERE, "factory_golden.json")
    with open(out, "w") as f:
        f.write("{\n")
        for gi, group in enumerate(sorted(golden)):   # one scenario per line: small file, readable diffs
            v = golden[group]
            if isinstance(v, dict):
                body = ",\n".join("  %s: %s" % (json.dumps(k), json.dumps(v[k], sort_keys=True, separators=(",", ":"))) for k in sorted(v))
                f.write(" %s: {\n%s\n }" % (json.dumps(group), body))
            else:
                f.write(" %s: %s" % (json.dumps(group), json.dumps(v)))
            f.write(",\n" if gi + 1 < len(golden) else "\n")
        f.write("}\n")
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()

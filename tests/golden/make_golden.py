"""Generates tests/golden/flat_expected.npz + flat_cases.json with the numpy oracle.

The reference ships no fixtures for this path and FAISS is not installable here
(SURVEY.md 8c), so these are the oracle's own known answers ("parity unpinned"):
inputs are regenerated from Philox seeds, only the expected (D, I) are stored.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE.parent))
from conftest import unit_rows  # noqa: E402
from oracle import flat_ip  # noqa: E402

CASES = [
    dict(name="config1_10k_4096", n=10000, d=4096, nq=100, k=10, seed_db=1234, seed_q=4321),
    dict(name="config1_10k_4096_f16", n=10000, d=4096, nq=100, k=10, seed_db=1234, seed_q=4321, storage="f16"),
    dict(name="k_gt_n", n=7, d=256, nq=3, k=10, seed_db=11, seed_q=12),
    dict(name="k500", n=3000, d=2048, nq=2, k=500, seed_db=13, seed_q=14),
    dict(name="d256", n=5000, d=256, nq=8, k=10, seed_db=15, seed_q=16),
    dict(name="odd_d", n=513, d=100, nq=3, k=7, seed_db=17, seed_q=18),
]


def main():
    out = {}
    for c in CASES:
        X = unit_rows(c["n"], c["d"], c["seed_db"])
        if c.get("storage") == "f16":
            X = X.astype(np.float16).astype(np.float32)
        Q = unit_rows(c["nq"], c["d"], c["seed_q"])
        D, I = flat_ip.search(X, Q, c["k"])
        out[c["name"] + "_D"] = D
        out[c["name"] + "_I"] = I
    np.savez_compressed(HERE / "flat_expected.npz", **out)
    (HERE / "flat_cases.json").write_text(json.dumps({"generator": "tests/golden/make_golden.py", "cases": CASES}, indent=1))
    print("wrote", HERE / "flat_expected.npz")


if __name__ == "__main__":
    main()

"""Pins oracle/simvolume_oracle.py against golden vectors produced by the reference's own
utils/similarity_volume.py (tools/gen_golden_simvolume.py)."""
import json
import os

import numpy as np
import pytest

from oracle import simvolume_oracle as so

GOLD = os.path.join(os.path.dirname(__file__), "golden", "simvolume_golden.json")
with open(GOLD) as f:
    CASES = json.load(f)["cases"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference_golden(case):
    sims = np.array(case["sims"], dtype=np.float32)
    got = so.simvolume_assignments(sims, case["num_per_length"])
    assert got == case["expected"]


@pytest.mark.parametrize("case", [c for c in CASES if c["Q"] * (c["M"] + 1) ** min(c["Q"], 3) < 40000],
                         ids=lambda c: c["name"])
def test_literal_and_fast_topk_agree(case):
    sims = np.array(case["sims"], dtype=np.float32)
    a = so.topk_cells(sims, case["num_per_length"], literal=True)
    b = so.topk_cells(sims, case["num_per_length"], literal=False)
    assert a[0] == b[0]
    for la, lb in zip(a[1], b[1]):
        assert [(c, float(v)) for c, v in la] == [(c, float(v)) for c, v in lb]

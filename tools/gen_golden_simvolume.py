#!/usr/bin/env python3
"""Generate golden vectors for the similarity-volume assignment search.

Runs ONLY in the build container (needs /root/reference).  It loads the reference's own
`utils/similarity_volume.py` by file path (with a 3-line stub for the absent, unused `numba`
import at similarity_volume.py:8), feeds it seeded similarity matrices and records the
assignment lists `SimVolume.get_top_indices_from_subvolumes(num_per_length=4)` returns after
`fast_construct_volume(min(Q, 3))` — exactly the call sequence of
object_memory/object_memory.py:974-982.

Output: tests/golden/simvolume_golden.json  (inputs as float32 lists + expected lists).
Only data is stored; no reference source is copied.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/utils/similarity_volume.py"
OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "simvolume_golden.json")


def load_reference():
    stub = types.ModuleType("numba")
    stub.jit = lambda *a, **k: (lambda f: f)
    stub.njit = lambda *a, **k: (lambda f: f)
    sys.modules.setdefault("numba", stub)
    spec = importlib.util.spec_from_file_location("ref_similarity_volume", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run_ref(mod, sims, num_per_length=4):
    sims = np.array(sims, dtype=np.float32)
    sv = mod.SimVolume(sims.copy())
    sv.fast_construct_volume(min(sims.shape[0], 3))
    assns = sv.get_top_indices_from_subvolumes(num_per_length=num_per_length)
    return [[[int(d), int(m)] for d, m in a] for a in assns]


def main():
    mod = load_reference()
    cases = []

    def add(name, sims, npl=4):
        sims = np.asarray(sims, dtype=np.float32)
        cases.append({
            "name": name,
            "num_per_length": npl,
            "Q": int(sims.shape[0]),
            "M": int(sims.shape[1]),
            "sims": [[float(x) for x in row] for row in sims],
            "expected": run_ref(mod, sims, npl),
        })
        print(name, sims.shape, cases[-1]["expected"])

    # the module's own __main__ case (similarity_volume.py:418-452): cs[i,j] = i + j, 10x4, npl=3
    cs = np.zeros((10, 4), dtype=np.float32)
    for i in range(10):
        for j in range(4):
            cs[i, j] = i + j
    add("ref_main_10x4", cs, npl=3)

    rng = np.random.default_rng(1234)
    # cosine-like matrices, incl. negatives
    for Q in (1, 2, 3, 4, 5, 7):
        for M in (3, 4, 7, 12, 20):
            if Q > M:
                continue
            if Q >= 5 and M > 12:
                continue
            add(f"uniform_Q{Q}_M{M}", rng.uniform(-1, 1, size=(Q, M)))
    # peaked (a matching instance per detection) + small noise
    for Q, M in ((3, 20), (4, 16), (7, 9)):
        s = rng.normal(0, 0.05, size=(Q, M))
        for i in range(Q):
            s[i, rng.integers(0, M)] = rng.uniform(0.6, 0.95)
        add(f"peaked_Q{Q}_M{M}", s)
    # heavy ties: few distinct values
    for Q, M in ((3, 6), (4, 8), (2, 5), (3, 12)):
        add(f"ties_Q{Q}_M{M}", rng.choice([0.25, 0.5, -0.5, 1.0], size=(Q, M)))
    # all-equal rows and zeros
    add("allequal_Q3_M5", np.full((3, 5), 0.5))
    add("zeros_Q3_M4", np.zeros((3, 4)))
    add("allneg_Q3_M6", -np.abs(rng.uniform(0.1, 1, size=(3, 6))))
    # tiny M (fewer finite cells than k -> -inf filler entries)
    add("tiny_Q2_M2", rng.uniform(-1, 1, size=(2, 2)))
    add("tiny_Q3_M3", rng.uniform(-1, 1, size=(3, 3)))
    add("tiny_Q1_M1", rng.uniform(0, 1, size=(1, 1)))
    add("tiny_Q1_M5", rng.uniform(-1, 1, size=(1, 5)))
    add("tiny_Q2_M3", rng.uniform(-1, 1, size=(2, 3)))
    # M smaller than the volume dimension (SimVolume used stand-alone; localise never does this)
    add("short_Q3_M2", rng.uniform(0, 1, size=(3, 2)))
    # duplicated columns
    s = rng.uniform(-1, 1, size=(4, 6))
    s[:, 3] = s[:, 1]
    s[:, 5] = s[:, 1]
    add("dupcols_Q4_M6", s)
    # config-1 scale (M=20, Q=7) and the M=51 log-scale case with Q=3
    add("c1_Q7_M20", rng.normal(0, 0.2, size=(7, 20)).clip(-1, 1))
    add("log_Q3_M51", rng.normal(0.1, 0.2, size=(3, 51)).clip(-1, 1))

    with open(OUT, "w") as f:
        json.dump({"generator": "tools/gen_golden_simvolume.py",
                   "reference": "utils/similarity_volume.py @ 2024-10-22",
                   "cases": cases}, f)
    print("wrote", OUT, len(cases), "cases")


def extra():
    """The parts of SimVolume localise() does not use (sub-volumes of another size than min(Q, 3), construct_volume, get_top_indices,
    conv_coords_to_pairs, construct_volume_choose_e): outputs of the reference's own module on small seeded matrices."""
    import contextlib
    import io
    mod = load_reference()
    rng = np.random.default_rng(77)
    cases = []
    for Q, M, size in ((4, 6, 2), (5, 5, 4), (3, 7, 2), (4, 5, 3), (2, 6, 2), (5, 6, 2), (1, 4, 1)):
        sims = rng.uniform(-0.3, 1, size=(Q, M)).astype(np.float32)
        with contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):
            a = mod.SimVolume(sims.copy())
            a.fast_construct_volume(size)
            assns = a.get_top_indices_from_subvolumes(3)
            c = {"Q": Q, "M": M, "subvolume_size": size, "sims": [[float(x) for x in r] for r in sims],
                 "assignments": [[[int(d), int(m)] for d, m in asg] for asg in assns]}
            if Q >= 2:
                vol, rep = mod.SimVolume(sims.copy()).construct_volume()
                top = mod.SimVolume(sims.copy()).get_top_indices(rep.copy(), 6)
                pairs = mod.SimVolume(sims.copy()).conv_coords_to_pairs(rep, top)
                ce = mod.SimVolume(sims.copy()).construct_volume_choose_e([Q - 1, 0])
                c.update({"volume_shape": list(vol.shape), "finite_cells": int(np.isfinite(rep).sum()),
                          "volume_sum": float(vol.astype(np.float64).sum()), "rep_finite_sum": float(rep[np.isfinite(rep)].astype(np.float64).sum()),
                          "top6": [[[int(x) for x in cell], float(v)] for cell, v in top],
                          "top6_pairs": [[[[int(i), int(j)] for i, j in pr], float(v)] for pr, v in pairs],
                          "choose_e_last_first": [[float(x) for x in r] for r in ce]})
        cases.append(c)
    out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "simvolume_extra_golden.json")
    json.dump({"generator": "tools/gen_golden_simvolume.py extra()", "cases": cases}, open(out, "w"))
    print("wrote", out, len(cases), "cases")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "extra":
        extra()
    else:
        main()

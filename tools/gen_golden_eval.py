#!/usr/bin/env python3
"""Golden vectors for the pose-error metric, produced by the reference's own utils/quaternion_ops.py (pure numpy; loaded by
file path so that nothing else of the reference is imported).  Writes tests/golden/eval_golden.json."""
import importlib.util
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ref_quaternion_ops", "/root/reference/utils/quaternion_ops.py")
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
Q = mod.QuaternionOps

rng = np.random.default_rng(20241022)
cases = []
for i in range(64):
    a, b = rng.normal(size=4), rng.normal(size=4)
    if i % 4 != 3:                                    # mostly unit quaternions, some unnormalised (the metric does not normalise)
        a, b = a / np.linalg.norm(a), b / np.linalg.norm(b)
    if i % 8 == 0:
        b = a.copy()
    if i % 8 == 1:
        b = -a
    if i % 8 == 2:
        b = a + 1e-9 * rng.normal(size=4)
    cases.append({"q1": a.tolist(), "q2": b.tolist(), "product": Q.quaternion_multiply(a, b).tolist(),
                  "conjugate": Q.quaternion_conjugate(a).tolist(), "error": float(Q.quaternion_error(a, b))})
spec = importlib.util.spec_from_file_location("ref_phrases", "/root/reference/object_memory/object_finder_phrases.py")   # pure python
ph = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ph)
name_lists = [["floor"], ["ground", "mat"], ["earth"], ["table"], ["floor mat"], ["wooden floor"], [], ["chair", "floor"], ["Floor"], ["ground floor"],
              ["desk", "earth"], ["grounds"]]
floor = [{"names": c, "is_floor": bool(ph.check_if_floor(c))} for c in name_lists]
with open(os.path.join(ROOT, "tests", "golden", "eval_golden.json"), "w") as fh:
    json.dump({"source": "reference utils/quaternion_ops.py QuaternionOps", "cases": cases, "check_if_floor": floor,
               "check_if_floor_source": "reference object_memory/object_finder_phrases.py check_if_floor"}, fh, indent=0)
print(len(cases), "cases")

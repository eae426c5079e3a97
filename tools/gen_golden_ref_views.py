#!/usr/bin/env python3
"""Oracle transcript of localise() on ALL EIGHT views of the reference's own synthetic room (data/our-synthetic/360_basic_test, fixtures
under tests/golden/ref_scene/) against the three objects of the memory it saved (out/360_trial_with_floor) -> tests/golden/ref_scene/
oracle_views.json.  The oracle (oracle/reg_oracle.py, C restatement) needs 15 minutes of CPU for the eight views (a 55 k-point detection
against a 54 k-point object in view 3), so its outputs are stored and the -m gpu test compares the HIP path with them; the CPU suite
re-runs the cheapest view against the file (tests/test_ref_scene.py).  Driver parameters: voxel 0.05, global / local factors 1.5 / 1.5
(tum_localisation_trial.py:473-488); detections = the objects that cover >= MIN_PIXELS pixels of the view (tests/ref_scene.py masks).

    python tools/gen_golden_ref_views.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import ref_scene as rs  # noqa: E402


def main():
    from oracle import reg_oracle as ro
    objs = rs.memory_objects()
    out = {"min_pixels": rs.MIN_PIXELS, "seed": rs.VIEW_SEED, "views": {}}
    job = 0
    for k, fr in rs.view_frames().items():
        t0 = time.time()
        cleaned, ccols = [], []
        for pts, inten in zip(fr["clouds"], fr["ints"]):
            keep = ro.radius_outlier(pts, 0.05, 8)
            cleaned.append(pts[keep])
            ccols.append(np.repeat(inten[keep][:, None], 3, axis=1))
        assns = rs.oracle_assignments(fr["det_emb"])
        pose, recs, best = ro.localise_from_assignments(cleaned, ccols, [o[0] for o in objs], [o[1] for o in objs], assns, 0.05, 1.5, 1.5,
                                                        seed=rs.VIEW_SEED, job_base=job, stale_means=False)
        te, re_ = rs.pose_error(pose, rs.pose_matrix(fr["pose"]))
        out["views"][str(k)] = {"seen": fr["seen"], "n_clean": [len(c) for c in cleaned], "assignments": assns, "job_base": job, "best": int(best),
                                "pose": [float(x) for x in pose], "gt_err_m": te, "gt_err_rad": re_,
                                "fitness": [float(r["fitness"]) for r in recs], "rmse": [float(r["rmse"]) for r in recs],
                                "full_fitness": [float(r["full_fitness"]) for r in recs], "full_rmse": [float(r["full_rmse"]) for r in recs],
                                "T": [np.asarray(r["T"]).tolist() for r in recs]}
        job += len(assns)
        print(k, fr["seen"], "best", best, "gt err %.3f m %.3f rad" % (te, re_), "%.0f s" % (time.time() - t0), flush=True)
    path = os.path.join(rs.DIR, "oracle_views.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()

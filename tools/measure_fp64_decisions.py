#!/usr/bin/env python3
"""CPU: count the discrete decisions that differ between the independent fp64 restatement (oracle/open3d_fp64.py) and the fp32 rule of
oracle_reg.c / the device, on the reference's three saved objects and the detections of all eight of its RGB-D views (the full version of
what tests/test_open3d_fp64.py asserts on a subset; ~5 CPU-minutes).  python tools/measure_fp64_decisions.py > profiles/r04/fp64_decisions.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import open3d_fp64 as o3  # noqa: E402
from oracle import reg_oracle as ro  # noqa: E402
from tests import ref_scene as rs  # noqa: E402
from tests.test_open3d_fp64 import angle_between, set_differences  # noqa: E402


def main():
    objs = rs.memory_objects()
    frames = rs.view_frames()
    clouds = [("obj%d" % i, o[0]) for i, o in enumerate(objs)]
    for k, f in frames.items():
        clouds += [(f"view{k}_det{d}", c) for d, c in enumerate(f["clouds"])]
    tot = n = 0
    for name, c in clouds:
        k64, _ = o3.radius_outlier_keep(c, 0.05, 8)
        k32 = ro.radius_outlier(np.asarray(c, np.float32), 0.05, 8)
        d = int((k64 != k32).sum())
        tot += d
        n += len(c)
        print(f"radius_outlier {name}: {len(c)} points, {int((~k64).sum())} removed, {d} decisions differ")
    print(f"radius_outlier total: {tot} of {n} decisions differ")
    for i, (p, c) in enumerate(objs):
        n64, idx64, cnt64 = o3.normals(p, 0.1, 30)
        p32 = p.astype(np.float32)
        idx32, cnt32 = ro.hybrid_sets(p32, 0.1, 30)
        nd, nt = set_differences(p, idx64, idx32)
        same = ~np.any(np.sort(idx64, axis=1) != np.sort(idx32, axis=1), axis=1)
        ang = angle_between(n64, ro.normals(p32, 0.1, 30))
        print(f"normals obj{i}: {len(p)} points, neighbour counts differ {int((cnt64 != cnt32).sum())}, 30-NN sets differ {nd} ({nt} exact fp64 ties), "
              f"direction where sets agree: median {np.median(ang[same]):.2e} p99.9 {np.quantile(ang[same], 0.999):.2e} max {ang[same].max():.2e} rad")


if __name__ == "__main__":
    main()

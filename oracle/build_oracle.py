"""TEST INFRASTRUCTURE ONLY.  CPU restatement of the memory-build numerics.

* `voxel_down_sample_with_colors` — literal restatement of /root/reference/utils/depth_utils.py:211-265 on (N, 3) arrays
  (python dict keyed by the voxel tuple, np.mean per voxel).  The reference function itself needs open3d (not in this image) only
  as the container of its arrays; the arithmetic is numpy's, which this restatement calls the same way, so the outputs are
  those of the reference on the same arrays.
* `cluster_dbscan` — restatement of open3d 0.17.0 PointCloud::ClusterDBSCAN (geometry/PointCloudCluster.cpp; third-party,
  pinned by /root/reference/environment.yml, source not in the image): neighbour lists by radius search with squared distance
  < eps^2 (nanoflann RadiusResultSet), sequential scan with noise -> border relabelling.  PARITY UNPINNED at the Open3D
  boundary; cross-checked against scikit-learn's DBSCAN (same core / border / numbering rules) in tests/test_oracle_build.py.
"""
import numpy as np


def voxel_down_sample_with_colors(points, colors, voxel_size):
    points = np.asarray(points, dtype=np.float64)
    voxel_indices = np.floor(points / voxel_size).astype(np.int64)
    voxel_dict = {}
    for i, idx in enumerate(voxel_indices):
        key = tuple(idx)
        if key not in voxel_dict:
            voxel_dict[key] = {"points": [], "colors": []}
        voxel_dict[key]["points"].append(points[i])
        if colors is not None:
            voxel_dict[key]["colors"].append(colors[i])
    dp, dc, cnt = [], [], []
    for voxel in voxel_dict.values():
        dp.append(np.mean(voxel["points"], axis=0))
        if colors is not None:
            dc.append(np.mean(voxel["colors"], axis=0))
        cnt.append(len(voxel["points"]))
    dp = np.array(dp, dtype=np.float64).reshape(-1, 3)
    return dp, (np.array(dc, dtype=np.float64).reshape(-1, 3) if colors is not None else None), np.array(cnt, dtype=np.int32)


def radius_neighbours(points, eps):
    """nbs[i] = indices j with ((dx^2 + dy^2) + dz^2) < eps^2, self included (brute force, blocks of rows)."""
    P = np.asarray(points, dtype=np.float64)
    n = len(P)
    out = []
    e2 = eps * eps
    for b in range(0, n, 512):
        d = P[b:b + 512, None, :] - P[None, :, :]
        d2 = d[..., 0] * d[..., 0]
        d2 = d2 + d[..., 1] * d[..., 1]
        d2 = d2 + d[..., 2] * d[..., 2]
        for row in d2 < e2:
            out.append(np.flatnonzero(row))
    return out


def cluster_dbscan(points, eps, min_points):
    nbs = radius_neighbours(points, eps)
    n = len(nbs)
    labels = np.full(n, -2, dtype=np.int64)
    cluster = 0
    for idx in range(n):
        if labels[idx] != -2:
            continue
        if len(nbs[idx]) < min_points:
            labels[idx] = -1
            continue
        nxt = set(int(v) for v in nbs[idx])
        visited = {idx}
        labels[idx] = cluster
        while nxt:
            nb = nxt.pop()
            visited.add(nb)
            if labels[nb] == -1:
                labels[nb] = cluster
            if labels[nb] != -2:
                continue
            labels[nb] = cluster
            if len(nbs[nb]) >= min_points:
                for q in nbs[nb]:
                    if int(q) not in visited:
                        nxt.add(int(q))
        cluster += 1
    return labels.astype(np.int32)

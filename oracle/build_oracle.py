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


# ---- consolidation transcript (object_memory.py:258-263, 297-368, 562-747; object_info.py:61-97) -------------------------------
class Obj:
    """ObjectInfo reduced to what the consolidation touches."""

    def __init__(self, name, emb, points, colors):
        self.names = [name]
        self.embeddings = [np.asarray(emb)]
        self.points = np.asarray(points, dtype=np.float64)
        self.colors = np.asarray(colors, dtype=np.float64)
        self.compute_means()

    @property
    def pcd(self):
        return self.points.T

    def compute_means(self):
        self.mean_emb = np.mean(np.array(self.embeddings), axis=0).squeeze()
        self.centroid = np.mean(self.pcd, axis=-1)

    def __add__(self, o):
        for n in o.names:
            if n not in self.names:
                self.names.append(n)
        self.embeddings += o.embeddings
        self.points = np.vstack((self.points, o.points))
        self.colors = np.vstack((self.colors, o.colors))
        return self


def downsample_all(memory, voxel_size):
    for o in memory:
        o.points, o.colors, _ = voxel_down_sample_with_colors(o.points, o.colors, voxel_size)


def _assign_and_merge(objs, all_points, labels):
    def is_point_in_array(points_array, query_point):
        return np.any(np.all(points_array == query_point, axis=1))
    assn = np.full(len(objs), -1)
    for index, obj in enumerate(objs):
        query_point = obj.pcd[:, 0]
        for label in np.unique(labels):
            if label == -1:
                continue
            if is_point_in_array(all_points[labels == label], query_point):
                assn[index] = label
    out = []
    for label in np.unique(labels):
        if label == -1:
            continue
        to_combine = [objs[i] for i in range(len(objs)) if assn[i] == label]
        if len(to_combine) == 0:
            continue
        acc = to_combine[0]
        for o in to_combine[1:]:
            acc = acc + o
        out.append(acc)
    return out


def recluster_objects_with_dbscan(memory, eps, min_points):
    all_points = np.concatenate([o.pcd for o in memory], axis=-1).T
    return _assign_and_merge(memory, all_points, cluster_dbscan(all_points, eps, min_points))


def recluster_IoU(memory, IoU_threshold, iou_func):
    from sklearn.cluster import AgglomerativeClustering
    n = len(memory)
    IoUs = np.zeros((n, n))
    thr = 1 - IoU_threshold
    for i in range(n):
        for j in range(i, n):
            if i == j:
                IoUs[i][j] = 1
                continue
            IoUs[i][j] = 1 - iou_func(memory[i].points, memory[j].points)
            IoUs[j][i] = IoUs[i][j]
    labels = AgglomerativeClustering(n_clusters=None, distance_threshold=thr, metric='precomputed', linkage='average').fit(IoUs).labels_
    new_memory = [None for _ in set(labels)]
    for lab, o in zip(labels, memory):
        new_memory[lab] = o if new_memory[lab] is None else new_memory[lab] + o
    for o in new_memory:
        o.compute_means()
    return new_memory


def recluster_via_clustering_and_IoU(memory, embedding_distance_threshold, eps, min_points, IoU_threshold, iou_func):
    from sklearn.cluster import AgglomerativeClustering
    memory = recluster_IoU(memory, IoU_threshold, iou_func)
    embs = np.array([o.mean_emb for o in memory])
    normalized = embs / np.linalg.norm(embs, axis=1, keepdims=True)
    D = 1 - np.dot(normalized, normalized.T)
    D -= np.min(D)
    D /= np.max(D)
    D = 1 - D
    labels = AgglomerativeClustering(n_clusters=None, distance_threshold=embedding_distance_threshold, metric='precomputed',
                                     linkage='average').fit(D).labels_
    out = []
    for u in set(labels):
        objs = [o for i, o in enumerate(memory) if labels[i] == u]
        all_points = np.concatenate([o.pcd for o in objs], axis=-1).T
        out = out + _assign_and_merge(objs, all_points, cluster_dbscan(all_points, eps, min_points))
    return out


def aabb_iou(p1, p2):            # utils/IoU_ops.py:9-51 on arrays
    a, b = np.asarray(p1).T, np.asarray(p2).T
    lo = np.stack([a.min(axis=-1), b.min(axis=-1)], axis=0).max(axis=0)
    hi = np.stack([a.max(axis=-1), b.max(axis=-1)], axis=0).min(axis=0)
    if (lo > hi).any():
        return 0
    v = hi - lo
    ov = v[0] * v[1] * v[2]
    e1, e2 = a.max(axis=-1) - a.min(axis=-1), b.max(axis=-1) - b.min(axis=-1)
    return ov / (e1[0] * e1[1] * e1[2] + e2[0] * e2[1] * e2[2] - ov)


def recluster_via_agglomerative_clustering(memory, embedding_distance_threshold):       # object_memory.py:379-437
    from sklearn.cluster import AgglomerativeClustering
    embs = np.array([o.mean_emb for o in memory])
    normalized = embs / np.linalg.norm(embs, axis=1, keepdims=True)
    D = np.dot(normalized, normalized.T)
    D -= np.min(D)
    D /= np.max(D)
    D = 1 - D
    labels = AgglomerativeClustering(n_clusters=None, distance_threshold=embedding_distance_threshold, metric='precomputed',
                                     linkage='average').fit(D).labels_
    new_memory = [None for _ in set(labels)]
    for lab, o in zip(labels, memory):
        new_memory[lab] = o if new_memory[lab] is None else new_memory[lab] + o
    return new_memory

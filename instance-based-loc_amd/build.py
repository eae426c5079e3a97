"""Device side of the memory build / consolidation step: batched voxel down-sampling and DBSCAN labels.

The build side of the reference keeps Open3D's double-precision clouds (`ObjectInfo.pointcloud`), so these two calls take and
return float64 arrays; everything numerical runs in libibloc_hip.so (csrc/build_memory.hip).

* `voxel_downsample_batch`  — `/root/reference/utils/depth_utils.py:211-265` for every object of a memory at once
  (`object_memory.py:258-263`); bit-identical to the reference's dict + `np.mean` (voxels in order of first occurrence)
* `dbscan_batch`            — `open3d ... cluster_dbscan(eps, min_points)` (`object_memory.py:305`, `:631`) for any number of
  independent groups of concatenated clouds
"""
import numpy as np
import torch

from . import _lib
from .registration import RegContext, _stream


def _offsets(sizes):
    off = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=off[1:])
    if off[-1] >= 2 ** 31:
        raise ValueError("more than 2^31 points in one call")
    return off.astype(np.int32)


def _dev_f64(arrays, device):
    if len(arrays) == 0:
        return torch.empty((0, 3), dtype=torch.float64, device=device)
    a = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.float64).reshape(-1, 3) for x in arrays], axis=0))
    return torch.from_numpy(a).to(device)


def voxel_downsample_batch(ctx: RegContext, points, colors, voxel_size, device="cuda", return_counts=False):
    """points / colors: lists of (N_i, 3) float64 arrays (colors may be None).  Returns (points, colors[, counts]) as lists of
    arrays, one per object, exactly what `voxel_down_sample_with_colors` returns object by object."""
    sizes = [len(p) for p in points]
    off = _offsets(sizes)
    n = int(off[-1])
    P = _dev_f64(points, device)
    Cc = _dev_f64(colors, device) if colors is not None else None
    if Cc is not None and Cc.shape != P.shape:
        raise ValueError("colors must match points")
    out_p = torch.empty((max(n, 1), 3), dtype=torch.float64, device=device)
    out_c = torch.empty((max(n, 1), 3), dtype=torch.float64, device=device) if Cc is not None else None
    out_n = torch.empty(max(n, 1), dtype=torch.int32, device=device) if return_counts else None
    out_off = np.zeros(len(sizes) + 1, dtype=np.int32)
    st = _lib.lib.ibl_voxel_downsample_batch(ctx.handle, P.data_ptr(), Cc.data_ptr() if Cc is not None else None, off.ctypes.data, len(sizes),
                                             float(voxel_size), out_p.data_ptr(), out_c.data_ptr() if out_c is not None else None,
                                             out_n.data_ptr() if out_n is not None else None, out_off.ctypes.data, _stream())
    _lib.check(st, "ibl_voxel_downsample_batch")
    m = int(out_off[-1])
    hp = out_p[:m].cpu().numpy()
    hc = out_c[:m].cpu().numpy() if out_c is not None else None
    pts = [hp[out_off[i]:out_off[i + 1]] for i in range(len(sizes))]
    cols = [hc[out_off[i]:out_off[i + 1]] for i in range(len(sizes))] if hc is not None else None
    if return_counts:
        hn = out_n[:m].cpu().numpy()
        return pts, cols, [hn[out_off[i]:out_off[i + 1]] for i in range(len(sizes))]
    return pts, cols


def dbscan_batch(ctx: RegContext, groups, eps, min_points, device="cuda"):
    """groups: list of (N_g, 3) float64 arrays (each the concatenation the reference clusters in one call).  Returns a list of
    int32 label arrays (-1 = noise, clusters numbered from 0 in Open3D's order) and the number of clusters per group."""
    sizes = [len(g) for g in groups]
    off = _offsets(sizes)
    n = int(off[-1])
    P = _dev_f64(groups, device)
    labels = torch.empty(max(n, 1), dtype=torch.int32, device=device)
    n_clusters = np.zeros(max(len(sizes), 1), dtype=np.int32)
    st = _lib.lib.ibl_dbscan_batch(ctx.handle, P.data_ptr(), off.ctypes.data, len(sizes), float(eps), int(min_points), labels.data_ptr(),
                                   n_clusters.ctypes.data, _stream())
    _lib.check(st, "ibl_dbscan_batch")
    h = labels[:n].cpu().numpy()
    return [h[off[i]:off[i + 1]] for i in range(len(sizes))], n_clusters[:len(sizes)].copy()


# ---- host helpers of the consolidation step ----------------------------------------------------------------------------------
_default_ctx = None


def default_ctx():
    """Arena for stand-alone calls (ObjectInfo.downsample outside an ObjectMemory)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = RegContext(1 << 30)
    return _default_ctx


def _rows(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, 3) + 0.0)      # -0.0 == 0.0 like the elementwise test
    return a.view([("x", np.float64), ("y", np.float64), ("z", np.float64)]).ravel()


def clusters_of_first_points(all_points, labels, first_points):
    """object_memory.py:318-333 / :638-649: an object joins the cluster that contains a point EQUAL to its first point; when
    several clusters do, the highest label wins (the reference loops over the labels in ascending order and overwrites).
    Returns (assignment per object, number of objects that matched more than one label)."""
    labels = np.asarray(labels)
    av, qv = _rows(all_points), _rows(first_points)
    out = np.full(len(qv), -1, dtype=np.int64)
    hit = np.isin(av, qv) & (labels >= 0)
    multi = 0
    if hit.any():
        rows, labs = av[hit], labels[hit]
        for i, q in enumerate(qv):
            m = labs[rows == q]
            if len(m):
                out[i] = m.max()
                multi += len(np.unique(m)) > 1
    return out, multi


def merge_objects_by_cluster(objects, assignments, n_clusters):
    """object_memory.py:339-362 / :652-667: per label in ascending order, the objects assigned to it are folded into the first
    one with ObjectInfo.__add__; labels nobody was assigned to produce nothing; unassigned objects (-1) are dropped."""
    merged = []
    for label in range(int(n_clusters)):
        group = [objects[i] for i in range(len(objects)) if assignments[i] == label]
        if not group:
            continue
        acc = group[0]
        for other in group[1:]:
            acc = acc + other
        merged.append(acc)
    return merged


def transform_points(points, pose):
    """utils/depth_utils.py:92-116 on an (N, 3) array: pose = x y z qx qy qz qw; the quaternion slice is normalised IN PLACE like
    the reference (`q /= norm` on a view of the caller's pose)."""
    from scipy.spatial.transform import Rotation
    t = pose[:3]
    q = pose[3:]
    q /= np.linalg.norm(q)
    R = Rotation.from_quat(q).as_matrix()
    return (R @ np.asarray(points).T).T + t

"""Drop-in for the reference's `utils/fpfh_register.py` on the MI355X build.

Same function names, argument meaning and return shapes as /root/reference/utils/fpfh_register.py:
  downsample_and_compute_fpfh(pcd, voxel_size)                     :86-98
  register_point_clouds(source, target, voxel_size, gdf, ldf)      :100-143
  evaluate_transform(source, target, trans_init, threshold)        :145-150
  get_transformation / get_SVD_transform (numpy Kabsch helpers)    :24-82
The arithmetic the reference delegates to Open3D runs in libibloc_hip.so (csrc/reg_*.hip).  Clouds may be
Open3D point clouds (anything exposing `.points` / `.colors`) or plain (N, 3) arrays / (points, colors) tuples;
inputs are never mutated.  Open3D itself is not required.
"""
import numpy as np
import torch

from ibloc_amd.engine import intensity_from_colors
from ibloc_amd.registration import CloudBatch, MemGrid, RegContext, evaluate_batch, normals_fpfh_batch, register_batch

_CTX = {"reg": None, "eval": None, "calls": 0}
RANSAC_SEED = 0          # Open3D draws from an unseeded global RNG; here the draw sequence is a function of (seed, call #)


def _ctx(kind="reg"):
    if _CTX[kind] is None:
        _CTX[kind] = RegContext(4 << 30)
    return _CTX[kind]


class Cloud:
    """Minimal point-cloud record returned by downsample_and_compute_fpfh (points / colors / normals as (N, 3) float64)."""

    def __init__(self, points, colors=None, normals=None):
        self.points = np.asarray(points, dtype=np.float64)
        self.colors = None if colors is None else np.asarray(colors, dtype=np.float64)
        self.normals = None if normals is None else np.asarray(normals, dtype=np.float64)

    def has_colors(self):
        return self.colors is not None and len(self.colors) == len(self.points) and len(self.points) > 0


class Feature:
    """Mirror of open3d.pipelines.registration.Feature: `.data` is (33, N) float64."""

    def __init__(self, data):
        self.data = data

    def dimension(self):
        return self.data.shape[0]

    def num(self):
        return self.data.shape[1]


def _as_cloud(pcd) -> Cloud:
    if isinstance(pcd, Cloud):
        return pcd
    if isinstance(pcd, (tuple, list)) and len(pcd) == 2 and np.ndim(pcd[0]) == 2:
        return Cloud(pcd[0], pcd[1])
    if hasattr(pcd, "points"):
        pts = np.asarray(pcd.points)
        cols = np.asarray(pcd.colors) if hasattr(pcd, "colors") else None
        if cols is not None and len(cols) != len(pts):
            cols = None
        return Cloud(pts, cols)
    return Cloud(np.asarray(pcd))


def _batch(c: Cloud) -> CloudBatch:
    inten = [intensity_from_colors(c.colors)] if c.has_colors() else None
    return CloudBatch.from_numpy([c.points.reshape(-1, 3)], inten)


def get_transformation(source, target):
    """Row-vector Procrustes of the reference (:24-65): returns (R, t) with target ~= source @ R + t."""
    source, target = np.array(source), np.array(target)
    cs, ct = np.mean(source, axis=0), np.mean(target, axis=0)
    H = np.dot((source - cs).T, target - ct)
    U, _, Vt = np.linalg.svd(H)
    R = np.dot(Vt.T, U.T)
    if np.linalg.det(R) < 0:
        Vt[2, :] *= -1
        R = np.dot(Vt.T, U.T)
    return R, ct - np.dot(cs, R)


def get_SVD_transform(p, q):
    """Column-convention Kabsch of the reference (:67-82): 4x4 T with q ~= R p + t."""
    u_p, u_q = np.mean(p, axis=0), np.mean(q, axis=0)
    W = (q - u_q).T @ (p - u_p)
    u, s, vh = np.linalg.svd(W, full_matrices=True)
    M = np.diag([1, 1, np.linalg.det(u) * np.linalg.det(vh)])
    R = u @ M @ vh
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, u_q - R @ u_p
    return T


def downsample_and_compute_fpfh(pcd, voxel_size):
    """No down-sampling happens in the reference either (:86-98 only deep-copies); normals use radius 2*voxel / 30 nn,
    FPFH radius 5*voxel / 100 nn."""
    c = _as_cloud(pcd)
    b = _batch(c)
    nrm, fpfh = normals_fpfh_batch(_ctx(), b, voxel_size * 2, 30, voxel_size * 5, 100)
    out = Cloud(c.points.copy(), None if c.colors is None else c.colors.copy(), nrm[:, :3].double().cpu().numpy())
    return out, Feature(fpfh.double().cpu().numpy().T.copy())


def register_point_clouds(source, target, voxel_size, global_dist_factor=1.5, local_dist_factor=0.4):
    s, t = _as_cloud(source), _as_cloud(target)
    colored = s.has_colors() and t.has_colors()       # coloured ICP throws without colours -> the reference's bare except
    _CTX["calls"] += 1
    out = register_batch(_ctx(), _batch(s), _batch(t), [[0, -1, -1]], [[0, -1, -1]], voxel_size, global_dist_factor,
                         local_dist_factor, seed=RANSAC_SEED, job_id_base=_CTX["calls"], have_colors=colored, center=False)
    return out["T"][0], float(out["rmse"][0]), float(out["fitness"][0])


def evaluate_transform(source, target, trans_init, threshold=0.02):
    s, t = _as_cloud(source), _as_cloud(target)
    ctx = _ctx("eval")
    ctx.reset()
    tb = _batch(t)
    grid = MemGrid(ctx, tb.pts4, cell=2 * threshold)
    sb = _batch(s)
    rmse, fit = evaluate_batch(ctx, grid, sb.pts4, [0], [sb.n], np.asarray(trans_init, dtype=np.float64).reshape(1, 16), threshold)
    grid.close()
    return float(rmse[0]), float(fit[0])

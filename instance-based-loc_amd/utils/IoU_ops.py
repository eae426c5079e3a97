"""Box overlap measures of the memory consolidation step, mirroring /root/reference/utils/IoU_ops.py on (N, 3) arrays or
anything with `.points`.

`calculate_3d_IoU` (:9-51) and `calculate_strict_overlap` (:53-95) are axis-aligned and pure numpy.  The object-aligned
`calculate_obj_aligned_3d_IoU` (:97-145) -- the measure `_recluster_IoU` uses by default (object_memory.py:710-747) -- is built in
the reference from two third-party pieces that are in neither this image nor the reference tree: Open3D's
`OrientedBoundingBox.create_from_points` (Qhull convex hull, then the PCA frame of the hull's vertices and the extents of the
hull in that frame) and the Objectron box IoU (exact volume of the intersection of two boxes; its submodule directory is empty).
Both are restated here on scipy's Qhull bindings: the hull + PCA box as Open3D 0.17 builds it, the intersection as the polytope
of the twelve face half-spaces.  Host-side numpy / scipy like the reference's (memory consolidation is offline work); parity at
the third-party boundary is unpinned, the geometry is checked against closed-form cases (tests/test_iou_ops.py).
"""
import numpy as np


def _pts(p):
    return np.asarray(p.points if hasattr(p, "points") else p, dtype=np.float64).reshape(-1, 3).T


def _boxes(pcd1, pcd2):
    a, b = _pts(pcd1), _pts(pcd2)
    if a.shape[1] == 0 or b.shape[1] == 0:
        return None
    a0, a1, b0, b1 = a.min(axis=-1), a.max(axis=-1), b.min(axis=-1), b.max(axis=-1)
    lo, hi = np.stack([a0, b0], axis=0).max(axis=0), np.stack([a1, b1], axis=0).min(axis=0)
    if (lo > hi).any():
        return None
    v = hi - lo
    e1, e2 = a1 - a0, b1 - b0
    return v[0] * v[1] * v[2], e1[0] * e1[1] * e1[2], e2[0] * e2[1] * e2[2]


def calculate_3d_IoU(pcd1, pcd2):
    r = _boxes(pcd1, pcd2)
    if r is None:
        return 0
    overlap, v1, v2 = r
    return overlap / (v1 + v2 - overlap)


def calculate_strict_overlap(pcd1, pcd2):
    r = _boxes(pcd1, pcd2)
    if r is None:
        return 0
    overlap, v1, v2 = r
    return overlap / min(v1, v2)


def oriented_bounding_box(points):
    """(centre (3,), R (3, 3) columns = box axes, half extents (3,)) of the PCA box of the convex hull of `points`, the construction
    of Open3D 0.17's OrientedBoundingBox::CreateFromPoints(robust=False): Qhull hull -> mean and covariance of the hull's vertices
    -> eigenvectors as axes (made right-handed) -> min / max of the hull in that frame.  Raises like Open3D when the hull cannot
    be built (fewer than 4 points, coplanar input): the caller maps that to IoU 0 (IoU_ops.py:115-124)."""
    from scipy.spatial import ConvexHull
    p = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    hull = ConvexHull(p)
    v = p[hull.vertices]
    mean = v.mean(axis=0)
    d = v - mean
    cov = d.T @ d / len(v)
    _, vec = np.linalg.eigh(cov)
    R = vec[:, ::-1].copy()                       # largest variance first
    R[:, 2] = np.cross(R[:, 0], R[:, 1])          # right-handed frame
    loc = (v - mean) @ R
    lo, hi = loc.min(axis=0), loc.max(axis=0)
    centre = mean + R @ ((lo + hi) / 2)
    return centre, R, (hi - lo) / 2


def _box_halfspaces(centre, R, half):
    """six rows [n | -offset] with n . x - offset <= 0 inside the box"""
    rows = []
    for a in range(3):
        n = R[:, a]
        rows.append(np.concatenate([n, [-(n @ centre + half[a])]]))
        rows.append(np.concatenate([-n, [-(-n @ centre + half[a])]]))
    return np.array(rows)


def oriented_box_intersection_volume(box1, box2):
    """exact volume of the intersection of two oriented boxes (what Objectron's iou.IoU computes by clipping faces)"""
    from scipy.optimize import linprog
    from scipy.spatial import ConvexHull, HalfspaceIntersection
    hs = np.vstack([_box_halfspaces(*box1), _box_halfspaces(*box2)])
    # Chebyshev centre of the polytope: an interior point for Qhull, and the test for an empty / flat intersection
    A = np.hstack([hs[:, :3], np.linalg.norm(hs[:, :3], axis=1, keepdims=True)])
    res = linprog(c=[0, 0, 0, -1], A_ub=A, b_ub=-hs[:, 3], bounds=[(None, None)] * 3 + [(0, None)], method="highs")
    if not res.success or res.x[3] <= 1e-12:
        return 0.0
    pts = HalfspaceIntersection(hs, res.x[:3]).intersections
    return float(ConvexHull(pts).volume)


def calculate_obj_aligned_3d_IoU(pcd1, pcd2):
    """IoU of the object-aligned boxes of two clouds (IoU_ops.py:97-145); 0 when a box cannot be built ("OBB failure", :115-124)."""
    a, b = _pts(pcd1).T, _pts(pcd2).T
    try:
        b1, b2 = oriented_bounding_box(a), oriented_bounding_box(b)
    except Exception:
        return 0
    v1, v2 = 8 * np.prod(b1[2]), 8 * np.prod(b2[2])
    try:
        inter = oriented_box_intersection_volume(b1, b2)
    except Exception:
        return 0.
    union = v1 + v2 - inter
    return inter / union if union > 0 else 0.

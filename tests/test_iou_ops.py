"""utils/IoU_ops.py: the object-aligned box IoU `_recluster_IoU` uses by default (/root/reference/utils/IoU_ops.py:97-145 builds it
from Open3D's OrientedBoundingBox and the Objectron box IoU, both absent here) against closed-form cases."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from ibloc_amd.utils import IoU_ops as iou


def box_points(half, R=np.eye(3), t=np.zeros(3), n=4000, seed=0):
    rng = np.random.default_rng(seed)
    p = rng.uniform(-1, 1, size=(n, 3)) * half
    corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) * half
    return np.vstack([p, corners]) @ R.T + t


def test_obb_recovers_a_rotated_box():
    half = np.array([0.5, 0.3, 0.1])
    R = Rotation.from_euler("xyz", [0.3, -0.7, 1.1]).as_matrix()
    c, Rb, h = iou.oriented_bounding_box(box_points(half, R, np.array([1.0, -2.0, 0.5])))
    assert np.allclose(c, [1.0, -2.0, 0.5], atol=2e-2)
    assert np.allclose(np.sort(h), np.sort(half), atol=2e-2)
    assert np.allclose(np.abs(Rb.T @ R), np.eye(3), atol=6e-2)          # the same axes up to order / sign (sorted by variance here)
    assert np.isclose(np.linalg.det(Rb), 1.0)


def test_identical_disjoint_and_touching():
    a = box_points(np.array([0.4, 0.3, 0.2]), seed=1)
    assert iou.calculate_obj_aligned_3d_IoU(a, a.copy()) == pytest.approx(1.0, abs=1e-9)
    assert iou.calculate_obj_aligned_3d_IoU(a, a + np.array([5.0, 0, 0])) == 0
    assert iou.calculate_obj_aligned_3d_IoU(a, a + np.array([0.8, 0, 0])) == pytest.approx(0.0, abs=1e-6)     # sharing a face only
    assert iou.calculate_obj_aligned_3d_IoU(a[:3], a) == 0                # "OBB failure": three points have no hull


def test_axis_aligned_overlap_matches_the_closed_form():
    half = np.array([0.5, 0.5, 0.5])
    a = box_points(half, seed=2)
    b = box_points(half, t=np.array([0.5, 0.25, 0.0]), seed=3)
    inter = 0.5 * 0.75 * 1.0
    assert iou.calculate_obj_aligned_3d_IoU(a, b) == pytest.approx(inter / (2 - inter), rel=1e-6)
    assert iou.calculate_3d_IoU(a, b) == pytest.approx(inter / (2 - inter), rel=1e-6)                         # the axis-aligned measure agrees


def _clip_area(poly, clip):
    """area of the intersection of two convex polygons (Sutherland-Hodgman), an independent 2-D check of the 3-D polytope volume"""
    out = poly
    for i in range(len(clip)):
        a, b = clip[i], clip[(i + 1) % len(clip)]
        inp, out = out, []
        side = lambda p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        for j in range(len(inp)):
            p, q = inp[j], inp[(j + 1) % len(inp)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if sp * sq < 0:
                t = sp / (sp - sq)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
        if not out:
            return 0.0
    x, y = np.array(out).T
    return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


@pytest.mark.parametrize("deg", [90.0, 30.0, 61.0])
def test_rotated_prisms_match_polygon_clipping(deg):
    """two rectangular prisms (distinct side lengths: the PCA frame of a square is arbitrary), one turned about z: the intersection is
    the clipped rectangle times the common height; the axis-aligned measure of the same clouds is a different number"""
    ha, hb = np.array([0.5, 0.2, 0.3]), np.array([0.45, 0.15, 0.2])
    Rz = Rotation.from_euler("z", deg, degrees=True).as_matrix()
    a = box_points(ha, seed=4)
    b = box_points(hb, Rz, np.array([0.05, 0.02, 0.0]), seed=5)
    rect = lambda h, R, t: [tuple((R[:2, :2] @ np.array([sx * h[0], sy * h[1]])) + t[:2]) for sx, sy in ((-1, -1), (1, -1), (1, 1), (-1, 1))]
    inter = _clip_area(rect(ha, np.eye(3), np.zeros(3)), rect(hb, Rz, np.array([0.05, 0.02, 0.0]))) * 0.4
    va, vb = 8 * np.prod(ha), 8 * np.prod(hb)
    want = inter / (va + vb - inter)
    assert 0.05 < want < 0.9
    assert iou.calculate_obj_aligned_3d_IoU(a, b) == pytest.approx(want, rel=1e-6)
    if deg != 90.0:
        assert abs(iou.calculate_3d_IoU(a, b) - want) > 0.01


def test_default_measure_of_recluster_iou():
    """ObjectMemory._recluster_IoU falls back to the object-aligned measure when no iou_func is given, as the reference does
    (object_memory.py:710-747): fragments of one object overlap strongly, a distant object not at all"""
    a = box_points(np.array([0.4, 0.3, 0.1]), seed=6)
    rng = np.random.default_rng(7)
    i = rng.permutation(len(a))
    assert iou.calculate_obj_aligned_3d_IoU(a[i[: len(a) // 2]], a[i[len(a) // 2:]]) > 0.5
    assert iou.calculate_obj_aligned_3d_IoU(a, a + np.array([3.0, 0, 0])) == 0

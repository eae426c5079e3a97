"""Box overlap measures of the memory consolidation step, mirroring /root/reference/utils/IoU_ops.py on (N, 3) arrays or
anything with `.points`.

`calculate_3d_IoU` (:9-51) and `calculate_strict_overlap` (:53-95) are axis-aligned and pure numpy.  The object-aligned
`calculate_obj_aligned_3d_IoU` (:97-145) is built from two third-party pieces that are in neither this image nor the reference
tree -- Open3D's `OrientedBoundingBox.create_from_points` (Qhull hull + PCA) and the Objectron box-IoU (its submodule directory
is empty) -- so it is not restated here: pass an `iou_func` of your own to `ObjectMemory._recluster_IoU`.
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


def calculate_obj_aligned_3d_IoU(pcd1, pcd2):
    raise NotImplementedError("object-aligned IoU needs Open3D's OrientedBoundingBox and the Objectron box IoU (third-party, not in this "
                              "build): pass iou_func= to ObjectMemory._recluster_IoU / recluster_via_clustering_and_IoU")

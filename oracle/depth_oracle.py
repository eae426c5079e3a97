"""CPU restatement of the reference's depth -> coloured object clouds (TEST INFRASTRUCTURE ONLY: the checker of
`ibl_unproject_masks`; nothing under instance-based-loc_amd/ imports it).

Follows /root/reference/utils/depth_utils.py:46-90 (`get_coloured_pointcloud_from_depth`, before its outlier step) and
:176-206 (`get_mask_coloured_pointclouds_from_depth`: one call per mask on `depth_image * mask`) line by line -- the numpy
expressions are the reference's own (float32 linspace grid, float64 products), so this oracle is the reference's
arithmetic itself and needs no further pinning; the points are then rounded to float32 (the product's HBM layout) and the
intensity is the mean of the float32 colours, as `intensity_from_colors` defines it."""
import numpy as np


def coloured_pointcloud_from_depth(depth_image, rgb_image, fx, fy):
    assert depth_image.shape[:2] == rgb_image.shape[:2], "Depth and RGB image dimensions do not match"
    w, h = depth_image.shape                                                    # :55 (the names are swapped in the reference)
    horizontal = np.tile(np.linspace(-h / 2, h / 2, h, dtype=np.float32), (w, 1))          # :57,60
    vertical = np.tile(np.linspace(w / 2, -w / 2, w, dtype=np.float32).reshape(-1, 1), (1, h))   # :58,61
    X = horizontal * depth_image / fx                                           # :63
    Y = vertical * depth_image / fy                                             # :64
    pts = np.stack([X, Y, depth_image], axis=2).reshape(-1, 3)                  # :68
    valid = pts[:, 2] != 0                                                      # :71
    cols = (rgb_image.astype(np.float32) / 255.0).reshape(-1, 3)[valid]         # :75-76
    return pts[valid], cols


def mask_clouds(depth_image, rgb_image, masks, fx, fy, depth_factor=1.0):
    """-> list of (points float32 (n, 3), intensity float32 (n,)) per mask (depth_utils.py:198-204 with the depth scaled as
    the callers do, object_memory.py:150: `depth_image / depth_factor`)."""
    out = []
    for m in masks:
        m2 = np.asarray(m).reshape(depth_image.shape[:2])
        pts, cols = coloured_pointcloud_from_depth((depth_image / depth_factor) * m2, rgb_image, fx, fy)
        c = cols.astype(np.float64)
        out.append((pts.astype(np.float32), ((c[:, 0] + c[:, 1] + c[:, 2]) / 3.0).astype(np.float32)))
    return out

"""Loader of tests/golden/ref_scene/ -- data the reference ships (a saved 3-object memory and three RGB-D views of its synthetic
room; tools/gen_fixture_ref_scene.py) -- and the query frames the tests derive from it."""
import os

import numpy as np
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_scene")
FX = FY = 300.0


def memory_objects():
    z = np.load(os.path.join(DIR, "memory_objects.npz"))
    return [(z[f"obj{i}_xyz"], z[f"obj{i}_rgb"].astype(np.float64) / 255.0) for i in range(3)]


def views():
    """{view id: (depth float32 (600, 600), rgb u8 (600, 600, 3), pose [x y z qx qy qz qw])}; the pose is assembled the way the
    reference's SynthDataloader does (dataloader/synthetic_dataloader.py:49-59: Euler 'xyz' in degrees -> quaternion)"""
    z = np.load(os.path.join(DIR, "views.npz"))
    out = {}
    for k in z["view_ids"]:
        q = Rotation.from_euler("xyz", z["euler_xyz_deg"][k - 1], degrees=True).as_quat()
        out[int(k)] = (z[f"depth{k}"], z[f"rgb{k}"], np.concatenate([z["position"][k - 1], q]))
    return out


def pose_matrix(pose):
    T = np.eye(4)
    T[:3, :3] = Rotation.from_quat(pose[3:] / np.linalg.norm(pose[3:])).as_matrix()
    T[:3, 3] = pose[:3]
    return T


def object_masks(depth, pose, objects, tol=0.01):
    """one boolean mask per memory object: the pixels of the view whose unprojected world point lies within `tol` of the object's
    saved cloud (stands in for the SAM masks the reference's perception front end would produce)"""
    from oracle import depth_oracle as do
    h, w = depth.shape
    pts, _ = do.coloured_pointcloud_from_depth(depth, np.zeros((h, w, 3), np.uint8), FX, FY)
    valid = (depth.reshape(-1) != 0)
    T = pose_matrix(pose)
    world = pts.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    masks = []
    for p, _ in objects:
        d, _ = cKDTree(p).query(world, k=1)
        m = np.zeros(h * w, dtype=bool)
        m[np.nonzero(valid)[0][d < tol]] = True
        masks.append(m.reshape(h, w))
    return masks

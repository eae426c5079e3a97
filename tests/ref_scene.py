"""Loader of tests/golden/ref_scene/ -- data the reference ships (a saved 3-object memory and the eight RGB-D views of its synthetic
room; tools/gen_fixture_ref_scene.py) -- and the query frames the tests derive from it."""
import os

import numpy as np
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_scene")
FX = FY = 300.0
MIN_PIXELS = 400          # an object is a detection of a view when its mask covers at least this many pixels
VIEW_SEED = 11            # RANSAC seed of the stored oracle transcript (tools/gen_golden_ref_views.py)
EMB_DIM = 32


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


_DET_CACHE = {}


def view_detections(view, seen):
    """(raw clouds [(xyz float32, intensity)], pose) of the saved objects `seen` in one view -- the mask search of a view costs ~20 s on
    the CPU, so the result is shared by the tests of one pytest process"""
    key = (int(view), tuple(seen))
    if key not in _DET_CACHE:
        from oracle import depth_oracle as do
        objs = memory_objects()
        depth, rgb, pose = views()[view]
        masks = object_masks(depth, pose, objs)
        _DET_CACHE[key] = (do.mask_clouds(depth, rgb, [masks[j] for j in seen], FX, FY), pose)
    return _DET_CACHE[key]


def pose_error(pose7, T):
    """(translation error in metres, rotation error in radians) of a [x y z qx qy qz qw] pose against a 4x4 matrix"""
    R = Rotation.from_quat(pose7[3:]).as_matrix()
    return float(np.linalg.norm(np.asarray(pose7[:3]) - T[:3, 3])), float(np.arccos(np.clip((np.trace(R.T @ T[:3, :3]) - 1) / 2, -1, 1)))


def memory_embeddings():
    """synthetic identity embeddings of the three saved objects (the reference's own info.pkl files hold dummy [1, 2, 3] vectors):
    two stored views per object around a random base direction"""
    rng = np.random.default_rng(4)
    base = rng.normal(size=(3, EMB_DIM))
    emb = [(base[j][None] + rng.normal(0, 0.05, size=(2, EMB_DIM))).astype(np.float32) for j in range(3)]
    return base, emb


def view_frames():
    """{view id: dict(seen, clouds, ints, det_emb, pose)} for all eight views: the detections of a view are the saved objects whose mask
    covers >= MIN_PIXELS pixels, unprojected by the oracle's transcript of utils/depth_utils.py"""
    from oracle import depth_oracle as do
    objs = memory_objects()
    base, _ = memory_embeddings()
    out = {}
    for k, (depth, rgb, pose) in sorted(views().items()):
        masks = object_masks(depth, pose, objs)
        seen = [j for j in range(3) if masks[j].sum() >= MIN_PIXELS]
        cl = do.mask_clouds(depth, rgb, [masks[j] for j in seen], FX, FY)
        rng = np.random.default_rng(100 + k)
        out[k] = dict(seen=seen, clouds=[c[0] for c in cl], ints=[c[1] for c in cl], pose=pose,
                      det_emb=(base[seen] + rng.normal(0, 0.05, size=(len(seen), EMB_DIM))).astype(np.float32))
    return out


def oracle_assignments(det_emb):
    from oracle import match_oracle as mo
    from oracle import simvolume_oracle as so
    _, emb = memory_embeddings()
    off = (np.arange(4) * 2).astype(np.int32)
    sims = mo.closest_similarity(mo.normalize_rows(det_emb), mo.normalize_rows(np.concatenate(emb)), off)
    return so.simvolume_assignments(sims, 4)


def oracle_views():
    import json
    return json.load(open(os.path.join(DIR, "oracle_views.json")))

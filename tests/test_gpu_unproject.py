"""-m gpu: ibl_unproject_masks (depth + masks -> coloured clouds) against the numpy transcript of
utils/depth_utils.py:46-90,176-206 in oracle/depth_oracle.py.  Bar: bit-exact float32 points and intensities, identical
cloud sizes and point order (integer / byte work and correctly rounded fp64 arithmetic on both sides)."""
import numpy as np
import pytest
import torch

from oracle import depth_oracle as do

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ibloc_amd.registration import RegContext
    c = RegContext(1 << 30)
    yield c
    c.close()


def _scene(rng, H, W, n_masks, u16):
    if u16:
        depth = rng.integers(0, 40000, size=(H, W)).astype(np.uint16)
        depth[rng.random((H, W)) < 0.2] = 0                                   # missing depth
    else:
        depth = rng.uniform(0.3, 6.0, size=(H, W)).astype(np.float32)
        depth[rng.random((H, W)) < 0.2] = 0.0
    rgb = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
    masks = rng.random((n_masks, H, W)) < 0.3
    if n_masks > 1:
        masks[1] = False                                                      # an empty mask -> an empty cloud
    return depth, rgb, masks


@pytest.mark.parametrize("H,W,n_masks,u16,factor", [(48, 64, 3, False, 1.0), (37, 53, 4, True, 5000.0), (1, 1, 1, False, 1.0),
                                                    (480, 640, 7, True, 5000.0), (5, 7, 0, False, 1.0), (33, 41, 2, False, 2.5),
                                                    (24, 31, 2, "f64", 1.7)])
def test_unproject_bit_exact(ctx, H, W, n_masks, u16, factor):
    from ibloc_amd.registration import unproject_masks
    rng = np.random.default_rng(H * 1000 + W)
    depth, rgb, masks = _scene(rng, H, W, n_masks, u16 is True)
    if u16 == "f64":
        depth, u16 = depth.astype(np.float64) * 1.0000001, False
    fx, fy = 525.0, 519.3
    d_t = torch.from_numpy(depth.view(np.int16)).cuda().view(torch.uint16) if u16 else torch.from_numpy(depth).cuda()
    got = unproject_masks(ctx, d_t, torch.from_numpy(rgb).cuda(), torch.from_numpy(masks).cuda(), fx, fy, factor)
    want = do.mask_clouds(depth, rgb, masks, fx, fy, factor)
    assert got.n_seg == n_masks
    p4 = got.pts4.cpu().numpy()
    for m, (pts, inten) in enumerate(want):
        b, e = got.seg_off_host[m], got.seg_off_host[m + 1]
        assert e - b == len(pts), (m, e - b, len(pts))
        assert np.array_equal(p4[b:e, :3], pts)
        assert np.array_equal(p4[b:e, 3], inten)
    if n_masks > 1:
        assert got.seg_off_host[2] == got.seg_off_host[1]                     # the empty mask


def test_unproject_feeds_outlier_removal(ctx):
    """the clouds go straight into the registration path: radius outlier removal of the unprojected clouds == oracle"""
    from ibloc_amd.registration import radius_outlier_batch, unproject_masks
    from oracle import reg_oracle as ro
    rng = np.random.default_rng(9)
    H, W = 120, 160
    yy, xx = np.mgrid[0:H, 0:W]
    depth = (1.0 + 0.002 * xx + 0.001 * yy).astype(np.float32)               # a slanted plane, ~2 mm pixel spacing
    depth[rng.random((H, W)) < 0.05] = 0.0
    rgb = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
    masks = np.zeros((2, H, W), dtype=bool)
    masks[0, 10:70, 10:90] = True
    masks[1, 60:110, 80:150] = True
    got = unproject_masks(ctx, torch.from_numpy(depth).cuda(), torch.from_numpy(rgb).cuda(), torch.from_numpy(masks).cuda(), 300.0, 300.0)
    keep = radius_outlier_batch(ctx, got, 0.05, 12).cpu().numpy().astype(bool)
    want = do.mask_clouds(depth, rgb, masks, 300.0, 300.0)
    for m, (pts, _) in enumerate(want):
        b, e = got.seg_off_host[m], got.seg_off_host[m + 1]
        assert np.array_equal(keep[b:e], ro.radius_outlier(pts, 0.05, 12))


def test_depth_utils_facade_matches_numpy_transcript():
    """utils.depth_utils (same names as the reference's module): clouds = the numpy transcript's values after radius outlier removal,
    voxel down-sampling = the python-dict restatement, pose helpers = the reference's expressions"""
    from scipy.spatial.transform import Rotation
    from ibloc_amd.utils import depth_utils as du
    from oracle import build_oracle as bo
    from oracle import depth_oracle as do
    from oracle import reg_oracle as ro
    rng = np.random.default_rng(21)
    H, W = 60, 84
    rgb = rng.integers(0, 255, size=(H, W, 3), dtype=np.uint8)
    for depth in ((1.5 + 0.4 * rng.random((H, W))).astype(np.float32), np.round(5000 * (1.5 + 0.4 * rng.random((H, W)))).astype(np.uint16) / 5000.0):
        depth = depth.copy()
        depth[rng.random((H, W)) < 0.05] = 0
        cfg = {"radius_nb_points": 3, "radius": 0.08}
        full = du.get_coloured_pointcloud_from_depth(depth, rgb, 70.0, 65.0, cfg)
        pts, cols = do.coloured_pointcloud_from_depth(depth, rgb, 70.0, 65.0)
        keep = ro.radius_outlier(pts.astype(np.float32), cfg["radius"], cfg["radius_nb_points"])
        assert np.array_equal(full.points, pts[keep].astype(np.float64)) and np.array_equal(full.colors, cols[keep].astype(np.float64))
        plain = du.get_pointcloud_from_depth(depth, 70.0, 65.0, None)
        assert np.array_equal(plain.points, pts.astype(np.float64)) and plain.colors is None
        masks = np.zeros((2, H, W), dtype=bool)
        masks[0, 5:40, 3:50] = True
        masks[1, 30:58, 40:80] = True
        got = du.get_mask_coloured_pointclouds_from_depth(depth, rgb, masks, 70.0, 65.0, None)
        for g, m in zip(got, masks):
            p, c = do.coloured_pointcloud_from_depth(depth * m, rgb, 70.0, 65.0)
            assert np.array_equal(g.points, p.astype(np.float64)) and np.array_equal(g.colors, c.astype(np.float64))
        ds = du.voxel_down_sample_with_colors(got[0], 0.05)
        wp, wc, _ = bo.voxel_down_sample_with_colors(got[0].points, got[0].colors, 0.05)
        assert np.array_equal(ds.points, wp) and np.array_equal(ds.colors, wc)
    pose = np.array([0.2, -0.1, 0.7, 0.3, -0.2, 0.1, 0.8])
    q = pose[3:] / np.linalg.norm(pose[3:])
    R, R2 = Rotation.from_quat(q).as_matrix(), Rotation.from_euler('xyz', [0, np.pi, 0]).as_matrix()
    t1 = du.transform_pointcloud(got[1], pose.copy())
    assert np.array_equal(t1.points, (R @ got[1].points.T).T + pose[:3]) and np.array_equal(t1.colors, got[1].colors)
    t2 = du.transform_pointcloud_kinect(got[1], pose.copy())
    assert np.array_equal(t2.points, (R @ R2 @ got[1].points.T).T - pose[:3])
    both = du.combine_point_clouds(got)
    assert len(both.points) == len(got[0].points) + len(got[1].points) and np.array_equal(du.compute_center(both), both.points.mean(axis=0))
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, [1, 2, 3]
    d = du.decompose_pose_matrix(T)
    assert np.allclose(d[:3], [1, 2, 3]) and (np.allclose(d[3:], q) or np.allclose(d[3:], -q))

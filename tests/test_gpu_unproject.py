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

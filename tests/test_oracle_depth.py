"""oracle/depth_oracle.py known answers (the oracle is the reference's numpy expressions, utils/depth_utils.py:55-76)."""
import numpy as np

from oracle import depth_oracle as do


def test_centred_pixel_grid_known_answer():
    depth = np.ones((2, 3), dtype=np.float32)
    depth[1, 1] = 0.0                                              # dropped
    rgb = np.arange(18, dtype=np.uint8).reshape(2, 3, 3) * 10
    pts, cols = do.coloured_pointcloud_from_depth(depth, rgb, 1.0, 1.0)
    # columns: linspace(-3/2, 3/2, 3) = -1.5, 0, 1.5; rows: linspace(2/2, -2/2, 2) = 1, -1 (the reference's w/h are swapped names)
    want = np.array([[-1.5, 1, 1], [0, 1, 1], [1.5, 1, 1], [-1.5, -1, 1], [1.5, -1, 1]], dtype=np.float64)
    assert np.array_equal(pts, want)
    assert cols.dtype == np.float32 and np.array_equal(cols[0], np.array([0, 10, 20], dtype=np.float32) / np.float32(255.0))
    assert np.array_equal(cols[3], rgb[1, 0].astype(np.float32) / np.float32(255.0))


def test_mask_clouds_scale_order_and_dtype_promotion():
    rng = np.random.default_rng(0)
    depth16 = rng.integers(1, 30000, size=(6, 5)).astype(np.uint16)
    rgb = rng.integers(0, 256, size=(6, 5, 3)).astype(np.uint8)
    masks = np.zeros((2, 6, 5), dtype=bool)
    masks[0, 1:3, 2:4] = True
    out = do.mask_clouds(depth16, rgb, masks, 500.0, 400.0, 5000.0)
    assert len(out[0][0]) == 4 and len(out[1][0]) == 0
    z = depth16[1:3, 2:4].reshape(-1).astype(np.float64) / 5000.0
    assert np.array_equal(out[0][0][:, 2], z.astype(np.float32))                  # row-major order, float64 depth -> float32
    col = np.linspace(-2.5, 2.5, 5, dtype=np.float32)[[2, 3, 2, 3]].astype(np.float64)
    assert np.array_equal(out[0][0][:, 0], (col * z / 500.0).astype(np.float32))
    # float32 depth: numpy keeps the whole expression in float32
    d32 = (depth16.astype(np.float32) / np.float32(5000.0))
    o32 = do.mask_clouds(d32, rgb, masks, 500.0, 400.0)
    x32 = (np.linspace(-2.5, 2.5, 5, dtype=np.float32)[[2, 3, 2, 3]] * d32[1:3, 2:4].reshape(-1)) / np.float32(500.0)
    assert x32.dtype == np.float32 and np.array_equal(o32[0][0][:, 0], x32)
    inten = out[0][1]
    c = (rgb[1, 2].astype(np.float32) / np.float32(255.0)).astype(np.float64)
    assert inten[0] == np.float32((c[0] + c[1] + c[2]) / 3.0)

"""-m gpu: memory-build kernels against the oracle: voxel down-sampling bit for bit, DBSCAN labels exactly."""
import numpy as np
import pytest

from oracle import build_oracle as bo

pytestmark = pytest.mark.gpu


def _ctx():
    from ibloc_amd.registration import RegContext
    return RegContext(4 << 30)


def test_voxel_downsample_bit_exact():
    from ibloc_amd.build import voxel_downsample_batch
    rng = np.random.default_rng(5)
    pts, cols = [], []
    for k, (n, box, shift) in enumerate([(5000, 0.4, 0.0), (1, 0.1, 3.0), (0, 0.1, 0.0), (12000, 0.25, -7.5), (300, 0.02, 100.0), (2500, 1.5, 0.0)]):
        p = rng.uniform(-box, box, size=(n, 3)) + shift
        if k == 3:
            p[::7] = p[::7][::-1]                                       # scrambled order inside voxels
            p[5:50] = p[4]                                              # exact duplicates
        pts.append(p)
        cols.append(rng.uniform(0, 1, size=(n, 3)))
    ctx = _ctx()
    for voxel in (0.05, 0.005, 0.013):
        gp, gc, gn = voxel_downsample_batch(ctx, pts, cols, voxel, return_counts=True)
        for i in range(len(pts)):
            wp, wc, wn = bo.voxel_down_sample_with_colors(pts[i], cols[i], voxel)
            assert gp[i].shape == wp.shape, (voxel, i)
            assert np.array_equal(gp[i], wp) and np.array_equal(gc[i], wc) and np.array_equal(gn[i], wn), (voxel, i)
    gp, gc = voxel_downsample_batch(ctx, pts, None, 0.05)
    assert gc is None and np.array_equal(gp[0], bo.voxel_down_sample_with_colors(pts[0], None, 0.05)[0])
    gp, gc = voxel_downsample_batch(ctx, [], [], 0.05)
    assert gp == [] and gc == []
    # idempotence at the same voxel size is NOT a property (means move inside voxels); monotone point counts are
    g2, _ = voxel_downsample_batch(ctx, gp if gp else pts, None, 0.1)
    assert all(len(a) <= len(b) for a, b in zip(g2, pts))
    with pytest.raises(RuntimeError):
        voxel_downsample_batch(ctx, [np.array([[0.0, 0, 0], [1000.0, 0, 0]])], None, 0.005)      # > 65536 voxels along x
    ctx.close()


def test_dbscan_labels_equal_sequential_scan():
    from ibloc_amd.build import dbscan_batch
    from tests.test_oracle_build import blobs
    rng = np.random.default_rng(6)
    groups = []
    for k in range(5):
        P = blobs(rng, 3 + k, 150, 0.05, 0.8 + 0.3 * k) + rng.uniform(-20, 20, size=3)
        rng.shuffle(P)
        groups.append(P)
    groups.insert(2, np.zeros((0, 3)))                                  # an empty group
    groups.append(rng.uniform(-5, 5, size=(40, 3)))                     # all noise
    groups.append(np.repeat(rng.uniform(-1, 1, size=(3, 3)), 30, axis=0))   # exact duplicates: three clusters of one location each
    ctx = _ctx()
    for eps, mp in [(0.1, 10), (0.15, 25), (0.05, 2), (0.2, 1)]:
        labels, ncl = dbscan_batch(ctx, groups, eps, mp)
        for gi, P in enumerate(groups):
            want = bo.cluster_dbscan(P, eps, mp) if len(P) else np.zeros(0, dtype=np.int32)
            assert np.array_equal(labels[gi], want), (eps, mp, gi)
            assert ncl[gi] == (want.max() + 1 if len(want) and want.max() >= 0 else 0)
    # a surface-like cloud at the density of a down-sampled object (the shape the consolidation step clusters)
    u = rng.uniform(0, 1, size=(6000, 2))
    S = np.stack([u[:, 0], u[:, 1], 0.05 * np.sin(6 * u[:, 0])], axis=1)
    S = np.concatenate([S, S[:2500] + [1.08, 0, 0]])
    labels, ncl = dbscan_batch(ctx, [S], 0.05, 20)
    assert np.array_equal(labels[0], bo.cluster_dbscan(S, 0.05, 20)) and ncl[0] >= 1
    assert dbscan_batch(ctx, [], 0.1, 5)[0] == []
    ctx.close()

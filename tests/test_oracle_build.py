"""Oracle of the memory-build numerics: DBSCAN restatement against scikit-learn, voxel restatement against first principles."""
import numpy as np
from sklearn.cluster import DBSCAN

from oracle import build_oracle as bo


def blobs(rng, n_blobs, pts, spread, box):
    c = rng.uniform(-box, box, size=(n_blobs, 3))
    return np.concatenate([c[i] + rng.normal(size=(pts, 3)) * spread for i in range(n_blobs)] + [rng.uniform(-box, box, size=(pts // 2, 3))])


def test_dbscan_restatement_equals_sklearn():
    rng = np.random.default_rng(0)
    for case, (eps, mp) in enumerate([(0.12, 8), (0.2, 20), (0.08, 3), (0.3, 1), (0.05, 50)]):
        P = blobs(rng, 5, 160, 0.06, 1.0)
        rng.shuffle(P)
        got = bo.cluster_dbscan(P, eps, mp)
        want = DBSCAN(eps=eps, min_samples=mp, algorithm="brute").fit(P).labels_
        assert np.array_equal(got, want), case
        if mp == 1:
            assert got.min() == 0                       # every point is a core point: no noise


def test_voxel_restatement_properties():
    rng = np.random.default_rng(1)
    P = rng.uniform(-0.3, 0.3, size=(4000, 3))
    C = rng.uniform(0, 1, size=(4000, 3))
    dp, dc, cnt = bo.voxel_down_sample_with_colors(P, C, 0.05)
    assert cnt.sum() == 4000 and len(dp) == len(dc) == len(cnt)
    keys = np.floor(P / 0.05).astype(np.int64)
    uniq, first = np.unique(keys, axis=0, return_index=True)
    assert len(uniq) == len(dp)
    order = np.argsort(first)                                            # voxels in order of first occurrence
    assert np.array_equal(np.floor(dp / 0.05).astype(np.int64), uniq[order])
    k = 7
    members = np.flatnonzero((keys == uniq[order][k]).all(axis=1))
    assert np.allclose(dp[k], P[members].mean(axis=0)) and np.allclose(dc[k], C[members].mean(axis=0))
    dp1, dc1, _ = bo.voxel_down_sample_with_colors(P[:1], None, 0.05)
    assert np.array_equal(dp1, P[:1]) and dc1 is None

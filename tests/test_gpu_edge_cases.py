"""-m gpu: edge cases the reference's inputs can produce -- frames without detections, single detections, empty or
tiny clouds (everything removed by the outlier filter), duplicated points, zero-length ranges."""
import numpy as np
import pytest
import torch

from ibloc_amd.synth import SynthWorld
from oracle import reg_oracle as ro

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ibloc_amd.registration import RegContext
    c = RegContext(4 << 30)
    yield c
    c.close()


def test_engine_with_empty_and_single_detection_frames(ctx):
    from ibloc_amd.engine import LocaliseEngine, MemoryShard, intensity_from_colors
    from ibloc_amd.registration import CloudBatch
    w = SynthWorld(5, pts_per_object=3000, E=2, D=32, seed=61)
    eng = LocaliseEngine(MemoryShard(ctx, list(w.embeddings), w.points, colors=w.colors))
    rng = np.random.default_rng(62)
    f1 = w.make_frame(rng, q=1, pts_per_object=3000, anchor=2)
    f2 = w.make_frame(rng, q=2, pts_per_object=3000, anchor=0)
    clouds, ints, embs = [], [], []
    for f in (f1, f2):
        for (p, c) in f["clouds"]:
            clouds.append(p)
            ints.append(intensity_from_colors(c))
        embs.append(f["det_emb"])
    # one extra detection whose cloud is so sparse that the outlier filter removes every point
    sparse = rng.uniform(-5, 5, size=(40, 3))
    clouds.append(sparse)
    ints.append(np.zeros(40, np.float32))
    embs.append(w.embeddings[3][:1])
    det = CloudBatch.from_numpy(clouds, ints)
    res = eng.localise_batch(det, [0, 1, 2, 1], det_emb=np.concatenate(embs), fpfh_global_dist_factor=1.5, fpfh_local_dist_factor=1.5)
    assert np.array_equal(res[0].pose, [0, 0, 0, 0, 0, 0, 1]) and res[0].assignments == []      # object_memory.py:895-896
    assert len(res[1].assignments) == 1 and res[1].assignments[0][0][0] == 0                    # Q = 1 -> one length-1 assignment
    assert len(res[2].assignments) == 3
    assert all(np.isfinite(r.pose).all() for r in res)
    assert res[3].records[0]["fitness"] == 0.0                                                  # empty cleaned cloud: nothing to register


def test_register_batch_with_empty_and_tiny_clouds(ctx):
    from ibloc_amd.registration import CloudBatch, register_batch
    w = SynthWorld(1, pts_per_object=1500, E=1, D=8, seed=63)
    good = (w.points[0] - w.points[0].mean(0)).astype(np.float32)
    col = ro.intensity(w.colors[0])
    det = CloudBatch.from_numpy([good, np.zeros((0, 3), np.float32), good[:2]], [col, np.zeros(0, np.float32), col[:2]])
    mem = CloudBatch.from_numpy([good, good[:1]], [col, col[:1]])
    out = register_batch(ctx, det, mem, [[0, -1, -1], [1, -1, -1], [2, -1, -1], [0, 1, -1]], [[0, -1, -1], [0, -1, -1], [1, -1, -1], [0, 1, -1]],
                         0.05, 1.5, 1.5, seed=1)
    assert np.isfinite(out["T"]).all()
    assert out["fitness"][0] > 0.99 and np.allclose(out["T"][0], np.eye(4), atol=2e-3)      # identical clouds: identity
    assert out["fitness"][1] == 0.0 and np.allclose(out["T"][1], np.eye(4))                   # empty source
    assert np.allclose(out["T"][2][:3, :3] @ out["T"][2][:3, :3].T, np.eye(3), atol=1e-9)      # 2-point source vs 1-point target


def test_normals_fpfh_on_degenerate_clouds(ctx):
    from ibloc_amd.registration import CloudBatch, normals_fpfh_batch
    dup = np.tile(np.array([[0.1, 0.2, 0.3]], np.float32), (50, 1))                # 50 identical points
    line = np.stack([np.linspace(0, 1, 200), np.zeros(200), np.zeros(200)], 1).astype(np.float32)
    two = np.array([[0, 0, 0], [0.01, 0, 0]], np.float32)
    b = CloudBatch.from_numpy([dup, line, two])
    nrm, fpfh = normals_fpfh_batch(ctx, b, 0.1, 30, 0.25, 100)
    nrm, fpfh = nrm.cpu().numpy(), fpfh.cpu().numpy()
    assert np.isfinite(nrm).all() and np.isfinite(fpfh).all()
    off = b.seg_off_host
    for i, c in enumerate([dup, line, two]):
        en = ro.normals(c, 0.1, 30)
        ef = ro.fpfh(c, en, 0.25, 100)
        gn, gf = nrm[off[i]:off[i + 1], :3], fpfh[off[i]:off[i + 1]]
        assert np.allclose(np.abs(np.sum(gn * en, 1)), 1.0, atol=1e-4) or i == 1      # a line has no unique normal
        assert np.mean(np.abs(gf - ef).max(1) < 2e-3) > 0.95, i
    assert np.allclose(nrm[off[2]:off[3], :3], [[0, 0, 1], [0, 0, 1]])               # < 3 neighbours -> (0, 0, 1)


def test_evaluate_zero_length_range_and_far_transform(ctx):
    from ibloc_amd.registration import CloudBatch, MemGrid, evaluate_batch
    w = SynthWorld(2, pts_per_object=1000, E=1, D=8, seed=64)
    mem = CloudBatch.from_numpy(w.points)
    grid = MemGrid(ctx, mem.pts4, 0.04)
    det = CloudBatch.from_numpy([w.points[0][:300]])
    far = np.eye(4)
    far[:3, 3] = 1e4
    rmse, fit = evaluate_batch(ctx, grid, det.pts4, [0, 0, 5], [300, 300, 5], np.stack([np.eye(4), far, np.eye(4)]), 0.02)
    assert fit[0] == 1.0 and rmse[0] == 0.0 and fit[1] == 0.0 and fit[2] == 0.0
    grid.close()


def test_many_spread_out_jobs_fit_the_icp_grid_budget(ctx):
    """Jobs whose target instances are far apart (a wrong assignment on real data) span tens of metres; a batch of them used to exceed the
    128 M-cell budget of the ICP grid at the nominal 3.75 cm cell and fail the whole call.  The cell now grows until the batch fits; the
    neighbour search is exact at any cell size, so every job still returns what it returns in a batch of its own."""
    from ibloc_amd.registration import CloudBatch, instance_features_batch, register_batch
    w = SynthWorld(4, pts_per_object=1500, E=1, D=8, seed=91)
    rng = np.random.default_rng(92)
    far = [np.float32([0, 0, 0]), np.float32([60, 0, 0]), np.float32([0, 70, 0]), np.float32([55, 65, 20])]
    mem_clouds = [w.points[i].astype(np.float32) + far[i] for i in range(4)]
    ints = [rng.uniform(0.1, 0.9, size=len(c)).astype(np.float32) for c in mem_clouds]
    mem = CloudBatch.from_numpy(mem_clouds, ints)
    det = CloudBatch.from_numpy([c + np.float32([0.02, -0.01, 0.03]) for c in mem_clouds], ints)
    mf = instance_features_batch(ctx, mem, 0.05, 2 * 0.05 * 1.5)
    df = instance_features_batch(ctx, det, 0.05)
    J = 72                                                         # 72 sides of ~129^3 cells each: 150 M cells at the nominal cell
    src = [[j % 4, (j + 1) % 4, -1] for j in range(J)]
    tgt = [[j % 4, (j + 1) % 4, -1] for j in range(J)]
    ids = np.arange(500, 500 + J, dtype=np.uint32)
    kw = dict(voxel_size=0.05, global_dist_factor=1.5, local_dist_factor=1.5, seed=9, det_features=df, mem_features=mf)
    big = register_batch(ctx, det, mem, src, tgt, job_ids=ids, **kw)
    assert ctx.status() & 1 == 0                                   # no grid overflow flag
    for j in (0, 1, 37, 71):
        one = register_batch(ctx, det, mem, [src[j]], [tgt[j]], job_ids=ids[j:j + 1], **kw)
        assert np.array_equal(one["T"][0], big["T"][j]) and one["fitness"][0] == big["fitness"][j]
        assert big["fitness"][j] > 0.5

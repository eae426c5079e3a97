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


def _fragments(seed):
    """12 box-surface objects observed as 2-3 overlapping fragments each, with noisy copies of the object's embedding"""
    rng = np.random.default_rng(seed)
    frags = []
    for k in range(12):
        c = np.array([(k % 4) * 1.5, (k // 4) * 1.5, 0.0]) + rng.uniform(-0.1, 0.1, size=3)
        u = rng.uniform(-1, 1, size=(3000, 3))
        ax = rng.integers(0, 3, size=3000)
        u[np.arange(3000), ax] = np.sign(u[np.arange(3000), ax])
        pts = c + u * np.array([0.25, 0.2, 0.3])
        cols = np.clip(0.5 + 0.4 * np.sin(pts * 5 + k), 0, 1)
        emb = rng.normal(size=48)
        for f in range(2 + k % 2):
            sel = rng.random(3000) < 0.7
            frags.append((f"obj{k}" if f else f"thing{k}", emb + 0.05 * rng.normal(size=48), pts[sel] + rng.normal(size=(sel.sum(), 3)) * 1e-3, cols[sel]))
    return frags


def _same_memory(mem, want):
    assert len(mem.memory) == len(want)
    for a, b in zip(mem.memory, want):
        assert a.names == b.names
        assert len(a.embeddings) == len(b.embeddings) and all(np.array_equal(x, y) for x, y in zip(a.embeddings, b.embeddings))
        assert np.array_equal(a.pointcloud.points, b.points) and np.array_equal(a.pointcloud.colors, b.colors)


def test_consolidation_matches_reference_transcript():
    """downsample_all_objects -> _recluster_IoU -> recluster_via_clustering_and_IoU (the driver's sequence, tum_localisation_trial.py:138-148)
    and recluster_objects_with_dbscan against the oracle transcript: identical objects, names, embeddings and clouds"""
    from ibloc_amd.object_memory.object_memory import ObjectMemory
    from ibloc_amd.utils.IoU_ops import calculate_3d_IoU
    for variant in ("driver", "dbscan"):
        mem = ObjectMemory(device="cuda", get_embeddings_func=lambda **kw: None, log_enabled=False, arena_bytes=2 << 30)
        want = []
        for name, emb, p, c in _fragments(11):
            mem.add_object(name, [emb], p, c)
            want.append(bo.Obj(name, emb, p, c))
        mem.downsample_all_objects(voxel_size=0.02)
        bo.downsample_all(want, 0.02)
        _same_memory(mem, want)
        if variant == "driver":
            mem._recluster_IoU(0.3, iou_func=calculate_3d_IoU)
            want = bo.recluster_IoU(want, 0.3, bo.aabb_iou)
            _same_memory(mem, want)
            mem.recluster_via_clustering_and_IoU(eps=0.08, embedding_distance_threshold=0.5, IoU_threshold=0.25, min_points_per_cluster=20,
                                                 iou_func=calculate_3d_IoU)
            want = bo.recluster_via_clustering_and_IoU(want, 0.5, 0.08, 20, 0.25, bo.aabb_iou)
        else:
            mem.recluster_objects_with_dbscan(eps=0.08, min_points_per_cluster=20)
            want = bo.recluster_objects_with_dbscan(want, 0.08, 20)
        _same_memory(mem, want)
        assert 8 <= len(mem.memory) <= 16 and [o.id for o in mem.memory] == list(range(len(mem.memory)))
        # the default measure of _recluster_IoU is the object-aligned box IoU (IoU_ops.py:97-145), as in the reference's driver
        from ibloc_amd.utils.IoU_ops import calculate_obj_aligned_3d_IoU
        mem._recluster_IoU(0.3)
        want = bo.recluster_IoU(want, 0.3, calculate_obj_aligned_3d_IoU)
        _same_memory(mem, want)
        mem._ctx.close()
    # recluster_via_agglomerative_clustering (:379-437): embedding clusters only
    mem = ObjectMemory(device="cuda", get_embeddings_func=lambda **kw: None, log_enabled=False, arena_bytes=1 << 30)
    want = []
    for name, emb, p, c in _fragments(11):
        mem.add_object(name, [emb], p, c)
        want.append(bo.Obj(name, emb, p, c))
    mem.recluster_via_agglomerative_clustering(embedding_distance_threshold=0.5)
    want = bo.recluster_via_agglomerative_clustering(want, 0.5)
    _same_memory(mem, want)
    assert 6 <= len(mem.memory) <= 12
    mem._ctx.close()


def test_process_detections_and_floor_removal():
    from ibloc_amd.object_memory.object_memory import ObjectMemory
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(3)
    mem = ObjectMemory(device="cuda", get_embeddings_func=lambda **kw: None, log_enabled=False, arena_bytes=1 << 30)
    pose = np.concatenate([[0.3, -1.0, 2.0], 2.0 * Rotation.from_euler("xyz", [0.2, -0.4, 1.0]).as_quat()])      # un-normalised quaternion
    clouds = [(rng.normal(size=(800, 3)), rng.uniform(size=(800, 3))), (rng.normal(size=(100, 3)), rng.uniform(size=(100, 3))),
              (rng.normal(size=(600, 3)) * [3, 0.01, 3], rng.uniform(size=(600, 3)))]
    embs = rng.normal(size=(3, 16))
    p0 = pose.copy()
    mem.process_detections(["mug", "pen", "floor"], embs, clouds, pose, min_points=500)
    assert abs(np.linalg.norm(pose[3:]) - 1) < 1e-12                      # normalised in place, like transform_pointcloud
    assert [o.names for o in mem.memory] == [["mug"]] and mem.floors is not None and mem.floors.names == ["floor"]
    R = Rotation.from_quat(p0[3:] / np.linalg.norm(p0[3:])).as_matrix()
    assert np.array_equal(mem.memory[0].pointcloud.points, (R @ clouds[0][0].T).T + p0[:3])
    mem.add_object("low", [embs[1]], np.array([[0, -5.0, 0], [0, -4.95, 0]]), np.zeros((2, 3)))
    mem.remove_points_below_floor()                                       # floor height = -5, thickness 0.1: "low" loses all points
    assert [o.names for o in mem.memory] == [["mug"]]
    assert mem.memory[0].pcd[1].min() > -5.0 + 0.1
    mem._ctx.close()


def test_process_image_through_stub_finder(tmp_path):
    """process_image (object_memory.py:163-256) with a stub ObjectFinder: clouds = the numpy transcript's points (at the float32 of
    the HBM layout) and colours after radius outlier removal, moved by the pose; min_points and the floor slot respected"""
    import torch
    from PIL import Image
    from ibloc_amd.object_memory.object_memory import ObjectMemory
    from oracle import depth_oracle as do
    from oracle import reg_oracle as ro
    rng = np.random.default_rng(8)
    H, W = 72, 96
    rgb = rng.integers(0, 255, size=(H, W, 3), dtype=np.uint8)
    depth = (2.0 + 0.3 * np.sin(np.arange(W)[None, :] / 9.0) + 0.2 * np.cos(np.arange(H)[:, None] / 7.0)).astype(np.float32)
    depth[rng.random((H, W)) < 0.03] = 0.0                               # holes
    Image.fromarray(rgb).save(tmp_path / "rgb.png")
    np.save(tmp_path / "depth.npy", depth)
    masks = torch.zeros((3, 1, H, W), dtype=torch.bool)
    masks[0, 0, 5:50, 5:60] = True
    masks[1, 0, 40:70, 50:90] = True
    masks[2, 0, 60:64, 2:8] = True                                       # too few points: skipped
    embs = rng.normal(size=(3, 24)).astype(np.float32)
    calls = []

    def emb_func(**kw):
        calls.append(kw["current_obj_phrase"])
        return torch.from_numpy(embs[len(calls) - 1])

    mem = ObjectMemory("cuda", None, None, 90.0, 85.0, get_embeddings_func=emb_func, log_enabled=False, arena_bytes=1 << 30,
                       object_finder=lambda path, floor: ([rgb[5:50, 5:60], rgb[40:70, 50:90], rgb[60:64, 2:8]],
                                                          [torch.zeros(4)] * 3, [m for m in masks], ["lamp", "floor", "crumb"]))
    pose = np.array([0.5, 0.25, -1.0, 0.1, 0.2, -0.3, 0.9])
    cfg = {"radius_nb_points": 4, "radius": 0.12}
    mem.process_image(str(tmp_path / "rgb.png"), str(tmp_path / "depth.npy"), pose.copy(), consider_floor=False,
                      outlier_removal_config=cfg, min_points=200, depth_factor=1.0)
    assert calls == ["lamp", "floor", "crumb"]
    assert [o.names for o in mem.memory] == [["lamp"]] and mem.floors.names == ["floor"]
    from scipy.spatial.transform import Rotation
    R = Rotation.from_quat(pose[3:] / np.linalg.norm(pose[3:])).as_matrix()
    for obj, mi in ((mem.memory[0], 0), (mem.floors, 1)):
        pts, cols = do.coloured_pointcloud_from_depth(depth * masks[mi, 0].numpy(), rgb, 90.0, 85.0)
        p32 = pts.astype(np.float32)
        keep = ro.radius_outlier(p32, cfg["radius"], cfg["radius_nb_points"])
        assert np.array_equal(obj.pointcloud.points, (R @ p32[keep].astype(np.float64).T).T + pose[:3])
        assert np.array_equal(obj.pointcloud.colors, cols[keep].astype(np.float64))
        assert np.array_equal(obj.embeddings[0], embs[mi])
    with pytest.raises(NotImplementedError):
        mem.process_image("a", "b", pose, False, will_cluster_later=False)
    # a 16-bit depth image with the Kinect factor: the reference's clouds are float64 products, and so are the memory's
    d16 = np.round(depth * 5000.0).astype(np.uint16)
    np.save(tmp_path / "depth16.npy", d16)
    calls.clear()
    mem.memory, mem.floors = [], None
    mem.process_image(str(tmp_path / "rgb.png"), str(tmp_path / "depth16.npy"), pose.copy(), consider_floor=False,
                      outlier_removal_config=cfg, min_points=200, depth_factor=5000.)
    for obj, mi in ((mem.memory[0], 0), (mem.floors, 1)):
        pts, cols = do.coloured_pointcloud_from_depth((d16 / 5000.) * masks[mi, 0].numpy(), rgb, 90.0, 85.0)
        assert pts.dtype == np.float64
        keep = ro.radius_outlier(pts.astype(np.float32), cfg["radius"], cfg["radius_nb_points"])
        assert np.array_equal(obj.pointcloud.points, (R @ pts[keep].T).T + pose[:3])
        assert np.array_equal(obj.pointcloud.colors, cols[keep].astype(np.float64))
    mem._ctx.close()

"""-m gpu: the batched localisation engine end to end (match -> assign -> register -> evaluate -> pose)
against the oracle transcript of object_memory.py:911-1131 and against the synthetic ground truth."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from ibloc_amd.synth import SynthWorld
from oracle import match_oracle as mo
from oracle import reg_oracle as ro
from oracle import simvolume_oracle as so

pytestmark = pytest.mark.gpu


def rot_deg(Ra, Rb):
    return float(np.degrees(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1))))


def test_localise_batch_matches_oracle_and_ground_truth():
    from ibloc_amd.engine import LocaliseEngine, MemoryShard, intensity_from_colors
    from ibloc_amd.registration import CloudBatch, RegContext
    ctx = RegContext(6 << 30)
    w = SynthWorld(9, pts_per_object=3000, E=3, D=64, seed=31)
    rng = np.random.default_rng(32)
    frames = [w.make_frame(rng, q=3, pts_per_object=3000, anchor=4), w.make_frame(rng, q=2, pts_per_object=3000, anchor=0)]
    mem = MemoryShard(ctx, list(w.embeddings), w.points, colors=w.colors)
    eng = LocaliseEngine(mem)
    clouds, ints, embs, qs = [], [], [], []
    for f in frames:
        for (p, c) in f["clouds"]:
            clouds.append(p)
            ints.append(intensity_from_colors(c))
        embs.append(f["det_emb"])
        qs.append(len(f["clouds"]))
    det = CloudBatch.from_numpy(clouds, ints)
    res = eng.localise_batch(det, qs, det_emb=np.concatenate(embs), fpfh_voxel_size=0.05, fpfh_global_dist_factor=1.5,
                             fpfh_local_dist_factor=1.5, seed=5, job_id_base=0)
    # ---- oracle transcript ---------------------------------------------------------------------
    off = np.concatenate([[0], np.cumsum([len(e) for e in w.embeddings])]).astype(np.int32)
    mem_n = mo.normalize_rows(np.concatenate(list(w.embeddings)))
    job = 0
    n_correct = 0
    for fi, f in enumerate(frames):
        sims = mo.closest_similarity(mo.normalize_rows(f["det_emb"]), mem_n, off)
        assns = so.simvolume_assignments(sims, 4)
        assert res[fi].assignments == assns                                    # bit-exact indices
        cleaned, ccols = [], []
        for (p, c) in f["clouds"]:
            k = ro.radius_outlier(p.astype(np.float32), 0.05, 8)
            cleaned.append(p[k])
            ccols.append(c[k])
        pose, recs, best = ro.localise_from_assignments(cleaned, ccols, w.points, w.colors, assns, 0.05, 1.5, 1.5, seed=5,
                                                        job_base=job, stale_means=True)
        job += len(assns)
        for k, (a, b) in enumerate(zip(res[fi].records, recs)):
            assert abs(a["full_fitness"] - b["full_fitness"]) < 5e-3, (fi, k, a["full_fitness"], b["full_fitness"])
        assert res[fi].best == best
        assert np.linalg.norm(res[fi].pose[:3] - pose[:3]) <= 0.01                # 1 cm / 0.5 deg vs the oracle
        assert rot_deg(Rotation.from_quat(res[fi].pose[3:]).as_matrix(), Rotation.from_quat(pose[3:]).as_matrix()) <= 0.5
        P = f["pose"]
        t_err = np.linalg.norm(res[fi].pose_corrected[:3] - P[:3, 3])
        r_err = np.radians(rot_deg(Rotation.from_quat(res[fi].pose_corrected[3:]).as_matrix(), P[:3, :3]))
        best_assn = res[fi].assignments[best]
        correct = all(f["ids"][d] == m for d, m in best_assn)
        print("frame", fi, "ids", f["ids"], "assns", assns, "best", best, "correct", correct, "t err", t_err, "r err", r_err,
              "full fitness", [round(r["full_fitness"], 4) for r in res[fi].records])
        # success (reference thresholds, tum_localisation_trial.py:274) at least as often as the oracle
        pose_o, _, _ = ro.localise_from_assignments(cleaned, ccols, w.points, w.colors, assns, 0.05, 1.5, 1.5, seed=5,
                                                    job_base=job - len(assns), stale_means=False)
        t_o = np.linalg.norm(pose_o[:3] - P[:3, 3])
        r_o = np.radians(rot_deg(Rotation.from_quat(pose_o[3:]).as_matrix(), P[:3, :3]))
        assert (t_err < 0.6 and r_err < 0.3) == (t_o < 0.6 and r_o < 0.3)
        n_correct += int(t_err < 0.6 and r_err < 0.3)
    assert n_correct >= 1
    ctx.close()


def test_feature_reuse_does_not_change_results():
    """LocaliseEngine.reuse_features (instance features + pair-level matching) against the reference's schedule
    (every assignment recomputes its features): identical assignments, transforms and poses"""
    import torch
    from ibloc_amd.engine import LocaliseEngine, MemoryShard, intensity_from_colors
    from ibloc_amd.registration import CloudBatch, RegContext
    from ibloc_amd.synth import SynthWorld
    w = SynthWorld(12, pts_per_object=2500, E=2, D=32, seed=61, spacing=1.6)      # close enough for recomputed groups
    rng = np.random.default_rng(62)
    frames = [w.make_frame(rng, q=3, pts_per_object=2500) for _ in range(3)]
    ctx = RegContext(6 << 30)
    mem = MemoryShard(ctx, list(w.embeddings), w.points, colors=w.colors)
    clouds = [c[0] for f in frames for c in f["clouds"]]
    ints = [intensity_from_colors(c[1]) for f in frames for c in f["clouds"]]
    det = CloudBatch.from_numpy(clouds, ints)
    emb = np.concatenate([f["det_emb"] for f in frames])
    qs = [len(f["ids"]) for f in frames]
    out = []
    for reuse in (True, False):
        eng = LocaliseEngine(mem, None)
        eng.reuse_features = reuse
        tm = {}
        out.append((eng.localise_batch(det, qs, det_emb=emb, fpfh_voxel_size=0.05, fpfh_global_dist_factor=1.5,
                                       fpfh_local_dist_factor=1.5, seed=3, timings=tm), tm["reuse"]))
    (ra, reuse_a), (rb, reuse_b) = out
    assert reuse_a[0] > 0 and reuse_b[0] == 0
    for a, b in zip(ra, rb):
        assert a.assignments == b.assignments and a.best == b.best
        assert np.array_equal(a.pose, b.pose) and np.array_equal(a.pose_corrected, b.pose_corrected)
        for x, y in zip(a.records, b.records):
            assert np.array_equal(x["T"], y["T"]) and np.array_equal(x["ransac_stats"], y["ransac_stats"])
            assert x["full_fitness"] == y["full_fitness"]
    ctx.close()


def test_fused_stage_b_entry_equals_the_staged_sequence():
    """`ibl_register_evaluate_batch` (SURVEY 8b's fused driver: outlier removal + compaction, detection features, registration, whole-memory
    evaluation, winner per frame in ONE C-ABI call) against the same stages issued one library call at a time from Python: every
    number of every FrameResult bit for bit; a frame without detections and an empty batch included; bad arguments rejected"""
    from ibloc_amd import _lib
    from ibloc_amd.engine import LocaliseEngine, MemoryShard, intensity_from_colors
    from ibloc_amd.registration import CloudBatch, RegContext, register_evaluate_batch
    from ibloc_amd.synth import SynthWorld
    w = SynthWorld(12, pts_per_object=2500, E=2, D=32, seed=61, spacing=1.6)
    rng = np.random.default_rng(63)
    frames = [w.make_frame(rng, q=q, pts_per_object=2500) for q in (3, 1, 2)]
    ctx = RegContext(6 << 30)
    mem = MemoryShard(ctx, list(w.embeddings), w.points, colors=w.colors)
    clouds = [c[0] for f in frames for c in f["clouds"]]
    rnd = np.random.default_rng(1)
    clouds[0] = np.concatenate([clouds[0], rnd.uniform(-3, 3, size=(50, 3))])           # stray points for the outlier removal
    ints = [intensity_from_colors(c[1]) for f in frames for c in f["clouds"]]
    ints[0] = np.concatenate([ints[0], np.zeros(50, np.float32)])
    det = CloudBatch.from_numpy(clouds, ints)
    emb = np.concatenate([f["det_emb"] for f in frames])
    qs = [len(f["ids"]) for f in frames]
    kw = dict(det_emb=emb, fpfh_voxel_size=0.05, fpfh_global_dist_factor=1.5, fpfh_local_dist_factor=1.5, seed=3, job_id_base=17)
    eng = LocaliseEngine(mem, None)
    fused = eng.localise_batch(det, qs, **kw)
    eng.fused_stage_b = False
    staged = eng.localise_batch(det, qs, **kw)
    assert sum(len(r.records) for r in fused) > 6
    for a, b in zip(fused, staged):
        assert a.assignments == b.assignments and a.best == b.best and a.n_clean == b.n_clean
        assert np.array_equal(a.pose, b.pose) and np.array_equal(a.pose_corrected, b.pose_corrected)
        for x, y in zip(a.records, b.records):
            for k in ("T", "T_global", "ransac_stats", "detected_mean", "memory_mean"):
                assert np.array_equal(x[k], y[k]), k
            for k in ("rmse", "fitness", "full_rmse", "full_fitness"):
                assert x[k] == y[k], k
    assert fused[0].n_clean < sum(len(c) for c in clouds[:3])                          # the stray points were removed
    # the entry itself: a frame without assignments gets best = -1, an out-of-range instance is rejected
    mf = mem.features(0.05, 1.5)
    r = register_evaluate_batch(ctx, det, qs, [fused[0].assignments, [], fused[2].assignments], mem.clouds, mf, mem.grid, 0.05, 1.5, 1.5, seed=3,
                                job_id_base=17)
    assert r["best"][1] == -1 and r["best"][0] == fused[0].best and len(r["T"]) == len(fused[0].assignments) + len(fused[2].assignments)
    with pytest.raises(_lib.IblError):
        register_evaluate_batch(ctx, det, qs, [[[[0, 999]]], [], []], mem.clouds, mf, mem.grid, 0.05, 1.5, 1.5)
    ctx.close()


def test_full_size_properties():
    """BASELINE configs[1] sizes (M = 1000 instances x 5000 points, E = 4, Q = 7): properties that need no oracle run --
    (1) batching: a frame localised alone equals the same frame inside a batch (same job ids -> same RANSAC draws);
    (2) determinism: two runs give identical bits; (3) the recovered pose is the ground truth for most frames;
    (4) the memory's resident instance features are built once and reused"""
    from ibloc_amd.engine import LocaliseEngine, MemoryShard, intensity_from_colors
    from ibloc_amd.registration import CloudBatch, RegContext
    from ibloc_amd.synth import SynthWorld
    w = SynthWorld(1000, pts_per_object=5000, E=4, D=64, seed=71)
    rng = np.random.default_rng(72)
    frames = [w.make_frame(rng, q=7, pts_per_object=5000) for _ in range(3)]
    ctx = RegContext(20 << 30)
    mem = MemoryShard(ctx, list(w.embeddings), w.points, colors=w.colors)
    eng = LocaliseEngine(mem, None)

    def run(fs, job_id_base):
        det = CloudBatch.from_numpy([c[0] for f in fs for c in f["clouds"]], [intensity_from_colors(c[1]) for f in fs for c in f["clouds"]])
        emb = np.concatenate([f["det_emb"] for f in fs])
        tm = {}
        res = eng.localise_batch(det, [len(f["ids"]) for f in fs], det_emb=emb, fpfh_voxel_size=0.05, fpfh_global_dist_factor=1.5,
                                 fpfh_local_dist_factor=1.5, seed=11, job_id_base=job_id_base, timings=tm)
        return res, tm
    batch, tm = run(frames, 0)
    again, _ = run(frames, 0)
    n_jobs = [len(r.assignments) for r in batch]
    ok = 0
    for f, r, r2 in zip(frames, batch, again):
        assert np.array_equal(r.pose, r2.pose) and r.assignments == r2.assignments                     # (2)
        ok += int(np.linalg.norm(r.pose_corrected[:3] - f["pose"][:3, 3]) < 0.3)
    assert ok >= 2                                                                                     # (3)
    base = 0
    for k, f in enumerate(frames):                                                                     # (1)
        alone, _ = run([f], base)
        assert alone[0].assignments == batch[k].assignments
        assert np.array_equal(alone[0].pose, batch[k].pose)
        for x, y in zip(alone[0].records, batch[k].records):
            assert np.array_equal(x["T"], y["T"]) and np.array_equal(x["ransac_stats"], y["ransac_stats"])
        base += n_jobs[k]
    assert tm["reuse"][0] > 0 and len(mem._features) == 1                                              # (4)
    assert ctx.status() & 1 == 0
    ctx.close()


def test_pipelined_stream_equals_batch_by_batch():
    """LocaliseEngine.localise_stream (embed + match of the next batch on a second stream while the current one registers)
    yields exactly what localise_batch returns batch by batch, with a real encoder in stage A"""
    import torch
    from ibloc_amd import vit as V
    from ibloc_amd.engine import LocaliseEngine, MemoryShard, intensity_from_colors
    from ibloc_amd.registration import CloudBatch, RegContext
    from ibloc_amd.synth import SynthWorld
    cfg = V.CONFIGS["tiny_dino"]
    enc = V.VitEncoder(cfg, V.random_weights(cfg, 4), device="cuda")
    rng = np.random.default_rng(71)
    M = 10
    base = rng.integers(0, 255, size=(M, 2, 40, 36, 3), dtype=np.uint8)          # two views per memory object
    mem_emb = enc.embed(torch.from_numpy(base.reshape(M * 2, 40, 36, 3)).cuda()).cpu().numpy().reshape(M, 2, -1)
    w = SynthWorld(M, pts_per_object=2000, E=2, D=mem_emb.shape[-1], seed=72)
    ctx = RegContext(6 << 30)
    mem = MemoryShard(ctx, [mem_emb[i] for i in range(M)], w.points, colors=w.colors)
    batches = []
    for b in range(4):
        frames = [w.make_frame(rng, q=3, pts_per_object=2000) for _ in range(2)]
        clouds = [c[0] for f in frames for c in f["clouds"]]
        ints = [intensity_from_colors(c[1]) for f in frames for c in f["clouds"]]
        ids = [i for f in frames for i in f["ids"]]
        crops = np.stack([np.clip(base[i, b % 2].astype(np.int16) + rng.integers(-6, 7, size=(40, 36, 3)), 0, 255).astype(np.uint8) for i in ids])
        batches.append(dict(det=CloudBatch.from_numpy(clouds, ints), q_per_frame=[3, 3], crops=torch.from_numpy(crops).cuda(), seed=9 + b))
    eng = LocaliseEngine(mem, enc)
    kw = dict(fpfh_voxel_size=0.05, fpfh_global_dist_factor=1.5, fpfh_local_dist_factor=1.5)
    one = [eng.localise_batch(b["det"], b["q_per_frame"], crops=b["crops"], seed=b["seed"], **kw) for b in batches]
    piped = list(eng.localise_stream(batches, **kw))
    assert len(piped) == len(one) == 4
    for ra, rb in zip(one, piped):
        for a, b in zip(ra, rb):
            assert a.assignments == b.assignments and a.best == b.best and len(a.assignments) > 0
            assert np.array_equal(a.pose, b.pose) and np.array_equal(a.pose_corrected, b.pose_corrected)
    assert list(eng.localise_stream([], **kw)) == []
    # concurrent lanes (own stream / scratch arena / encoder workspace each): same results again, in order
    for workers in (2, 3):
        lanes = list(eng.localise_concurrent(batches, workers=workers, worker_arena_bytes=2 << 30, **kw))
        assert len(lanes) == 4
        for ra, rb in zip(one, lanes):
            for a, b in zip(ra, rb):
                assert a.assignments == b.assignments and a.best == b.best
                assert np.array_equal(a.pose, b.pose) and np.array_equal(a.pose_corrected, b.pose_corrected)
    ctx.close()


def test_sharded_evaluate_equals_whole_memory():
    """ibl_evaluate_points against two instance-range shards of the memory, combined by the element-wise minimum (what the
    all-reduce(MIN) of parallel.evaluate_sharded does across ranks), gives the whole-memory fitness exactly and its rmse to rounding"""
    from ibloc_amd.parallel import fitness_rmse_from_d2
    from ibloc_amd.registration import CloudBatch, MemGrid, RegContext, evaluate_batch, evaluate_points
    rng = np.random.default_rng(91)
    ctx = RegContext(2 << 30)
    mem = [rng.uniform(-1, 1, size=(4000, 3)).astype(np.float32) + rng.uniform(-3, 3, size=3).astype(np.float32) for _ in range(9)]
    allm = np.concatenate(mem)
    det = (allm[rng.choice(len(allm), 6000)] + rng.normal(size=(6000, 3)).astype(np.float32) * 0.01).astype(np.float32)
    det4 = torch.from_numpy(np.concatenate([det, np.zeros((6000, 1), np.float32)], axis=1)).cuda()

    def grid(points):
        p4 = torch.from_numpy(np.concatenate([points, np.zeros((len(points), 1), np.float32)], axis=1)).cuda().contiguous()
        return MemGrid(ctx, p4, cell=0.04), p4

    jb, je = [0, 2500, 2500], [2500, 6000, 6000]
    T = np.stack([np.eye(4), np.eye(4), np.eye(4)])
    T[2, :3, 3] = [0.004, -0.003, 0.002]
    g_all, keep_all = grid(allm)
    g_a, keep_a = grid(np.concatenate(mem[:4]))
    g_b, keep_b = grid(np.concatenate(mem[4:]))
    rmse, fit = evaluate_batch(ctx, g_all, det4, jb, je, T, 0.02)
    d2_all, rmse_p, fit_p = evaluate_points(ctx, g_all, det4, jb, je, T, 0.02)
    assert np.array_equal(rmse, rmse_p) and np.array_equal(fit, fit_p) and d2_all.numel() == 2500 + 3500 + 3500
    da, _, _ = evaluate_points(ctx, g_a, det4, jb, je, T, 0.02)
    db, _, _ = evaluate_points(ctx, g_b, det4, jb, je, T, 0.02)
    merged = torch.minimum(da, db)
    assert torch.equal(merged, d2_all)                                   # the nearest point overall is the nearer of the two shards'
    f2, r2 = fitness_rmse_from_d2(merged, [2500, 3500, 3500])
    assert np.array_equal(f2, fit) and np.allclose(r2, rmse, rtol=1e-12, atol=0) and 0.3 < fit[0] <= 1.0
    ctx.close()

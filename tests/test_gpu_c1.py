"""-m gpu: BASELINE configs[0] ("C1") END TO END on the HIP path -- DINOv2 ViT-S/14 (D = 384) embeddings of Q = 4 crops per
frame of sizes H, W ~ U{64..400} through `ibl_preprocess_crops` (PIL-exact resize + centre crop), a 20-instance memory with E = 4 stored
views, 5 000-point clouds, FPFH + RANSAC + coloured ICP, whole-memory evaluation, pose -- against the oracle transcript of
ObjectMemory.localise (object_memory/object_memory.py:911-1131): the fp32 torch ViT, the numpy similarity volume and the C registration
restatement.  Same generator as `bench.py --config C1`."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

import bench
from oracle import match_oracle as mo
from oracle import reg_oracle as ro
from oracle import simvolume_oracle as so
from oracle import vit_oracle as vo

pytestmark = pytest.mark.gpu
M, E, Q, N_FRAMES = 20, 4, 4, 6


def rot_deg(Ra, Rb):
    return float(np.degrees(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1))))


def test_c1_end_to_end_against_the_oracle_transcript():
    from ibloc_amd import preprocess as pp
    from ibloc_amd import vit as V
    from ibloc_amd.engine import LocaliseEngine, MemoryShard, intensity_from_colors
    from ibloc_amd.registration import CloudBatch, RegContext
    from ibloc_amd.synth import SynthWorld
    cfg = V.CONFIGS["dinov2_vits14"]
    w = V.random_weights(cfg, 20)
    enc = V.VitEncoder(cfg, w)
    world = SynthWorld(M, pts_per_object=5000, E=E, D=cfg.out_dim, seed=11)
    gen = bench.VarCrops(21)
    rng = np.random.default_rng(10)
    mem_crops = [gen.make(k, rng) for k in range(M) for _ in range(E)]
    assert len({c.shape[:2] for c in mem_crops}) > 40 and min(min(c.shape[:2]) for c in mem_crops) >= 64
    mem_hip = enc.embed(V.PackedCrops(mem_crops)).cpu().numpy()
    frames = [world.make_frame(rng, q=Q, pts_per_object=5000) for _ in range(N_FRAMES)]
    det_crops = [gen.make(k, rng) for f in frames for k in f["ids"]]
    qs = [len(f["ids"]) for f in frames]
    ctx = RegContext(6 << 30)
    eng = LocaliseEngine(MemoryShard(ctx, list(mem_hip.reshape(M, E, -1)), world.points, colors=world.colors), enc)
    det = CloudBatch.from_numpy([c[0] for f in frames for c in f["clouds"]], [intensity_from_colors(c[1]) for f in frames for c in f["clouds"]])
    res = eng.localise_batch(det, qs, crops=V.PackedCrops(det_crops), fpfh_voxel_size=0.05, fpfh_global_dist_factor=1.5,
                             fpfh_local_dist_factor=1.5, seed=10)
    det_hip = enc.embed(det_crops).cpu().numpy()                # the list form uploads per call: same bytes, same embeddings
    assert np.array_equal(det_hip, enc.embed(V.PackedCrops(det_crops)).cpu().numpy())

    # (1) embedding gate: every crop within 1e-3 rel-L2 of the fp32 forward on the PIL-preprocessed crop
    recipe = pp.RECIPES[cfg.recipe]
    mem_o = vo.embed_crops(w, cfg, recipe, mem_crops, device="cuda")
    det_o = vo.embed_crops(w, cfg, recipe, det_crops, device="cuda")
    rel = np.concatenate([np.linalg.norm(mem_hip - mem_o, axis=1) / np.linalg.norm(mem_o, axis=1),
                          np.linalg.norm(det_hip - det_o, axis=1) / np.linalg.norm(det_o, axis=1)])
    print("C1 embedding rel-L2: mean", rel.mean(), "max", rel.max())
    assert rel.max() < 1e-3

    # (2) identical embeddings -> bit-exact assignment lists (the north star's contract), then the registration transcript
    off = (np.arange(M + 1) * E).astype(np.int32)
    memn = mo.normalize_rows(mem_hip)
    memn_o = mo.normalize_rows(mem_o)
    r0 = job = same_lists = n_ok = n_ok_oracle = 0
    for fi, f in enumerate(frames):
        sims = mo.closest_similarity(mo.normalize_rows(det_hip[r0:r0 + Q]), memn, off)
        assns = so.simvolume_assignments(sims, 4)
        assert res[fi].assignments == assns, fi
        # the oracle's own fp32 embeddings: the best single match is the same instance; the whole list on all but at most one frame
        assns_o = so.simvolume_assignments(mo.closest_similarity(mo.normalize_rows(det_o[r0:r0 + Q]), memn_o, off), 4)
        assert [a for a in assns if len(a) == 1][0] == [a for a in assns_o if len(a) == 1][0]
        same_lists += int(assns == assns_o)
        cleaned, ccols = [], []
        for (p, c) in f["clouds"]:
            k = ro.radius_outlier(p.astype(np.float32), 0.05, 8)
            cleaned.append(p[k])
            ccols.append(c[k])
        assert res[fi].n_clean == sum(len(c) for c in cleaned)
        pose, recs, best = ro.localise_from_assignments(cleaned, ccols, world.points, world.colors, assns, 0.05, 1.5, 1.5, seed=10,
                                                        job_base=job, stale_means=True)
        pose_c, _, _ = ro.localise_from_assignments(cleaned, ccols, world.points, world.colors, assns, 0.05, 1.5, 1.5, seed=10,
                                                    job_base=job, stale_means=False)
        job += len(assns)
        assert res[fi].best == best
        for a, b in zip(res[fi].records, recs):
            assert abs(a["full_fitness"] - b["full_fitness"]) < 5e-3
        assert np.linalg.norm(res[fi].pose[:3] - pose[:3]) <= 0.01
        assert rot_deg(Rotation.from_quat(res[fi].pose[3:]).as_matrix(), Rotation.from_quat(pose[3:]).as_matrix()) <= 0.5
        P = f["pose"]
        for got, counter in ((res[fi].pose_corrected, "hip"), (pose_c, "oracle")):
            ok = np.linalg.norm(got[:3] - P[:3, 3]) < 0.6 and np.radians(rot_deg(Rotation.from_quat(got[3:]).as_matrix(), P[:3, :3])) < 0.3
            n_ok += int(ok and counter == "hip")
            n_ok_oracle += int(ok and counter == "oracle")
        r0 += Q
    print("C1: frames whose assignment list equals the fp32 oracle's:", same_lists, "of", N_FRAMES, "; localised (0.6 m / 0.3 rad):", n_ok,
          "oracle:", n_ok_oracle)
    assert same_lists >= N_FRAMES - 1
    assert n_ok >= n_ok_oracle and n_ok >= N_FRAMES - 1          # success at least as often as the CPU restatement
    ctx.close()

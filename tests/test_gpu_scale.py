"""-m gpu: the configurations beyond the bench default, one slice each on a single GPU (BASELINE.json configs[2..4]):
  C3  DATOR RGB-D encoder in the engine (crops = (rgb, depth) pair) -- assignments from the HIP embeddings equal those from the
      fp32 oracle embeddings of the same crops;
  C4  match + candidate selection + assignment against a 50 000-instance embedding memory (replicated on one GPU);
  C5  registration / evaluation of 100 000-point objects (a length-2 job is 2 x 200 000 points) against the oracle."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from ibloc_amd.synth import SynthWorld
from oracle import match_oracle as mo
from oracle import reg_oracle as ro
from oracle import simvolume_oracle as so

pytestmark = pytest.mark.gpu


def test_c3_dator_encoder_in_the_engine():
    from ibloc_amd import dator as D
    from ibloc_amd.engine import LocaliseEngine, MemoryShard
    from ibloc_amd.registration import RegContext
    from oracle import dator_oracle as do
    wr, wd, wh = D.random_stream_weights(3), D.random_stream_weights(4), D.random_head_weights(5)
    enc = D.DatorEncoder(wr, wd, wh)
    rng = np.random.default_rng(6)
    M, E = 24, 2
    yy, xx = np.meshgrid(np.linspace(0, 1, 256), np.linspace(0, 1, 128), indexing="ij")

    def crop(k, r):
        img = np.stack([0.5 + 0.5 * np.sin(2 * np.pi * ((1 + k % 5) * xx * (c + 1) + (1 + k // 5) * yy) + k) for c in range(3)], -1)
        img = np.clip(img + r.normal(0, 0.02, img.shape), 0, 1)
        dep = (1.0 + 0.2 * k + 0.5 * np.sin(3 * yy[::4, ::4] + k)).astype(np.float32) + r.normal(0, 0.01, (64, 32)).astype(np.float32)
        return (img * 255).astype(np.uint8), dep

    mem_rgb, mem_dep = zip(*[crop(k, rng) for k in range(M) for _ in range(E)])
    mem_emb = enc.embed(torch.from_numpy(np.stack(mem_rgb)).cuda(), torch.from_numpy(np.stack(mem_dep)).cuda()).cpu().numpy().reshape(M, E, 128)
    ctx = RegContext(64 << 20)
    eng = LocaliseEngine(MemoryShard(ctx, list(mem_emb)), enc)
    q = [3, 1, 7, 2]
    ids = rng.integers(0, M, size=sum(q))
    q_rgb, q_dep = zip(*[crop(int(k), rng) for k in ids])
    crops = (torch.from_numpy(np.stack(q_rgb)).cuda(), torch.from_numpy(np.stack(q_dep)).cuda())
    res = eng.localise_batch(None, q, crops=crops, register=False)
    # the oracle's embeddings (torch fp32 restatement of build_FourDNet) of the same crops, matched and assigned by the oracle
    fr, fd = D.fold_lora(wr), D.fold_lora(wd)
    mem_o = do.embed(fr, fd, wh, list(mem_rgb), list(mem_dep)).reshape(M, E, 128)
    det_o = do.embed(fr, fd, wh, list(q_rgb), list(q_dep))
    rel = np.linalg.norm(mem_emb - mem_o) / np.linalg.norm(mem_o)
    print("C3 memory embeddings rel-L2 vs the fp32 oracle:", rel)
    assert rel < 1e-3, rel
    off = (np.arange(M + 1) * E).astype(np.int32)
    memn = mo.normalize_rows(mem_o.reshape(M * E, 128))
    r0 = 0
    same = 0
    for f, n in enumerate(q):
        sims = mo.closest_similarity(mo.normalize_rows(det_o[r0:r0 + n]), memn, off)
        same += int(res[f].assignments == so.simvolume_assignments(sims, 4))
        top1 = [a for a in res[f].assignments if len(a) == 1][0][0]
        assert top1[1] == ids[r0 + top1[0]]                                       # the best single match is the true instance
        r0 += n
    print("frames whose assignment list equals the fp32 oracle's:", same, "of", len(q))
    assert same >= len(q) - 1
    ctx.close()


def test_c4_match_and_assign_against_50k_instances():
    from ibloc_amd.engine import LocaliseEngine, MemoryShard
    from ibloc_amd.registration import RegContext
    rng = np.random.default_rng(40)
    M, E, D, F = 50000, 4, 768, 8
    base = rng.normal(size=(M, D)).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    emb = base[:, None, :] + rng.normal(0, 0.35 / np.sqrt(D), size=(M, E, D)).astype(np.float32)
    ctx = RegContext(64 << 20)
    eng = LocaliseEngine(MemoryShard(ctx, list(emb)))
    q = np.full(F, 7)
    ids = rng.integers(0, M, size=7 * F)
    det = base[ids] + rng.normal(0, 0.1 / np.sqrt(D), size=(7 * F, D)).astype(np.float32)
    tm = {}
    a = eng.localise_batch(None, q, det_emb=det, register=False, timings=tm)
    assert eng.stats["fallback_frames"] == 0
    eng.use_candidates = False
    b = eng.localise_batch(None, q, det_emb=det, register=False)
    assert [r.assignments for r in a] == [r.assignments for r in b]
    for f in range(F):
        singles = [x[0] for x in a[f].assignments if len(x) == 1]
        assert singles and singles[0][1] == ids[7 * f + singles[0][0]]
    print("M = 50 000 stage times (ms, 8 frames):", {k: round(v, 3) for k, v in tm.items() if isinstance(v, float)})
    ctx.close()


def _deg(Ra, Rb):
    return float(np.degrees(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1))))


def test_c5_registration_of_100k_point_objects_matches_oracle():
    from ibloc_amd.engine import LocaliseEngine, MemoryShard, intensity_from_colors
    from ibloc_amd.registration import CloudBatch, RegContext
    N = 100000
    w = SynthWorld(3, pts_per_object=N, E=1, D=16, seed=71, spacing=6.0, extent=(0.6, 2.4))
    rng = np.random.default_rng(72)
    f = w.make_frame(rng, q=2, pts_per_object=N, anchor=0)
    ctx = RegContext(20 << 30)
    eng = LocaliseEngine(MemoryShard(ctx, list(w.embeddings), w.points, colors=w.colors))
    det = CloudBatch.from_numpy([c[0] for c in f["clouds"]], [intensity_from_colors(c[1]) for c in f["clouds"]])
    res = eng.localise_batch(det, [2], det_emb=f["det_emb"], fpfh_voxel_size=0.05, fpfh_global_dist_factor=1.5, fpfh_local_dist_factor=1.5,
                             seed=13)[0]
    assert eng.check_status() & 1 == 0                       # no scratch grid collapsed under 400 000 points per job
    sims = mo.closest_similarity(mo.normalize_rows(f["det_emb"]), mo.normalize_rows(np.concatenate(list(w.embeddings))),
                                 np.arange(4, dtype=np.int32))
    assns = so.simvolume_assignments(sims, 4)
    assert res.assignments == assns
    cleaned, ccols = [], []
    for p, c in f["clouds"]:
        k = ro.radius_outlier(p.astype(np.float32), 0.05, 8)
        cleaned.append(p[k])
        ccols.append(c[k])
    assert res.n_clean == sum(len(c) for c in cleaned) and res.n_clean > 1.5 * N
    # the oracle on the single-object assignment (100 000 source x 100 000 target points; its feature matching alone is 2 x 10^10
    # 33-d distances on the host), the two-object job (2 x 200 000 points) against the device's own uncached schedule and the truth
    one = [a for a in assns if len(a) == 1][:1]
    k1 = assns.index(one[0])
    pose_o, recs, _ = ro.localise_from_assignments(cleaned, ccols, w.points, w.colors, one, 0.05, 1.5, 1.5, seed=13, job_base=k1,
                                                   stale_means=False)
    a, b = res.records[k1], recs[0]
    dt = np.linalg.norm(a["T_global"][:3, 3] - b["T_global"][:3, 3])
    dr = _deg(a["T_global"][:3, :3], b["T_global"][:3, :3])
    print(f"100k-point objects: vs oracle {dt:.2e} m / {dr:.2e} deg; fitness {a['fitness']:.4f} / {b['fitness']:.4f}; "
          f"full fitness {a['full_fitness']:.4f} / {b['full_fitness']:.4f}")
    assert dt <= 0.01 and dr <= 0.5 and abs(a["fitness"] - b["fitness"]) < 5e-3 and abs(a["full_fitness"] - b["full_fitness"]) < 5e-3
    eng.reuse_features = False               # the reference's schedule: features of every job recomputed on its concatenated clouds
    res2 = eng.localise_batch(det, [2], det_emb=f["det_emb"], fpfh_voxel_size=0.05, fpfh_global_dist_factor=1.5, fpfh_local_dist_factor=1.5,
                              seed=13)[0]
    P = f["pose"]
    for k, (x, y) in enumerate(zip(res.records, res2.records)):
        assert np.array_equal(x["T"], y["T"]) and np.array_equal(x["ransac_stats"], y["ransac_stats"]), k      # cached == uncached, bit for bit
    two = [k for k, a_ in enumerate(assns) if len(a_) == 2][0]
    G = res.records[two]["T_global"]
    assert np.linalg.norm(G[:3, 3] - P[:3, 3]) < 0.6 and np.radians(_deg(G[:3, :3], P[:3, :3])) < 0.3
    assert eng.check_status() & 1 == 0
    ctx.close()

"""-m gpu: the HIP path on data the reference itself ships (tests/golden/ref_scene: the 11 k / 54 k / 9.8 k-point objects of a memory
it saved, and RGB-D views of the room).  Real densities, a cloud ten times the synthetic ones, real depth images: radius-outlier
masks and unprojected clouds bit for bit against the oracle, normals / FPFH within the usual tolerances, and a whole localise()
of two real views against the saved memory -- same assignments, same winner, pose within 1 cm / 0.5 deg of the oracle and within
the reference's success rule (0.6 m / 0.3 rad) of the ground-truth camera pose of poses.json."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from oracle import depth_oracle as do
from oracle import match_oracle as mo
from oracle import reg_oracle as ro
from oracle import simvolume_oracle as so
from tests import ref_scene as rs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ibloc_amd.registration import RegContext
    c = RegContext(6 << 30)
    yield c
    c.close()


def test_radius_outlier_masks_on_the_saved_objects(ctx):
    from ibloc_amd.registration import CloudBatch, radius_outlier_batch
    objs = rs.memory_objects()
    rng = np.random.default_rng(1)
    cs = [np.concatenate([p, p[rng.integers(0, len(p), 60)] + rng.uniform(0.2, 0.6, size=(60, 3))]).astype(np.float32) for p, _ in objs]
    b = CloudBatch.from_numpy(cs)
    for r, nb in [(0.05, 8), (0.05, 12), (0.02, 20)]:
        keep = radius_outlier_batch(ctx, b, r, nb).cpu().numpy().astype(bool)
        for i, c in enumerate(cs):
            assert np.array_equal(keep[b.seg_off_host[i]:b.seg_off_host[i + 1]], ro.radius_outlier(c, r, nb)), (i, r, nb)


def test_unprojection_of_the_real_views_bit_exact(ctx):
    from ibloc_amd.registration import unproject_masks
    objs = rs.memory_objects()
    for k, (depth, rgb, pose) in rs.views().items():
        masks = rs.object_masks(depth, pose, objs) + [np.ones(depth.shape, bool)]
        got = unproject_masks(ctx, torch.from_numpy(depth).cuda(), torch.from_numpy(rgb).cuda(), torch.from_numpy(np.stack(masks)).cuda(),
                              rs.FX, rs.FY)
        exp = do.mask_clouds(depth, rgb, masks, rs.FX, rs.FY)
        p4 = got.pts4.cpu().numpy()
        for i, (ep, ei) in enumerate(exp):
            seg = p4[got.seg_off_host[i]:got.seg_off_host[i + 1]]
            assert seg.shape[0] == len(ep) and np.array_equal(seg[:, :3], ep) and np.array_equal(seg[:, 3], ei), (k, i)


@pytest.fixture(scope="module")
def saved_objects_oracle():
    """centred fp32 clouds of the three saved objects, their intensities, and the oracle's normals / FPFH / colour gradients"""
    objs = rs.memory_objects()
    cs = [(p - p.mean(0)).astype(np.float32) for p, _ in objs]
    ints = [ro.intensity(c).astype(np.float32) for _, c in objs]
    en = [ro.normals(c, 0.1, 30) for c in cs]
    ef = [ro.fpfh(c, n, 0.25, 100) for c, n in zip(cs, en)]
    eg = [ro.color_gradient(c, n, i, 0.15, 30) for c, n, i in zip(cs, en, ints)]
    return cs, ints, en, ef, eg


def test_normals_and_fpfh_on_the_54k_point_object(ctx, saved_objects_oracle):
    from ibloc_amd.registration import CloudBatch, normals_fpfh_batch
    cs, _, en, ef, _ = saved_objects_oracle
    b = CloudBatch.from_numpy(cs)
    nrm, fpfh = normals_fpfh_batch(ctx, b, 0.1, 30, 0.25, 100)
    torch.cuda.synchronize()
    nrm, fpfh = nrm.cpu().numpy(), fpfh.cpu().numpy()
    for i, c in enumerate(cs):
        lo, hi = b.seg_off_host[i], b.seg_off_host[i + 1]
        assert np.mean(np.abs(nrm[lo:hi, :3] - en[i]).max(1) < 1e-5) > 0.999, i
        assert np.mean(np.abs(fpfh[lo:hi] - ef[i]).max(1) < 2e-3) > 0.995, i


def test_instance_features_product_entry_vs_oracle_on_the_saved_objects(ctx, saved_objects_oracle):
    """`ibl_instance_features_batch` -- the entry the product path calls (one fused search for normals + features, rows in matching
    order, fp16 operand rows, colour gradients) -- directly against the oracle on the reference's own 11 k / 54 k / 9.8 k-point objects"""
    from ibloc_amd.registration import FEAT_ORDER, CloudBatch, instance_features_batch
    cs, ints, en, ef, eg = saved_objects_oracle
    b = CloudBatch.from_numpy(cs, ints)
    feat = instance_features_batch(ctx, b, 0.05, grad_radius=0.15)           # voxel 0.05: normals 0.1 / 30, FPFH 0.25 / 100 (fpfh_register.py:88-97)
    torch.cuda.synchronize()
    assert ctx.status() == 0
    nrm = feat.normals[:b.n].cpu().numpy()
    f = feat.fpfh[:b.n].cpu().numpy()
    g = feat.grad[:b.n].cpu().numpy()
    for i, c in enumerate(cs):
        lo, hi = b.seg_off_host[i], b.seg_off_host[i + 1]
        assert np.mean(np.abs(nrm[lo:hi, :3] - en[i]).max(1) < 1e-5) > 0.999, i
        row = np.abs(f[lo:hi] - ef[i][:, FEAT_ORDER]).max(1)                   # resident rows are stored in matching order
        assert np.mean(row < 2e-3) > 0.995, (i, float(np.mean(row < 2e-3)))
        gerr = np.abs(g[lo:hi, :3] - eg[i]).max(1)
        tol = 1e-4 * max(1.0, float(np.abs(eg[i]).max()))
        assert np.mean(gerr < tol) > 0.995, (i, float(np.mean(gerr < tol)))
        assert np.array_equal(feat.bbox[i, :3], c.min(0)) and np.array_equal(feat.bbox[i, 3:], c.max(0))


def _err(pose7, T):
    R = Rotation.from_quat(pose7[3:]).as_matrix()
    return float(np.linalg.norm(pose7[:3] - T[:3, 3])), float(np.arccos(np.clip((np.trace(R.T @ T[:3, :3]) - 1) / 2, -1, 1)))


def test_localise_real_views_against_the_saved_memory(ctx):
    """the whole path (match -> assign -> outlier removal -> register -> evaluate -> pose) on real clouds: the three saved objects are
    the memory (synthetic identity embeddings, the reference's own info.pkl files hold dummies), the detections are the objects'
    pixels of views 1 and 8"""
    from ibloc_amd.engine import LocaliseEngine, MemoryShard
    from ibloc_amd.registration import CloudBatch
    objs = rs.memory_objects()
    vs = rs.views()
    rng = np.random.default_rng(4)
    D = 32
    base = rng.normal(size=(3, D))
    emb = [(base[j][None] + rng.normal(0, 0.05, size=(2, D))).astype(np.float32) for j in range(3)]
    eng = LocaliseEngine(MemoryShard(ctx, emb, [o[0] for o in objs], colors=[o[1] for o in objs]))
    frames = [(1, [0, 2]), (8, [0, 1])]
    clouds, ints, qs, det_emb = [], [], [], []
    for k, seen in frames:
        depth, rgb, pose = vs[k]
        masks = rs.object_masks(depth, pose, objs)
        for pts, inten in do.mask_clouds(depth, rgb, [masks[j] for j in seen], rs.FX, rs.FY):
            clouds.append(pts)
            ints.append(inten)
        qs.append(len(seen))
        det_emb.append((base[seen] + rng.normal(0, 0.05, size=(len(seen), D))).astype(np.float32))
    det = CloudBatch.from_numpy(clouds, ints)
    res = eng.localise_batch(det, qs, det_emb=np.concatenate(det_emb), fpfh_voxel_size=0.05, fpfh_global_dist_factor=1.5,
                             fpfh_local_dist_factor=1.5, seed=11)
    off = (np.arange(4) * 2).astype(np.int32)
    mem_n = mo.normalize_rows(np.concatenate(emb))
    c0 = job = 0
    for fi, (k, seen) in enumerate(frames):
        sims = mo.closest_similarity(mo.normalize_rows(det_emb[fi]), mem_n, off)
        assns = so.simvolume_assignments(sims, 4)
        assert res[fi].assignments == assns
        cleaned, ccols = [], []
        for d in range(qs[fi]):
            keep = ro.radius_outlier(clouds[c0 + d], 0.05, 8)
            cleaned.append(clouds[c0 + d][keep])
            ccols.append(np.repeat(ints[c0 + d][keep][:, None], 3, axis=1))
        c0 += qs[fi]
        assert res[fi].n_clean == sum(len(c) for c in cleaned)
        pose_o, recs, best = ro.localise_from_assignments(cleaned, ccols, [o[0] for o in objs], [o[1] for o in objs], assns, 0.05, 1.5, 1.5,
                                                          seed=11, job_base=job, stale_means=False)
        job += len(assns)
        assert res[fi].best == best
        for a, b in zip(res[fi].records, recs):
            assert abs(a["full_fitness"] - b["full_fitness"]) < 5e-3 and abs(a["fitness"] - b["fitness"]) < 5e-3
        To = rs.pose_matrix(pose_o)
        te, re_ = _err(res[fi].pose_corrected, To)
        assert te <= 0.01 and np.degrees(re_) <= 0.5, (k, te, re_)                 # 1 cm / 0.5 deg vs the oracle
        tg, rg = _err(res[fi].pose_corrected, rs.pose_matrix(vs[k][2]))
        print(f"view {k}: vs oracle {te:.2e} m / {np.degrees(re_):.2e} deg; vs ground truth {tg:.3f} m / {rg:.3f} rad; "
              f"full fitness {[round(r['full_fitness'], 3) for r in res[fi].records]}")
        assert tg < 0.6 and rg < 0.3                                               # the reference's success rule


def test_localise_all_eight_views_against_the_saved_memory(ctx):
    """every view of the reference's room (data/our-synthetic/360_basic_test, poses.json) in ONE batched call against the memory the
    reference saved, compared view by view with the oracle's stored transcript (tests/golden/ref_scene/oracle_views.json,
    tools/gen_golden_ref_views.py): same detections after outlier removal, same assignment lists, same winner, candidate fitnesses,
    pose within 1 cm / 0.5 deg of the oracle's, and the reference's success rule against the ground-truth pose wherever the oracle
    meets it (six views; views 2 and 7 fail on both sides, tests/test_ref_scene.py)."""
    from ibloc_amd.engine import LocaliseEngine, MemoryShard
    from ibloc_amd.registration import CloudBatch
    from tests.test_ref_scene import ORACLE_FAILS
    objs = rs.memory_objects()
    _, emb = rs.memory_embeddings()
    frames = rs.view_frames()
    oracle = rs.oracle_views()
    eng = LocaliseEngine(MemoryShard(ctx, emb, [o[0] for o in objs], colors=[o[1] for o in objs]))
    order = sorted(frames)
    clouds = [c for k in order for c in frames[k]["clouds"]]
    ints = [c for k in order for c in frames[k]["ints"]]
    qs = [len(frames[k]["seen"]) for k in order]
    det = CloudBatch.from_numpy(clouds, ints)
    res = eng.localise_batch(det, qs, det_emb=np.concatenate([frames[k]["det_emb"] for k in order]), fpfh_voxel_size=0.05,
                             fpfh_global_dist_factor=1.5, fpfh_local_dist_factor=1.5, seed=oracle["seed"])
    n_ok = 0
    for k, r in zip(order, res):
        v = oracle["views"][str(k)]
        assert frames[k]["seen"] == v["seen"] and r.n_clean == sum(v["n_clean"]), k
        assert r.assignments == v["assignments"], k
        te, re_ = rs.pose_error(r.pose_corrected, rs.pose_matrix(np.asarray(v["pose"])))
        tg, rg = rs.pose_error(r.pose_corrected, rs.pose_matrix(frames[k]["pose"]))
        ff = [rec["full_fitness"] for rec in r.records]
        print(f"view {k}: seen {v['seen']} best {r.best} (oracle {v['best']}); vs oracle {te:.2e} m / {np.degrees(re_):.2e} deg; "
              f"vs ground truth {tg:.3f} m / {rg:.3f} rad; full fitness {[round(x, 3) for x in ff]} (oracle {[round(x, 3) for x in v['full_fitness']]})")
        ok = tg < 0.6 and rg < 0.3
        n_ok += int(ok)
        if k in ORACLE_FAILS:
            assert not ok, k                         # the same verdict as the oracle: not localised
            continue
        assert ok, (k, tg, rg)                       # the reference's success rule against poses.json
        assert r.best == v["best"], k
        assert np.allclose(ff, v["full_fitness"], atol=5e-3), k
        assert np.allclose([rec["fitness"] for rec in r.records], v["fitness"], atol=5e-3), k
        assert te <= 0.01 and np.degrees(re_) <= 0.5, (k, te, re_)
    assert n_ok == 8 - len(ORACLE_FAILS)

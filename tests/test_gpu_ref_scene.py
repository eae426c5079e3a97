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


def test_normals_and_fpfh_on_the_54k_point_object(ctx):
    from ibloc_amd.registration import CloudBatch, normals_fpfh_batch
    objs = rs.memory_objects()
    cs = [(p - p.mean(0)).astype(np.float32) for p, _ in objs]
    b = CloudBatch.from_numpy(cs)
    nrm, fpfh = normals_fpfh_batch(ctx, b, 0.1, 30, 0.25, 100)
    torch.cuda.synchronize()
    nrm, fpfh = nrm.cpu().numpy(), fpfh.cpu().numpy()
    for i, c in enumerate(cs):
        lo, hi = b.seg_off_host[i], b.seg_off_host[i + 1]
        en = ro.normals(c, 0.1, 30)
        assert np.mean(np.abs(nrm[lo:hi, :3] - en).max(1) < 1e-5) > 0.999, i
        ef = ro.fpfh(c, en, 0.25, 100)
        assert np.mean(np.abs(fpfh[lo:hi] - ef).max(1) < 2e-3) > 0.995, i


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

"""-m gpu: batched registration (features -> matching -> RANSAC -> coloured ICP) and whole-memory evaluation
vs the C oracle.

Tolerances (parity at the Open3D boundary is unpinned; the checker is the oracle with identical Philox draws):
  * final centred transform: |dt| <= 1 cm, rotation <= 0.5 deg for jobs whose RANSAC stage
    agreed; the RANSAC stage itself must agree bit-for-bit in its statistics for >= 80 % of the jobs (a single
    differing feature correspondence changes the draw -> hypothesis mapping) and in every case land within
    the ICP basin (final poses agree);
  * evaluate: fitness within 2 points, rmse within 1e-5."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from ibloc_amd.synth import SynthWorld
from oracle import reg_oracle as ro

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ibloc_amd.registration import RegContext
    c = RegContext(6 << 30)
    yield c
    c.close()


def rot_deg(A, B):
    return float(np.degrees(np.arccos(np.clip((np.trace(A[:3, :3].T @ B[:3, :3]) - 1) / 2, -1, 1))))


@pytest.fixture(scope="module")
def scene():
    w = SynthWorld(9, pts_per_object=3000, E=1, D=8, seed=21)
    rng = np.random.default_rng(22)
    f = w.make_frame(rng, q=3, pts_per_object=3000, anchor=4)
    return w, f


def test_register_batch_vs_oracle(ctx, scene):
    from ibloc_amd.registration import CloudBatch, register_batch
    w, f = scene
    ids = f["ids"]
    det = CloudBatch.from_numpy([c[0] for c in f["clouds"]], [ro.intensity(c[1]) for c in f["clouds"]])
    mem = CloudBatch.from_numpy(w.points, [ro.intensity(c) for c in w.colors])
    assns = [[[0, ids[0]]], [[1, ids[1]]], [[0, ids[0]], [1, ids[1]]], [[0, ids[0]], [1, ids[1]], [2, ids[2]]],
             [[0, ids[1]]]]                                # the last one is a wrong assignment
    js = [[d for d, m in a] for a in assns]
    jt = [[m for d, m in a] for a in assns]
    pad = lambda L: [x + [-1] * (3 - len(x)) for x in L]
    out = register_batch(ctx, det, mem, pad(js), pad(jt), 0.05, 1.5, 1.5, seed=77, job_id_base=100)
    assert ctx.status() & 1 == 0
    n_same_ransac = 0
    for j, a in enumerate(assns):
        cd = np.concatenate([f["clouds"][d][0].astype(np.float32) for d, m in a]).astype(np.float64)
        cm = np.concatenate([w.points[m].astype(np.float32) for d, m in a]).astype(np.float64)
        dm, mm = cd.mean(0), cm.mean(0)
        assert np.allclose(out["means"][j, 0], dm, atol=1e-9) and np.allclose(out["means"][j, 1], mm, atol=1e-9)
        src, tgt = (cd - dm).astype(np.float32), (cm - mm).astype(np.float32)
        si = np.concatenate([ro.intensity(f["clouds"][d][1]) for d, m in a])
        ti = np.concatenate([ro.intensity(w.colors[m]) for d, m in a])
        T, rmse, fit, Tr, stats = ro.register_point_clouds(src, si, tgt, ti, 0.05, 1.5, 1.5, seed=77, job_id=100 + j,
                                                           src_raw=cd.astype(np.float32), tgt_raw=cm.astype(np.float32))
        same = np.array_equal(stats, out["ransac_stats"][j])
        n_same_ransac += same
        print(f"job {j}: oracle stats {stats} gpu {out['ransac_stats'][j]} fit {fit:.4f}/{out['fitness'][j]:.4f} "
              f"dt {np.linalg.norm(T[:3, 3] - out['T'][j][:3, 3]):.2e} drot {rot_deg(T, out['T'][j]):.3f}")
        if same:
            assert np.allclose(Tr, out["T_ransac"][j], atol=1e-6)
        if j < 4:       # correct assignments: both converge to the same pose
            assert np.linalg.norm(T[:3, 3] - out["T"][j][:3, 3]) <= 0.01
            assert rot_deg(T, out["T"][j]) <= 0.5
            assert abs(fit - out["fitness"][j]) < 5e-3 and abs(rmse - out["rmse"][j]) < 1e-3
            # and that pose is the ground truth
            G = out["T"][j].copy()
            G[:3, 3] = G[:3, 3] + mm - G[:3, :3] @ dm
            assert np.linalg.norm(G[:3, 3] - f["pose"][:3, 3]) < 0.05 and rot_deg(G, f["pose"]) < 1.5
    assert n_same_ransac == len(assns)          # (hypothesis i of job j is a pure function of (seed, job id, i): the walks coincide)


def test_point_to_point_fallback(ctx, scene):
    from ibloc_amd.registration import CloudBatch, register_batch
    w, f = scene
    src = w.points[0][:2000].astype(np.float32) - w.points[0][:2000].mean(0).astype(np.float32)
    R = Rotation.from_euler("xyz", [2, -3, 1.5], degrees=True).as_matrix()
    tgt = (src.astype(np.float64) @ R.T + [0.01, -0.02, 0.015]).astype(np.float32)
    det, mem = CloudBatch.from_numpy([src]), CloudBatch.from_numpy([tgt])
    out = register_batch(ctx, det, mem, [[0, -1, -1]], [[0, -1, -1]], 0.05, 1.5, 1.5, have_colors=False, center=False)
    T, fit, rmse, it = ro.icp(src, None, tgt, None, None, None, 0.075, np.eye(4), colored=False)
    assert np.allclose(out["T"][0], T, atol=1e-6)
    assert abs(out["fitness"][0] - fit) < 1e-9 and abs(out["rmse"][0] - rmse) < 1e-7
    assert rot_deg(out["T"][0], np.vstack([np.c_[R, [0.01, -0.02, 0.015]], [0, 0, 0, 1]])) < 0.05


def test_evaluate_batch_vs_oracle(ctx, scene):
    from ibloc_amd.registration import CloudBatch, MemGrid, evaluate_batch
    w, f = scene
    mem = CloudBatch.from_numpy(w.points, [ro.intensity(c) for c in w.colors])
    det = CloudBatch.from_numpy([c[0] for c in f["clouds"]])
    grid = MemGrid(ctx, mem.pts4, 0.04)
    P = f["pose"]
    Ts = [P, np.eye(4), P.copy()]
    Ts[2] = P.copy()
    Ts[2][:3, 3] += [0.01, 0, 0.005]
    n = det.n
    rmse, fit = evaluate_batch(ctx, grid, det.pts4, [0, 0, det.seg_off_host[1]], [n, n, n], np.stack(Ts), 0.02)
    all_det = np.concatenate([c[0] for c in f["clouds"]]).astype(np.float32)
    all_mem = np.concatenate(w.points).astype(np.float32)
    for j, (b, T) in enumerate(zip([0, 0, det.seg_off_host[1]], Ts)):
        er, ef = ro.evaluate(all_det[b:], all_mem, T, 0.02)
        assert abs(ef - fit[j]) <= 2.0 / (n - b) and abs(er - rmse[j]) < 1e-5, (j, ef, fit[j], er, rmse[j])
    assert fit[0] > fit[2] > fit[1]
    grid.close()


def test_evaluate_pruned_scan_equals_the_full_scan_far_from_the_origin(ctx, monkeypatch):
    """round 4: the evaluation reads the query's own cell first and another cell of its +-threshold box only while the face it shares with
    the own cell is nearer than the best distance so far (IBL_EVAL_FULLSCAN=1: every cell).  The bound must hold where floorf(x / cell)
    and the geometric cell faces disagree by rounding: points 240 m from the origin, queries on cell faces, outliers, an empty job"""
    from ibloc_amd.registration import MemGrid, evaluate_points
    rng = np.random.default_rng(91)
    base = np.array([241.37, -229.91, 3.03])
    mem = (rng.uniform(-0.4, 0.4, size=(60000, 3)) * [1, 1, 0.02] + base).astype(np.float32)              # a noisy slab
    lattice = (np.stack(np.meshgrid(np.arange(40), np.arange(40), indexing="ij"), -1).reshape(-1, 2) * 0.01 + base[:2] + 0.5).astype(np.float32)
    mem = np.concatenate([mem, np.concatenate([lattice, np.full((len(lattice), 1), base[2], np.float32)], 1)])
    det = np.concatenate([mem[rng.integers(0, len(mem), 20000)] + rng.normal(0, 0.004, size=(20000, 3)).astype(np.float32),
                          (np.round(mem[rng.integers(0, len(mem), 4000)] / 0.04) * 0.04).astype(np.float32),       # on cell faces / corners
                          (rng.uniform(-1, 1, size=(4000, 3)) + base).astype(np.float32)])                           # mostly outliers
    mem4 = torch.from_numpy(np.concatenate([mem, np.zeros((len(mem), 1), np.float32)], 1)).cuda()
    det4 = torch.from_numpy(np.concatenate([det, np.zeros((len(det), 1), np.float32)], 1)).cuda()
    grid = MemGrid(ctx, mem4, 0.04)
    T = np.stack([np.eye(4), np.eye(4), np.eye(4)])
    T[1][:3, 3] = [0.013, -0.007, 0.004]
    jb, je = [0, 0, 5], [len(det), len(det), 5]
    d_p, rmse_p, fit_p = evaluate_points(ctx, grid, det4, jb, je, T, 0.02)
    monkeypatch.setenv("IBL_EVAL_FULLSCAN", "1")
    d_f, rmse_f, fit_f = evaluate_points(ctx, grid, det4, jb, je, T, 0.02)
    torch.cuda.synchronize()
    assert torch.equal(d_p, d_f) and np.array_equal(rmse_p, rmse_f) and np.array_equal(fit_p, fit_f)
    assert 0.5 < fit_p[0] < 0.99 and fit_p[2] == 0
    grid.close()


@pytest.mark.parametrize("spacing", [2.5, 1.0])
def test_register_cached_is_bit_identical(ctx, spacing):
    """instance features (per-cloud normals / FPFH / gradients reused across jobs) must not change a single bit of the
    results: far-apart instances are served from the cache, instances within the influence radius are recomputed"""
    from ibloc_amd.registration import CloudBatch, instance_features_batch, register_batch
    w = SynthWorld(9, pts_per_object=2500, E=1, D=8, seed=31, spacing=spacing)
    rng = np.random.default_rng(32)
    f = w.make_frame(rng, q=3, pts_per_object=2500, anchor=4)
    ids = f["ids"]
    det = CloudBatch.from_numpy([c[0] for c in f["clouds"]], [ro.intensity(c[1]) for c in f["clouds"]])
    mem = CloudBatch.from_numpy(w.points, [ro.intensity(c) for c in w.colors])
    js = [[0, -1, -1], [1, -1, -1], [0, 1, -1], [1, 0, -1], [0, 1, 2], [0, 1, -1], [2, 0, -1]]
    jt = [[ids[0], -1, -1], [ids[1], -1, -1], [ids[0], ids[1], -1], [ids[1], ids[0], -1], [ids[0], ids[1], ids[2]],
          [ids[0], ids[1], -1], [0, 8, -1]]
    base = register_batch(ctx, det, mem, js, jt, 0.05, 1.5, 1.5, seed=5, job_id_base=7)
    assert base["reuse"][0] == 0 and base["reuse"][2] < 2 * len(js)          # nothing cached; identical sides are computed once
    fd = instance_features_batch(ctx, det, 0.05)
    fm = instance_features_batch(ctx, mem, 0.05, grad_radius=2 * (0.05 * 1.5))
    assert fd.bbox.shape == (3, 6) and np.all(fd.bbox[:, 3:] > fd.bbox[:, :3])
    reused = []
    for a, b in ((fd, None), (None, fm), (fd, fm)):
        out = register_batch(ctx, det, mem, js, jt, 0.05, 1.5, 1.5, seed=5, job_id_base=7, det_features=a, mem_features=b)
        for k in ("T", "rmse", "fitness", "T_ransac", "ransac_stats", "means"):
            assert np.array_equal(base[k], out[k]), (k, spacing)
        reused.append(out["reuse"])
    print("reuse", spacing, reused)
    assert reused[2][0] > 0                                 # single-instance sides always come from the cache
    if spacing == 1.0:
        assert reused[2][1] > 0                             # overlapping instances were recomputed in context
    assert ctx.status() & 1 == 0


def test_instance_features_equal_standalone_and_errors(ctx, scene):
    from ibloc_amd import _lib
    from ibloc_amd.registration import CloudBatch, instance_features_batch, normals_fpfh_batch, register_batch
    w, f = scene
    mem = CloudBatch.from_numpy(w.points[:4], [ro.intensity(c) for c in w.colors[:4]])
    ft = instance_features_batch(ctx, mem, 0.05, grad_radius=0.15)
    nrm, fp = normals_fpfh_batch(ctx, mem, 0.1, 30, 0.25, 100)
    from ibloc_amd.registration import FEAT_ORDER
    assert torch.equal(ft.normals[:mem.n], nrm)
    assert torch.equal(ft.fpfh[:mem.n], fp[:, torch.from_numpy(FEAT_ORDER).cuda()])      # rows are stored in matching order
    for s in range(4):
        p = w.points[s].astype(np.float32)
        assert np.array_equal(ft.bbox[s, :3], p.min(0)) and np.array_equal(ft.bbox[s, 3:], p.max(0))
    g = ro.color_gradient(w.points[1].astype(np.float32), nrm[mem.seg_off_host[1]:mem.seg_off_host[2], :3].cpu().numpy(),
                          ro.intensity(w.colors[1]), 0.15, 30)
    got = ft.grad[mem.seg_off_host[1]:mem.seg_off_host[2], :3].cpu().numpy()
    assert np.allclose(got, g, atol=2e-4 * max(1.0, np.abs(g).max()))
    det = CloudBatch.from_numpy([c[0] for c in f["clouds"]], [ro.intensity(c[1]) for c in f["clouds"]])
    with pytest.raises(_lib.IblError):                      # features of another voxel size
        register_batch(ctx, det, mem, [[0, -1, -1]], [[0, -1, -1]], 0.04, 1.5, 1.5, mem_features=ft)
    with pytest.raises(_lib.IblError):                      # gradients of another radius
        register_batch(ctx, det, mem, [[0, -1, -1]], [[0, -1, -1]], 0.05, 1.5, 1.0, mem_features=ft)


def test_instance_features_chunked_equals_single_pass(ctx):
    """more than 2^20 points: ibl_instance_features_batch walks the clouds in chunks of whole clouds with rebased offsets;
    every chunk must reproduce what one pass over its clouds gives"""
    from ibloc_amd.registration import CloudBatch, FEAT_ORDER, instance_features_batch, normals_fpfh_batch
    rng = np.random.default_rng(5)
    w = SynthWorld(4, pts_per_object=6000, E=1, D=8, seed=41)
    clouds, ints = [], []
    for k in range(200):                                    # 200 clouds x 6000 points = 1.2 M points -> two chunks
        p = w.points[k % 4] + rng.normal(0, 0.5, size=3)
        clouds.append(p)
        ints.append(ro.intensity(w.colors[k % 4]))
    big = CloudBatch.from_numpy(clouds, ints)
    assert big.n > (1 << 20)
    ft = instance_features_batch(ctx, big, 0.05, grad_radius=0.15)
    order = torch.from_numpy(FEAT_ORDER).cuda()
    for lo, hi in ((0, 3), (172, 176), (197, 200)):         # first chunk, across the chunk boundary (cloud 174), last clouds
        sub = CloudBatch.from_numpy(clouds[lo:hi], ints[lo:hi])
        nrm, fp = normals_fpfh_batch(ctx, sub, 0.1, 30, 0.25, 100)
        b, e = big.seg_off_host[lo], big.seg_off_host[hi]
        assert torch.equal(ft.normals[b:e], nrm) and torch.equal(ft.fpfh[b:e], fp[:, order])
        one = instance_features_batch(ctx, sub, 0.05, grad_radius=0.15)
        assert torch.equal(ft.grad[b:e], one.grad[:sub.n]) and np.array_equal(ft.bbox[lo:hi], one.bbox)
    assert ctx.status() & 1 == 0


def test_matrix_core_feature_search_equals_valu_search(ctx):
    """the MFMA-filtered nearest-feature search (fp16 products with the norms folded in + exact recheck) against the full VALU scan it replaces
    (IBL_FEAT_VALU=1): every output of the registration must be bit-identical -- the filter may only drop rows that cannot be
    the exact minimum.  Includes a degenerate job (a cloud registered onto itself: every distance 0 is an exact tie)."""
    import os
    from ibloc_amd.registration import CloudBatch, instance_features_batch, register_batch
    w = SynthWorld(9, pts_per_object=3000, E=1, D=8, seed=91, spacing=1.4)
    rng = np.random.default_rng(92)
    f = w.make_frame(rng, q=3, pts_per_object=3000, anchor=4)
    ids = f["ids"]
    det = CloudBatch.from_numpy([c[0] for c in f["clouds"]] + [w.points[0]], [ro.intensity(c[1]) for c in f["clouds"]] + [ro.intensity(w.colors[0])])
    mem = CloudBatch.from_numpy(w.points, [ro.intensity(c) for c in w.colors])
    js = [[0, -1, -1], [0, 1, -1], [0, 1, 2], [2, 1, -1], [3, -1, -1], [1, -1, -1]]
    jt = [[ids[0], -1, -1], [ids[0], ids[1], -1], [ids[0], ids[1], ids[2]], [8, 0, -1], [0, -1, -1], [7, -1, -1]]
    fd = instance_features_batch(ctx, det, 0.05)
    fm = instance_features_batch(ctx, mem, 0.05, grad_radius=0.15)
    # the fp16 search operands: the CENTRED row to 2^-11 relative, then 8 8 | |x|^2 / 8 as hi + lo | C |x|^2 + A rounded up | zeros; norms
    # exact to fp32 rounding; centring shrinks the norms (the filter's band: to 0.33 - 0.44 on 5 000-point objects, 0.7 on these sparser ones)
    from ibloc_amd.registration import FEAT_MU
    raw = fm.fpfh[:mem.n].double()
    rows = (fm.fpfh[:mem.n] - torch.from_numpy(FEAT_MU).cuda()).double()
    assert (rows * rows).sum().item() < 0.8 * (raw * raw).sum().item()
    op = fm.fpfh_split[:mem.n].double()
    assert ((op[:, :33] - rows).abs() <= 2.0 ** -11 * rows.abs() + 1e-7).all()
    nrm = fm.fpfh_norm[:mem.n].double()
    assert torch.allclose(nrm, (rows * rows).sum(1), rtol=1e-5)
    assert (op[:, 33] == 8).all() and (op[:, 34] == 8).all() and torch.count_nonzero(op[:, 38:]).item() == 0
    assert ((op[:, 35] + op[:, 36]) * 8 - nrm).abs().max().item() <= 2.0 ** -20 * nrm.max().item()
    assert (op[:, 37] >= (1.0e-3 * nrm + 4.0e-3) * (1 - 1e-6)).all() and (op[:, 37] <= (1.0e-3 * nrm + 4.0e-3) * (1 + 2.0 ** -9) + 1e-6).all()
    outs = []
    for env in ({}, {"IBL_FEAT_VALU": "1"}, {"IBL_FEAT_CAND_CAP": "100"}):        # matrix cores | VALU scan | overflow -> fallback
        os.environ.update(env)
        try:
            outs.append(register_batch(ctx, det, mem, js, jt, 0.05, 1.5, 1.5, seed=3, job_id_base=40, det_features=fd, mem_features=fm))
        finally:
            for k in env:
                os.environ.pop(k, None)
    # compact memory features (no resident fp16 operand rows: the search builds them from the fp32 rows as it stages them) -- the same
    # bits in the operands, so the same candidates and the same results
    fc = instance_features_batch(ctx, mem, 0.05, grad_radius=0.15, compact=True)
    assert fc.fpfh_split is None and torch.equal(fc.fpfh[:mem.n], fm.fpfh[:mem.n]) and torch.equal(fc.fpfh_norm[:mem.n], fm.fpfh_norm[:mem.n])
    outs.append(register_batch(ctx, det, mem, js, jt, 0.05, 1.5, 1.5, seed=3, job_id_base=40, det_features=fd, mem_features=fc))
    for k in ("T", "rmse", "fitness", "T_ransac", "ransac_stats"):
        assert np.array_equal(outs[0][k], outs[1][k]) and np.array_equal(outs[0][k], outs[2][k]) and np.array_equal(outs[0][k], outs[3][k]), k
    # the self-registration converges to the identity
    assert np.allclose(outs[0]["T"][4], np.eye(4), atol=1e-6) and outs[0]["fitness"][4] > 0.999


def test_ransac_survivor_list_overflow_is_redone_with_a_full_list(ctx):
    """a round in which more hypotheses pass the edge-length test than the survivor list holds (sized for ~6 % of a round)
    used to fail the whole batch.  Clouds registered onto exact copies of themselves: every correspondence is right, so every
    hypothesis survives; a fixed budget keeps all four jobs running into the 262 144-hypothesis round, whose 1 M survivors exceed the
    list.  The call must redo itself with a full-size list (status bit 5) and return what the same jobs give one at a time (a single
    job's rounds fit the list)."""
    from ibloc_amd.registration import CloudBatch, register_batch
    w = SynthWorld(4, pts_per_object=1500, E=1, D=8, seed=17, spacing=3.0)
    ints = [ro.intensity(c) for c in w.colors]
    det = CloudBatch.from_numpy(w.points, ints)
    mem = CloudBatch.from_numpy(w.points, ints)
    js = [[j, -1, -1] for j in range(4)]
    kw = dict(seed=5, ransac_max_iter=300000, fixed_budget=True)
    ctx.status()                                                          # clear
    out = register_batch(ctx, det, mem, js, js, 0.05, 1.5, 1.5, job_ids=np.arange(4, dtype=np.uint32) + 7, **kw)
    assert ctx.status() & 32, "the survivor list did not overflow: the case no longer exercises the redo"
    for j in range(4):
        one = register_batch(ctx, det, mem, [js[j]], [js[j]], 0.05, 1.5, 1.5, job_ids=np.array([7 + j], dtype=np.uint32), **kw)
        assert ctx.status() & 32 == 0
        for k in ("T", "rmse", "fitness", "T_ransac", "ransac_stats"):
            assert np.array_equal(out[k][j], one[k][0]), (j, k)
        assert np.allclose(out["T"][j], np.eye(4), atol=1e-6) and out["fitness"][j] > 0.999

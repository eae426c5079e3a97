"""The oracle on data the reference itself ships (tests/golden/ref_scene: a memory it saved and RGB-D views of the room it was built
from).  These are the only reference-held inputs for the stages it delegates to Open3D, so they pin what can be pinned without
Open3D: the unprojection / pose conventions against the reference's OWN saved output, and the registration chain on real densities
(11 k / 54 k / 9.8 k-point instances) against the ground-truth camera poses of poses.json."""
import numpy as np
import pytest
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation

from oracle import depth_oracle as do
from oracle import reg_oracle as ro
from tests import ref_scene as rs


def test_unprojected_views_land_on_the_saved_memory():
    """depth -> camera-frame points (utils/depth_utils.py:46-90) -> world frame with the dataloader's pose
    (dataloader/synthetic_dataloader.py:49-59, utils/depth_utils.py:92-116) must reproduce the clouds the reference saved: each
    object of its memory was merged from these views, so a large share of its points is matched to < 1 cm by a single view"""
    objs = rs.memory_objects()
    expect = {1: (0, 0.6), 8: (0, 0.6), 3: (1, 0.35)}             # view -> (object, share of its points the view explains)
    vs = rs.views()
    for k in expect:
        depth, rgb, pose = vs[k]
        pts, cols = do.coloured_pointcloud_from_depth(depth, rgb, rs.FX, rs.FY)
        T = rs.pose_matrix(pose)
        world = pts.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
        j, share = expect[k]
        d, i = cKDTree(world).query(objs[j][0], k=1)
        assert np.mean(d < 0.01) > share, (k, j, float(np.mean(d < 0.01)))
        # ... and the colours agree where the geometry does (u8 / 255 on both sides)
        near = d < 0.002
        assert near.sum() > 500 and np.abs(cols[i[near]] - objs[j][1][near]).mean() < 0.05


def test_real_clouds_survive_outlier_removal_and_have_real_density():
    objs = rs.memory_objects()
    for p, _ in objs:
        keep = ro.radius_outlier(p.astype(np.float32), 0.05, 8)
        assert keep.mean() > 0.99                                 # the reference down-samples its memory at 5 mm: dense surfaces


@pytest.mark.parametrize("view,objs_seen", [(8, [0]), (1, [0, 2])])
def test_oracle_registers_a_real_view_against_the_saved_objects(view, objs_seen):
    """detections of a real view (camera frame) against the objects the reference saved (world frame): the recovered transform is
    the view's camera pose within the reference's success rule (0.6 m / 0.3 rad, tum_localisation_trial.py:274).  The table alone is
    symmetric about the vertical axis (its single-object registration lands half a turn off, on either implementation), which is why
    the reference registers assignments of up to three objects: view 1 uses the armchair and the table together."""
    objs = rs.memory_objects()
    clouds, pose = rs.view_detections(view, objs_seen)
    det, cols = [], []
    for pts, inten in clouds:
        keep = ro.radius_outlier(pts, 0.05, 8)
        assert keep.sum() > 2000
        det.append(pts[keep])
        cols.append(np.repeat(inten[keep][:, None], 3, axis=1))
    assn = [[d, j] for d, j in enumerate(objs_seen)]
    est, recs, best = ro.localise_from_assignments(det, cols, [o[0] for o in objs], [o[1] for o in objs], [assn], 0.05, 1.5, 1.5, seed=3,
                                                   stale_means=False)
    T = rs.pose_matrix(pose)
    terr = np.linalg.norm(est[:3] - T[:3, 3])
    R = Rotation.from_quat(est[3:]).as_matrix()
    rerr = np.arccos(np.clip((np.trace(R.T @ T[:3, :3]) - 1) / 2, -1, 1))
    assert terr < 0.6 and rerr < 0.3, (terr, rerr)
    assert recs[best]["fitness"] > 0.5 and recs[best]["full_fitness"] > 0.5


def _fpfh_independent(P, N, radius, max_nn):
    """Open3D's FPFH formulas (Feature.cpp ComputeFPFHFeature) restated with array arithmetic in float64 on
    scipy's kd-tree neighbourhoods -- shares no code, summation order or search structure with oracle/oracle_reg.c."""
    n = len(P)
    d, idx = cKDTree(P).query(P, k=max_nn)
    ok = d < radius                                                   # hybrid search: at most max_nn neighbours, strictly inside the radius
    cnt = ok.sum(1)
    d2 = d * d
    p1, n1 = P[:, None, :], N[:, None, :]
    p2, n2 = P[idx], N[idx]
    dv = p2 - p1
    r = np.linalg.norm(dv, axis=2)
    rs_ = np.where(r > 0, r, 1.0)
    a1 = (n1 * dv).sum(2) / rs_
    a2 = (n2 * dv).sum(2) / rs_
    swap = np.arccos(np.clip(np.abs(a1), 0, 1)) > np.arccos(np.clip(np.abs(a2), 0, 1))
    na = np.where(swap[..., None], n2, n1)
    nb = np.where(swap[..., None], n1, n2)
    dd = np.where(swap[..., None], -dv, dv)
    f3 = np.where(swap, -a2, a1)
    v = np.cross(dd, na)
    vn = np.linalg.norm(v, axis=2)
    good = (r > 0) & (vn > 0)
    v = v / np.where(vn > 0, vn, 1.0)[..., None]
    w = np.cross(na, v)
    f1 = np.arctan2((w * nb).sum(2), (na * nb).sum(2))
    f2 = (v * nb).sum(2)
    f1, f2, f3 = (np.where(good, f, 0.0) for f in (f1, f2, f3))
    h = np.stack([np.floor(11 * (f1 + np.pi) / (2 * np.pi)), np.floor(11 * (f2 + 1) * 0.5), np.floor(11 * (f3 + 1) * 0.5)], axis=2)
    h = np.clip(h, 0, 10).astype(np.int64) + np.array([0, 11, 22])
    use = ok & (idx != np.arange(n)[:, None]) & (cnt > 1)[:, None]
    inc = np.where(cnt > 1, 100.0 / np.maximum(cnt - 1, 1), 0.0)
    spfh = np.zeros((n, 33))
    rows = np.repeat(np.arange(n), max_nn * 3).reshape(n, max_nn, 3)
    np.add.at(spfh, (rows[use], h[use]), np.repeat(inc, max_nn * 3).reshape(n, max_nn, 3)[use])
    wgt = np.where(use & (d2 > 0), 1.0 / np.where(d2 > 0, d2, 1.0), 0.0)
    acc = (spfh[idx] * wgt[..., None]).sum(1)
    s = acc.reshape(n, 3, 11).sum(2)
    sc = np.where(s != 0, 100.0 / np.where(s != 0, s, 1.0), 0.0)
    out = acc * np.repeat(sc, 11, axis=1) + spfh
    out[cnt <= 1] = 0.0
    return out, cnt


def test_oracle_fpfh_on_a_real_object_matches_an_independent_restatement():
    """the independent checks of the registration oracle ran on 400-point synthetic clouds only.  Here: the
    reference's own 11 209-point object (5 mm surface sampling: every 0.25 m neighbourhood saturates the 100-neighbour cap, the
    regime the hot path lives in), normals from the oracle, FPFH from an array restatement that shares nothing with it."""
    P32 = rs.memory_objects()[0][0].astype(np.float32)
    nrm = ro.normals(P32, 0.1, 30)
    got = ro.fpfh(P32, nrm, 0.25, 100)
    exp, cnt = _fpfh_independent(P32.astype(np.float64), nrm.astype(np.float64), 0.25, 100)
    assert np.mean(cnt == 100) > 0.99                                  # saturated neighbourhoods
    row_err = np.abs(got - exp).max(1)
    # equidistant neighbours at the 100th place (a 5 mm lattice has many) may be chosen differently, and a pair feature within an
    # ulp of a bin edge may fall on either side: a few rows move by one histogram count (100 / 99 per count, before weighting)
    # measured: median 3.7e-6, 99.78 % of the rows within 1e-2, 99.99 % within 0.5
    assert np.median(row_err) < 1e-5 and np.mean(row_err < 1e-2) > 0.995 and np.mean(row_err < 0.5) > 0.9995, (
        float(np.median(row_err)), float(np.mean(row_err < 1e-2)), float(np.mean(row_err < 0.5)))
    sums = got.reshape(len(got), 3, 11).sum(-1)
    assert np.all(np.abs(sums - 200) < 0.05)


# ---- round 3: all eight views of the reference's room ---------------------------------------------------------------------------
# tests/golden/ref_scene/oracle_views.json = the oracle's localise() transcript on every view (tools/gen_golden_ref_views.py, 15 minutes of
# CPU); the -m gpu test compares the HIP path with it view by view.
ORACLE_FAILS = {2, 7}      # views the ORACLE itself does not localise (see test_every_view_of_the_room_against_its_ground_truth_pose)


def test_every_view_of_the_room_against_its_ground_truth_pose():
    """the reference's success rule (trans < 0.6 m and rot < 0.3 rad, tum_localisation_trial.py:274) against poses.json for all eight
    views.  Six localise (within 2 cm / 0.02 rad); views 2 and 7 do not, on the oracle and on the device alike: the only well-seen
    object there is the large armchair from behind (6 900 / 16 400 points against 54 k in memory) and its registration lands half a turn
    off -- the reference's own log reports 54 of 86 frames localised (new_codebase_results.log:9484-10085)."""
    o = rs.oracle_views()
    assert sorted(int(k) for k in o["views"]) == list(range(1, 9))
    for k, v in o["views"].items():
        ok = v["gt_err_m"] < 0.6 and v["gt_err_rad"] < 0.3
        assert ok == (int(k) not in ORACLE_FAILS), (k, v["gt_err_m"], v["gt_err_rad"])
        if ok:
            assert v["gt_err_m"] < 0.03 and v["gt_err_rad"] < 0.03


def test_stored_oracle_transcript_equals_a_rerun():
    """the stored transcript is what the oracle computes today: view 4 (one 9.5 k-point detection, 15 s) recomputed"""
    o = rs.oracle_views()
    fr = rs.view_frames()[4]
    v = o["views"]["4"]
    cleaned, ccols = [], []
    for pts, inten in zip(fr["clouds"], fr["ints"]):
        keep = ro.radius_outlier(pts, 0.05, 8)
        cleaned.append(pts[keep])
        ccols.append(np.repeat(inten[keep][:, None], 3, axis=1))
    assert [len(c) for c in cleaned] == v["n_clean"] and fr["seen"] == v["seen"]
    assns = rs.oracle_assignments(fr["det_emb"])
    assert assns == v["assignments"]
    objs = rs.memory_objects()
    pose, recs, best = ro.localise_from_assignments(cleaned, ccols, [ob[0] for ob in objs], [ob[1] for ob in objs], assns, 0.05, 1.5, 1.5,
                                                    seed=o["seed"], job_base=v["job_base"], stale_means=False)
    assert best == v["best"] and np.allclose(pose, v["pose"], atol=1e-9)
    assert np.allclose([r["full_fitness"] for r in recs], v["full_fitness"], atol=1e-12)

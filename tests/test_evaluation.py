"""Evaluation harness: pose-error metric against vectors of the reference's own QuaternionOps, the TUM pose
convention and the result file against the restated driver loop (oracle/eval_oracle.py)."""
import json
import os

import numpy as np
from scipy.spatial.transform import Rotation

from ibloc_amd import evaluation as E
from ibloc_amd.utils.quaternion_ops import QuaternionOps as Q
from oracle import eval_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "eval_golden.json")


def test_quaternion_ops_match_reference_vectors():
    cases = json.load(open(GOLD))["cases"]
    assert len(cases) == 64
    for c in cases:
        assert np.array_equal(Q.quaternion_multiply(c["q1"], c["q2"]), c["product"])
        assert np.array_equal(Q.quaternion_conjugate(c["q1"]), c["conjugate"])
        assert Q.quaternion_error(c["q1"], c["q2"]) == c["error"]
        assert O.quaternion_error(c["q1"], c["q2"]) == c["error"]          # the oracle is pinned too


def test_tum_pose_convention(tmp_path):
    rng = np.random.default_rng(3)
    lines = []
    for _ in range(25):
        t, q = rng.normal(size=3), Rotation.random(random_state=rng).as_quat()
        lines.append(" ".join(repr(float(v)) for v in np.concatenate([t, q])))
    for ln in lines:
        got, want = E.tum_pose(ln.split()), O.tum_pose(ln)
        assert np.array_equal(got, want)
        # half turn about y on the right, translation negated
        R = Rotation.from_quat([float(v) for v in ln.split()[3:]]).as_matrix() @ np.diag([-1.0, 1.0, -1.0])
        assert np.allclose(Rotation.from_quat(got[3:]).as_matrix(), R, atol=1e-12)
    p = tmp_path / "groundtruth.txt"
    p.write_text("# tx ty tz qx qy qz qw\n" + "\n".join(lines) + "\n\n")
    poses = E.load_tum_groundtruth(str(p), start_file_index=2, last_file_index=23, sampling_period=5)
    assert len(poses) == len(lines[2:23:5])
    for got, ln in zip(poses, lines[2:23:5]):
        assert np.array_equal(got, O.tum_pose(ln))
    try:
        E.tum_pose("1 2 3 4".split())
        assert False
    except ValueError:
        pass


def test_report_text_matches_driver_loop(tmp_path):
    rng = np.random.default_rng(11)
    rep = E.LocalisationReport()
    tg, es, ch = [], [], []
    scales = [0.01, 0.05, 0.2, 0.5, 0.8, 1.2, 2.0, 4.0]
    for i in range(40):
        t = np.concatenate([rng.normal(size=3), Rotation.random(random_state=rng).as_quat()])
        s = scales[i % len(scales)]
        dq = (Rotation.from_rotvec(rng.normal(size=3) * s) * Rotation.from_quat(t[3:])).as_quat()
        e = np.concatenate([t[:3] + rng.normal(size=3) * s * 0.6, dq])
        a = ([[int(rng.integers(0, 5)), int(rng.integers(0, 50))] for _ in range(int(rng.integers(1, 4)))], [int(rng.integers(0, 50))])
        tg.append(t), es.append(e), ch.append(a)
        te, re = rep.add(t, e, a)
        assert te == float(np.linalg.norm(t[:3] - e[:3]))
    # boundary values sit in the 'other' bins / fail the strict success test
    t0 = np.array([0, 0, 0, 1.0, 0, 0, 0])
    for e0 in (np.array([3.0, 0, 0, 1.0, 0, 0, 0]), np.array([0.6, 0, 0, 1.0, 0, 0, 0]), np.array([0, 0, 0, np.cos(1.5), np.sin(1.5), 0, 0])):
        tg.append(t0), es.append(e0), ch.append(([[0, 1]], []))
        rep.add(t0, e0, ([[0, 1]], []))
    want, te, re = O.report_text(tg, es, ch)
    assert rep.text() == want
    assert rep.trans_errors == [float(v) for v in te] and rep.rot_errors == [float(v) for v in re]
    d, r = rep.bins()
    assert d["other"] >= 1 and d["3.0"] + d["other"] == len(rep) and r["1.5"] + r["other"] == len(rep)
    assert not rep.success(len(rep) - 2)                    # translation error exactly 0.6
    out = tmp_path / "results.txt"
    rep.write(str(out))
    assert out.read_text() == want
    s = rep.summary()
    assert s["total"] == 43 and 0 < s["successes"] < 43


def test_floor_phrase_check_matches_reference_vectors():
    from ibloc_amd.object_memory.object_finder_phrases import check_if_floor
    cases = json.load(open(GOLD))["check_if_floor"]
    assert len(cases) == 12 and any(c["is_floor"] for c in cases) and not all(c["is_floor"] for c in cases)
    for c in cases:
        assert check_if_floor(c["names"]) == c["is_floor"], c

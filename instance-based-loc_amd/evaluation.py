"""Evaluation harness of the reference's TUM driver: ground-truth pose convention, per-query pose errors and
the result file.  Host-side only (no device work): it exists so that a run through `ObjectMemory.localise` prints the same
result file as `/root/reference/tum_localisation_trial.py:231-344`.

* `tum_pose` / `load_tum_groundtruth`  — `/root/reference/dataloader/tum_dataloader.py:57-78` (Kinect world frame:
  translation negated, rotation right-multiplied by a half turn about y; pose = x y z qx qy qz qw)
* `pose_errors`                        — `tum_localisation_trial.py:231-232`
* `LocalisationReport`                 — `tum_localisation_trial.py:254-344` (per-query lines, cumulative bins, means,
  medians, success rate; success = translation error < 0.6 and rotation error < 0.3)
"""
import numpy as np
from scipy.spatial.transform import Rotation

from .utils.quaternion_ops import QuaternionOps

TRANSLATION_BINS = ("0.1", "0.3", "0.6", "1.0", "1.5", "3.0")
ROTATION_BINS = ("0.1", "0.3", "0.6", "1.0", "1.5")
SUCCESS_TRANSLATION, SUCCESS_ROTATION = 0.6, 0.3


def tum_pose(fields):
    """One ground-truth row (tx ty tz qx qy qz qw) -> the pose the reference localises against."""
    f = [float(v) for v in fields]
    if len(f) != 7:
        raise ValueError(f"a ground-truth row has 7 fields (tx ty tz qx qy qz qw), got {len(f)}")
    half_turn = Rotation.from_euler("xyz", [0, np.pi, 0]).as_matrix()
    q = Rotation.from_matrix(Rotation.from_quat(f[3:]).as_matrix() @ half_turn).as_quat()
    return np.array([-f[0], -f[1], -f[2], q[0], q[1], q[2], q[3]], dtype=np.float64)


def load_tum_groundtruth(path, start_file_index=0, last_file_index=None, sampling_period=10):
    """Poses of `groundtruth.txt`, subsampled like the image lists (`tum_dataloader.py:51-55,78`).  Blank lines and `#`
    comments are skipped (the reference would stop on them)."""
    poses = []
    with open(path, "r") as fh:
        for line in fh:
            s = line.split()
            if not s or s[0].startswith("#"):
                continue
            poses.append(tum_pose(s))
    return poses[start_file_index:last_file_index:sampling_period]


def pose_errors(target_pose, estimated_pose):
    """(translation error, rotation error).  The quaternion halves are handed to `QuaternionOps.quaternion_error` exactly as the
    reference hands them (elements 3..6 of each pose, whatever their component order)."""
    t, e = np.asarray(target_pose, dtype=np.float64), np.asarray(estimated_pose, dtype=np.float64)
    return float(np.linalg.norm(t[:3] - e[:3])), float(QuaternionOps.quaternion_error(t[3:], e[3:]))


class LocalisationReport:
    def __init__(self):
        self.targets, self.estimates, self.trans_errors, self.rot_errors, self.assignments = [], [], [], [], []

    def add(self, target_pose, estimated_pose, chosen_assignment):
        """`chosen_assignment` = the second value `localise()` returns: (assignment, moved objects)."""
        te, re = pose_errors(target_pose, estimated_pose)
        self.targets.append(np.asarray(target_pose, dtype=np.float64))
        self.estimates.append(np.asarray(estimated_pose, dtype=np.float64).tolist())
        self.trans_errors.append(te)
        self.rot_errors.append(re)
        self.assignments.append(chosen_assignment)
        return te, re

    def __len__(self):
        return len(self.trans_errors)

    def success(self, i):
        return self.trans_errors[i] < SUCCESS_TRANSLATION and self.rot_errors[i] < SUCCESS_ROTATION

    def bins(self):
        """Cumulative counts per threshold plus 'other' (>= the last threshold)."""
        d = {k: 0 for k in TRANSLATION_BINS + ("other",)}
        r = {k: 0 for k in ROTATION_BINS + ("other",)}
        for te, re in zip(self.trans_errors, self.rot_errors):
            for k in TRANSLATION_BINS:
                d[k] += te < float(k)
            d["other"] += not te < float(TRANSLATION_BINS[-1])
            for k in ROTATION_BINS:
                r[k] += re < float(k)
            r["other"] += not re < float(ROTATION_BINS[-1])
        return d, r

    def summary(self):
        n = len(self)
        return {"total": n, "successes": sum(self.success(i) for i in range(n)),
                "avg_translation_error": sum(self.trans_errors) / n, "avg_rotation_error": sum(self.rot_errors) / n,
                "median_translation_error": float(np.median(self.trans_errors)),
                "median_rotation_error": float(np.median(self.rot_errors))}

    def text(self):
        n = len(self)
        out = []
        for i in range(n):
            out.append(f"Pose {i + 1}, image {n}\n")
            out.append(f"Translation error: {self.trans_errors[i]}\n")
            out.append(f"Rotation errors: {self.rot_errors[i]}\n")
            out.append(f"Assignment: {self.assignments[i][0]}\n")
            out.append(f"Moved objects: {self.assignments[i][1]}\n")
            out.append("SUCCESS\n" if self.success(i) else "MISALIGNED\n")
            out.append("\n")
        d, r = self.bins()
        out.append(f"Bagged results for {n} eval indices\n")
        for k in TRANSLATION_BINS:
            out.append(f"Translation error less than {k}: {d[k]}\n")
        out.append(f"Translation error greater than {TRANSLATION_BINS[-1]}: {d['other']}\n")
        out.append("\n")
        for k in ROTATION_BINS:
            out.append(f"Rotation error less than {k}: {r[k]}\n")
        out.append(f"Rotation error greater than {ROTATION_BINS[-1]}: {r['other']}\n")
        out.append("\n")
        s = self.summary()
        out.append(f"Average Translation Error: {s['avg_translation_error']}\n")
        out.append(f"Average Rotation Error: {s['avg_rotation_error']}\n")
        out.append(f"Median Translation Error: {np.median(self.trans_errors)}\n")
        out.append(f"Median Rotation Error: {np.median(self.rot_errors)}\n")
        out.append(f"Total Success Rate: {s['successes'] / s['total'] * 100}\n")
        return "".join(out)

    def write(self, path):
        with open(path, "w") as fh:
            fh.write(self.text())

"""TEST INFRASTRUCTURE ONLY.  Literal restatement of the reference driver's result accounting
(/root/reference/tum_localisation_trial.py:231-232,254-344) and of the TUM ground-truth convention
(/root/reference/dataloader/tum_dataloader.py:57-78).  The quaternion metric inside is pinned by vectors produced by the
reference's own utils/quaternion_ops.py (tests/golden/eval_golden.json, tools/gen_golden_eval.py); the driver and the
dataloader themselves cannot be imported here (they import the perception stack / open3d), so the report text is a
restatement checked line by line against the reference's f-strings."""
import numpy as np
from scipy.spatial.transform import Rotation


def quaternion_multiply(q1, q2):              # utils/quaternion_ops.py:5-12
    w1, x1, y1, z1 = q1
    w2, x2, y2, z2 = q2
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def quaternion_error(q1, q2):                 # utils/quaternion_ops.py:21-25
    w, x, y, z = q1
    c = np.array([w, -x, -y, -z])
    a = quaternion_multiply(c, np.asarray(q2))
    b = quaternion_multiply(c, -np.asarray(q2))
    return min(np.abs(np.arctan2(np.linalg.norm(a[1:]), a[0])), np.abs(np.arctan2(np.linalg.norm(b[1:]), b[0])))


def tum_pose(line):                            # dataloader/tum_dataloader.py:57-76
    R2 = Rotation.from_euler('xyz', [0, np.pi, 0]).as_matrix()
    split_pose = line.split()
    R1 = Rotation.from_quat([float(i) for i in split_pose[3:]]).as_matrix()
    q = Rotation.from_matrix(R1 @ R2).as_quat()
    split_pose[:3] = [-float(i) for i in split_pose[:3]]
    split_pose[3:] = q
    return np.array([float(i) for i in split_pose])


def report_text(targets, estimates, chosen_assignments):   # tum_localisation_trial.py:231-232,254-344
    trans_errors = [np.linalg.norm(np.asarray(t)[:3] - np.asarray(e)[:3]) for t, e in zip(targets, estimates)]
    rot_errors = [quaternion_error(np.asarray(t)[3:], np.asarray(e)[3:]) for t, e in zip(targets, estimates)]
    n = len(targets)
    f = []
    d_tr = {'0.1': 0, '0.3': 0, '0.6': 0, '1.0': 0, '1.5': 0, '3.0': 0, 'other': 0}
    r_tr = {'0.1': 0, '0.3': 0, '0.6': 0, '1.0': 0, '1.5': 0, 'other': 0}
    total = successes = 0
    avg_t = avg_r = 0
    for idx in range(n):
        ok = trans_errors[idx] < 0.6 and rot_errors[idx] < 0.3
        successes += ok
        total += 1
        f.append(f"Pose {idx + 1}, image {n}\n")
        f.append(f"Translation error: {trans_errors[idx]}\n")
        f.append(f"Rotation errors: {rot_errors[idx]}\n")
        f.append(f"Assignment: {chosen_assignments[idx][0]}\n")
        f.append(f"Moved objects: {chosen_assignments[idx][1]}\n")
        f.append("SUCCESS\n" if ok else "MISALIGNED\n")
        avg_t += trans_errors[idx]
        avg_r += rot_errors[idx]
        for k in ('0.1', '0.3', '0.6', '1.0', '1.5'):
            if trans_errors[idx] < float(k):
                d_tr[k] += 1
        if trans_errors[idx] < 3.0:
            d_tr['3.0'] += 1
        else:
            d_tr['other'] += 1
        for k in ('0.1', '0.3', '0.6', '1.0'):
            if rot_errors[idx] < float(k):
                r_tr[k] += 1
        if rot_errors[idx] < 1.5:
            r_tr['1.5'] += 1
        else:
            r_tr['other'] += 1
        f.append("\n")
    f.append(f"Bagged results for {n} eval indices\n")
    for k in ('0.1', '0.3', '0.6', '1.0', '1.5', '3.0'):
        f.append(f"Translation error less than {k}: {d_tr[k]}\n")
    f.append(f"Translation error greater than 3.0: {d_tr['other']}\n")
    f.append("\n")
    for k in ('0.1', '0.3', '0.6', '1.0', '1.5'):
        f.append(f"Rotation error less than {k}: {r_tr[k]}\n")
    f.append(f"Rotation error greater than 1.5: {r_tr['other']}\n")
    f.append("\n")
    f.append(f"Average Translation Error: {avg_t / total}\n")
    f.append(f"Average Rotation Error: {avg_r / total}\n")
    f.append(f"Median Translation Error: {np.median(trans_errors)}\n")
    f.append(f"Median Rotation Error: {np.median(rot_errors)}\n")
    f.append(f"Total Success Rate: {successes / total * 100}\n")
    return "".join(f), trans_errors, rot_errors

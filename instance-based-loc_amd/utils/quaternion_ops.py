"""Pose-error metric of the reference's drivers (/root/reference/utils/quaternion_ops.py:3-25, used by
tum_localisation_trial.py to report the rotation error of a localisation): quaternions are (w, x, y, z)."""
import numpy as np


class QuaternionOps:
    @staticmethod
    def quaternion_multiply(q1, q2):
        """Hamilton product q1 * q2, scalar-first."""
        a, b = np.asarray(q1, dtype=np.float64), np.asarray(q2, dtype=np.float64)
        s = a[0] * b[0] - np.dot(a[1:], b[1:])
        v = a[0] * b[1:] + b[0] * a[1:] + np.cross(a[1:], b[1:])
        return np.concatenate(([s], v))

    @staticmethod
    def quaternion_conjugate(q):
        q = np.asarray(q, dtype=np.float64)
        return q * np.array([1.0, -1.0, -1.0, -1.0])

    @staticmethod
    def quaternion_error(q1, q2):
        """Angle in [0, pi/2] of the relative rotation conj(q1) * q2, taking the nearer of q2 and -q2 (half the rotation angle
        between the two orientations, as the reference reports it)."""
        d = QuaternionOps.quaternion_multiply(QuaternionOps.quaternion_conjugate(q1), q2)
        ang = np.abs(np.arctan2(np.linalg.norm(d[1:]), d[0]))
        ang_neg = np.abs(np.arctan2(np.linalg.norm(d[1:]), -d[0]))
        return min(ang, ang_neg)

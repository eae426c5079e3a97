"""Pose-error metric of the reference's drivers (/root/reference/utils/quaternion_ops.py:3-25, used by
tum_localisation_trial.py to report the rotation error of a localisation): quaternions are (w, x, y, z)."""
import numpy as np


class QuaternionOps:
    @staticmethod
    def quaternion_multiply(q1, q2):
        """Hamilton product q1 * q2, scalar-first.  Each component is summed left to right in the reference's term order, so
        that errors (and the result files that print them) agree to the last digit."""
        a0, a1, a2, a3 = (float(v) for v in q1)
        b0, b1, b2, b3 = (float(v) for v in q2)
        return np.array([a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3,
                         a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2,
                         a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1,
                         a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0])

    @staticmethod
    def quaternion_conjugate(q):
        q = np.asarray(q, dtype=np.float64)
        return q * np.array([1.0, -1.0, -1.0, -1.0])

    @staticmethod
    def quaternion_error(q1, q2):
        """Angle in [0, pi/2] of the relative rotation conj(q1) * q2, taking the nearer of q2 and -q2 (half the rotation angle
        between the two orientations, as the reference reports it)."""
        c = QuaternionOps.quaternion_conjugate(q1)
        q2 = np.asarray(q2, dtype=np.float64)
        d, e = QuaternionOps.quaternion_multiply(c, q2), QuaternionOps.quaternion_multiply(c, -q2)
        return min(np.abs(np.arctan2(np.linalg.norm(d[1:]), d[0])), np.abs(np.arctan2(np.linalg.norm(e[1:]), e[0])))

"""utils.quaternion_ops (the reference's pose-error metric) against scipy's rotation algebra."""
import numpy as np
from scipy.spatial.transform import Rotation

from ibloc_amd.utils.quaternion_ops import QuaternionOps as Q


def _wxyz(r):
    x, y, z, w = r.as_quat()
    return np.array([w, x, y, z])


def test_product_conjugate_and_error():
    rng = np.random.default_rng(0)
    for _ in range(20):
        a, b = Rotation.random(random_state=rng), Rotation.random(random_state=rng)
        got = Q.quaternion_multiply(_wxyz(a), _wxyz(b))
        want = _wxyz(a * b)
        assert np.allclose(got, want) or np.allclose(got, -want)
        assert np.allclose(Q.quaternion_multiply(_wxyz(a), Q.quaternion_conjugate(_wxyz(a))), [1, 0, 0, 0])
        rel = (a.inv() * b).magnitude()                       # rotation angle in [0, pi]
        half = min(rel, 2 * np.pi - rel) / 2
        assert abs(Q.quaternion_error(_wxyz(a), _wxyz(b)) - half) < 1e-9
        assert abs(Q.quaternion_error(_wxyz(a), -_wxyz(b)) - half) < 1e-9
    assert Q.quaternion_error([1, 0, 0, 0], [1, 0, 0, 0]) == 0.0

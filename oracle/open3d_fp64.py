"""ORACLE (test infrastructure; never imported by the product path).

An INDEPENDENT fp64 restatement of the Open3D 0.17.0 stages the reference calls around its registration --
`remove_radius_outlier` (object_memory/object_memory.py:994-995), `estimate_normals` (utils/fpfh_register.py:91-92), the colour
gradients and one Gauss-Newton step of `registration_colored_icp` (utils/fpfh_register.py:132-135) and `evaluate_registration`
(utils/fpfh_register.py:146-148) -- written on scipy's cKDTree and numpy only.  It shares no code and none of the conventions of
oracle/oracle_reg.c and the device (those work on fp32 coordinates with an fp32 fmaf distance, a uniform grid, fixed summation orders
and a fast analytic eigen solver): here the coordinates, the distances and every reduction are float64, neighbours come from a kd-tree,
eigenvectors from LAPACK (`numpy.linalg.eigh`) and the 6 x 6 system from `numpy.linalg.solve`.  Its purpose is to COUNT how many
discrete decisions (kept points, 30-neighbour sets, inlier sets at a threshold) change between "Open3D's arithmetic" (fp64 kd-tree,
SURVEY App. A) and the fp32 rule the oracle and the device share, on the reference's own objects and views.

**parity unpinned at the Open3D boundary**: Open3D itself is neither importable nor present as source in this environment; this file
follows the public 0.17.0 algorithms like oracle_reg.c does, independently of it."""
import numpy as np
from scipy.spatial import cKDTree


def radius_outlier_keep(points, radius=0.05, nb_points=8):
    """PointCloud.remove_radius_outlier(nb_points, radius): a point is kept when its radius search (itself included) returns MORE than
    nb_points points.  -> (keep mask, neighbour counts)."""
    p = np.asarray(points, dtype=np.float64)
    tree = cKDTree(p)
    # nanoflann's radius search keeps squared distances strictly below radius^2; count through the tree, then settle the (measure-zero
    # in fp64) boundary explicitly
    cnt = tree.query_ball_point(p, r=radius, return_length=True)
    on_edge = np.nonzero(tree.query_ball_point(p, r=radius * (1 + 1e-12), return_length=True) !=
                         tree.query_ball_point(p, r=radius * (1 - 1e-12), return_length=True))[0]
    for i in on_edge:
        idx = tree.query_ball_point(p[i], r=radius * (1 + 1e-12))
        d2 = np.sum((p[idx] - p[i]) ** 2, axis=1)
        cnt[i] = int(np.sum(d2 < radius * radius))
    return cnt > nb_points, cnt


def hybrid_neighbours(points, radius, max_nn, queries=None):
    """KDTreeSearchParamHybrid(radius, max_nn): the <= max_nn nearest points with squared distance < radius^2, nearest first (the query
    itself first when it belongs to the cloud).  -> (idx [Q][max_nn] padded with -1, count [Q], squared distances padded with inf)."""
    p = np.asarray(points, dtype=np.float64)
    q = p if queries is None else np.asarray(queries, dtype=np.float64)
    k = min(max_nn, len(p))
    d, i = cKDTree(p).query(q, k=k)
    if k == 1:
        d, i = d[:, None], i[:, None]
    d2 = np.sum((p[np.minimum(i, len(p) - 1)] - q[:, None, :]) ** 2, axis=2)
    ok = np.isfinite(d) & (d2 < radius * radius)
    idx = np.where(ok, i, -1)
    if k < max_nn:
        idx = np.concatenate([idx, -np.ones((len(q), max_nn - k), dtype=idx.dtype)], axis=1)
        d2 = np.concatenate([d2, np.full((len(q), max_nn - k), np.inf)], axis=1)
        ok = idx >= 0
    return idx, ok.sum(axis=1), np.where(ok, d2, np.inf)


def normals(points, radius=0.1, max_nn=30):
    """estimate_normals(KDTreeSearchParamHybrid(radius, max_nn)) without prior normals: covariance of the neighbours about their mean,
    eigenvector of the smallest eigenvalue; fewer than 3 neighbours -> (0, 0, 1).  The SIGN of a normal is whatever Open3D's solver
    returns (no orientation step): compare directions up to sign.  -> (normals [N][3], neighbour idx, neighbour count)."""
    p = np.asarray(points, dtype=np.float64)
    idx, cnt, _ = hybrid_neighbours(p, radius, max_nn)
    valid = idx >= 0
    nb = p[np.maximum(idx, 0)]                                   # [N][k][3]
    w = valid[..., None].astype(np.float64)
    n = np.maximum(cnt, 1)[:, None]
    mean = (nb * w).sum(axis=1) / n
    c = (nb - mean[:, None, :]) * w
    cov = np.einsum("nki,nkj->nij", c, c) / n[:, :, None]
    _, vec = np.linalg.eigh(cov)
    out = vec[:, :, 0].copy()
    out[cnt < 3] = (0.0, 0.0, 1.0)
    return out, idx, cnt


def color_gradients(points, normals_, intensity, radius, max_nn=30):
    """InitializePointCloudForColoredICP(target, KDTreeSearchParamHybrid(radius, 30)): per point with >= 4 neighbours the least-squares
    intensity gradient over the neighbours projected on the tangent plane, with the extra row (nn - 1) * n -> 0 that keeps the gradient
    orthogonal to the normal; zero otherwise."""
    p = np.asarray(points, dtype=np.float64)
    nrm = np.asarray(normals_, dtype=np.float64)
    it = np.asarray(intensity, dtype=np.float64)
    idx, cnt, _ = hybrid_neighbours(p, radius, max_nn)
    out = np.zeros_like(p)
    for i in np.nonzero(cnt >= 4)[0]:
        nn = int(cnt[i])
        adj = p[idx[i, 1:nn]]
        proj = adj - ((adj - p[i]) @ nrm[i])[:, None] * nrm[i][None, :]
        A = np.concatenate([proj - p[i], (nn - 1) * nrm[i][None, :]], axis=0)
        b = np.concatenate([it[idx[i, 1:nn]] - it[i], [0.0]])
        out[i] = np.linalg.lstsq(A, b, rcond=None)[0]
    return out


def nearest_within(target_tree, target_points, queries, max_dist):
    """per query the nearest target point, accepted when its squared distance is < max_dist^2 -> (index or -1, squared distance)"""
    d, j = target_tree.query(queries, k=1)
    d2 = np.sum((target_points[np.minimum(j, len(target_points) - 1)] - queries) ** 2, axis=1)
    ok = np.isfinite(d) & (d2 < max_dist * max_dist)
    return np.where(ok, j, -1), d2


def evaluate_registration(source, target, T, max_dist=0.02):
    """-> (fitness, inlier_rmse, inlier mask): transform the source, nearest target point within max_dist per source point."""
    s = np.asarray(source, dtype=np.float64)
    t = np.asarray(target, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64).reshape(4, 4)
    moved = s @ T[:3, :3].T + T[:3, 3]
    j, d2 = nearest_within(cKDTree(t), t, moved, max_dist)
    inl = j >= 0
    n = int(inl.sum())
    return n / max(1, len(s)), (float(np.sqrt(d2[inl].sum() / n)) if n else 0.0), inl


def vector6_to_matrix(x):
    """TransformVector6dToMatrix4d: R = Rz(x[2]) Ry(x[1]) Rx(x[0]), t = x[3:6]"""
    a, b, c = x[0], x[1], x[2]
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = x[3:6]
    return T


def colored_icp_step(source, source_intensity, target, target_normals, target_intensity, target_gradients, T, max_dist,
                     lambda_geometric=0.968):
    """ONE iteration of registration_colored_icp from the transform T: correspondences (nearest target within max_dist of the moved
    source point), the two residuals per correspondence
        r_G = sqrt(lambda) (s - t) . n_t
        r_I = sqrt(1 - lambda) (I_s - (I_t + dI_t . (proj_t(s) - t))),   proj_t(s) = s - ((s - t) . n_t) n_t
    with their 6-vector Jacobians [s x g, g] (g = n_t for r_G, g = -(I - n n^T) dI_t for r_I), the normal equations J^T J x = -J^T r
    and the update exp(x) T.  -> dict(T_new, x, JTJ, JTr, fitness, inlier_rmse, n_corr, corr)"""
    s = np.asarray(source, dtype=np.float64)
    t = np.asarray(target, dtype=np.float64)
    nt_all = np.asarray(target_normals, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64).reshape(4, 4)
    moved = s @ T[:3, :3].T + T[:3, 3]
    j, d2 = nearest_within(cKDTree(t), t, moved, max_dist)
    sel = np.nonzero(j >= 0)[0]
    vs, vt, nt = moved[sel], t[j[sel]], nt_all[j[sel]]
    Is = np.asarray(source_intensity, dtype=np.float64)[sel]
    It = np.asarray(target_intensity, dtype=np.float64)[j[sel]]
    dit = np.asarray(target_gradients, dtype=np.float64)[j[sel]]
    sg, sp = np.sqrt(lambda_geometric), np.sqrt(1.0 - lambda_geometric)
    diff = vs - vt
    dn = np.sum(diff * nt, axis=1)
    rG = sg * dn
    JG = sg * np.concatenate([np.cross(vs, nt), nt], axis=1)
    vs_proj = vs - dn[:, None] * nt
    is_proj = np.sum(dit * (vs_proj - vt), axis=1) + It
    ditM = -(dit - np.sum(dit * nt, axis=1)[:, None] * nt)       # -dI^T (I - n n^T)
    rI = sp * (Is - is_proj)
    JI = sp * np.concatenate([np.cross(vs, ditM), ditM], axis=1)
    JTJ = JG.T @ JG + JI.T @ JI
    JTr = JG.T @ rG + JI.T @ rI
    x = np.linalg.solve(JTJ, -JTr) if len(sel) >= 6 else np.zeros(6)
    n = len(sel)
    return dict(T_new=vector6_to_matrix(x) @ T, x=x, JTJ=JTJ, JTr=JTr, fitness=n / max(1, len(s)),
                inlier_rmse=float(np.sqrt(d2[sel].sum() / n)) if n else 0.0, n_corr=n, corr=j)

"""ORACLE (test infrastructure): ctypes wrappers of oracle/oracle_reg.c (see its header for the
reference lines restated), plus the pure-numpy pieces of ObjectMemory.localise()
(object_memory/object_memory.py:1023-1034 centring, :1096-1131 un-centring, selection and pose)."""
import ctypes as C

import numpy as np
from scipy.spatial.transform import Rotation

from .clib import lib

_vp = C.c_void_p


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return _vp(a.ctypes.data) if a is not None else None


def radius_outlier(pts, radius, nb_points):
    pts = _f32(pts)
    keep = np.zeros(len(pts), dtype=np.uint8)
    lib.oracle_radius_outlier(_p(pts), C.c_int(len(pts)), C.c_double(radius), C.c_int(nb_points), _p(keep))
    return keep.astype(bool)


def normals(pts, radius, max_nn=30):
    pts = _f32(pts)
    out = np.zeros_like(pts)
    lib.oracle_normals(_p(pts), C.c_int(len(pts)), C.c_double(radius), C.c_int(max_nn), _p(out))
    return out


def fpfh(pts, nrm, radius, max_nn=100):
    pts, nrm = _f32(pts), _f32(nrm)
    out = np.zeros((len(pts), 33), dtype=np.float32)
    lib.oracle_fpfh(_p(pts), _p(nrm), C.c_int(len(pts)), C.c_double(radius), C.c_int(max_nn), _p(out))
    return out


def feature_match(fs, ft, mutual=True, ransac_n=3):
    fs, ft = _f32(fs), _f32(ft)
    corr = np.zeros((len(fs), 2), dtype=np.int32)
    lib.oracle_feature_match.restype = C.c_int
    n = lib.oracle_feature_match(_p(fs), C.c_int(len(fs)), _p(ft), C.c_int(len(ft)), C.c_int(int(mutual)), C.c_int(ransac_n),
                                 _p(corr))
    return corr[:n].copy()


def ransac(src, tgt, corr, max_dist, seed=0, job_id=0, max_iter=4000000, confidence=0.99, edge_sim=0.9):
    src, tgt = _f32(src), _f32(tgt)
    corr = np.ascontiguousarray(corr, dtype=np.int32)
    T = np.zeros(16, dtype=np.float64)
    stats = np.zeros(3, dtype=np.int64)
    lib.oracle_ransac(_p(src), _p(tgt), _p(corr), C.c_int(len(corr)), C.c_double(max_dist), C.c_uint64(seed), C.c_uint32(job_id),
                      C.c_int(max_iter), C.c_double(confidence), C.c_double(edge_sim), _p(T), _p(stats))
    return T.reshape(4, 4), stats


def color_gradient(pts, nrm, intensity, radius, max_nn=30):
    pts, nrm, intensity = _f32(pts), _f32(nrm), _f32(intensity)
    out = np.zeros_like(pts)
    lib.oracle_color_gradient(_p(pts), _p(nrm), _p(intensity), C.c_int(len(pts)), C.c_double(radius), C.c_int(max_nn), _p(out))
    return out


def icp(src, src_int, tgt, tgt_nrm, tgt_int, tgt_grad, max_dist, T_init, colored=True, lambda_geometric=0.968, max_iter=30,
        rel_fitness=1e-6, rel_rmse=1e-6):
    src, tgt = _f32(src), _f32(tgt)
    T0 = np.ascontiguousarray(T_init, dtype=np.float64).reshape(16)
    T = np.zeros(16, dtype=np.float64)
    fit, rmse, iters = C.c_double(), C.c_double(), C.c_int()
    args = [None if a is None else _f32(a) for a in (src_int, tgt_nrm, tgt_int, tgt_grad)]
    lib.oracle_icp(_p(src), _p(args[0]), C.c_int(len(src)), _p(tgt), _p(args[1]), _p(args[2]), _p(args[3]), C.c_int(len(tgt)),
                   C.c_double(max_dist), _p(T0), C.c_int(int(colored)), C.c_double(lambda_geometric), C.c_int(max_iter),
                   C.c_double(rel_fitness), C.c_double(rel_rmse), _p(T), C.byref(fit), C.byref(rmse), C.byref(iters))
    return T.reshape(4, 4), fit.value, rmse.value, iters.value


def evaluate(src, tgt, T, max_dist=0.02):
    src, tgt = _f32(src), _f32(tgt)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    fit, rmse = C.c_double(), C.c_double()
    lib.oracle_evaluate(_p(src), C.c_int(len(src)), _p(tgt), C.c_int(len(tgt)), _p(T), C.c_double(max_dist), C.byref(fit),
                        C.byref(rmse))
    return rmse.value, fit.value        # order of utils/fpfh_register.py:150


def hybrid_sets(pts, radius, max_nn):
    """the fp32 rule's neighbour sets: (idx [n][max_nn] padded with -1 in (d2, index) order, count [n])"""
    pts = _f32(pts)
    idx = np.full((len(pts), max_nn), -1, dtype=np.int32)
    cnt = np.zeros(len(pts), dtype=np.int32)
    lib.oracle_hybrid_sets(_p(pts), C.c_int(len(pts)), C.c_double(radius), C.c_int(max_nn), _p(idx), _p(cnt))
    return idx, cnt


def correspondences(src, tgt, T, max_dist):
    """the fp32 rule's accepted correspondence (target index or -1) of every transformed source point"""
    src, tgt = _f32(src), _f32(tgt)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    corr = np.zeros(len(src), dtype=np.int32)
    lib.oracle_correspondences(_p(src), C.c_int(len(src)), _p(tgt), C.c_int(len(tgt)), _p(T), C.c_double(max_dist), _p(corr))
    return corr


def register_point_clouds(src, src_int, tgt, tgt_int, voxel_size, global_dist_factor=1.5, local_dist_factor=0.4, seed=0,
                          job_id=0, ransac_max_iter=4000000, have_colors=True, src_raw=None, tgt_raw=None):
    """== utils/fpfh_register.py:100-143; returns (T 4x4, inlier_rmse, fitness, T_ransac, ransac_stats).
    src_raw / tgt_raw: the clouds before the caller centred them (features are evaluated there, see oracle_reg.c)."""
    src, tgt = _f32(src), _f32(tgt)
    sr = _f32(src_raw) if src_raw is not None else None
    tr = _f32(tgt_raw) if tgt_raw is not None else None
    assert sr is None or sr.shape == src.shape
    assert tr is None or tr.shape == tgt.shape
    T = np.zeros(16, dtype=np.float64)
    Tr = np.zeros(16, dtype=np.float64)
    stats = np.zeros(3, dtype=np.int64)
    fit, rmse = C.c_double(), C.c_double()
    si = _f32(src_int) if src_int is not None else None
    ti = _f32(tgt_int) if tgt_int is not None else None
    lib.oracle_register_raw(_p(src), _p(sr), _p(si), C.c_int(len(src)), _p(tgt), _p(tr), _p(ti), C.c_int(len(tgt)), C.c_double(voxel_size),
                        C.c_double(global_dist_factor), C.c_double(local_dist_factor), C.c_int(int(have_colors)), C.c_uint64(seed),
                        C.c_uint32(job_id), C.c_int(ransac_max_iter), _p(T), C.byref(rmse), C.byref(fit), _p(Tr), _p(stats))
    return T.reshape(4, 4), rmse.value, fit.value, Tr.reshape(4, 4), stats


def intensity(colors):
    colors = np.asarray(colors, dtype=np.float64)
    return ((colors[:, 0] + colors[:, 1] + colors[:, 2]) / 3.0).astype(np.float32)


def localise_from_assignments(det_clouds, det_cols, mem_clouds, mem_cols, assns, voxel_size=0.05, global_dist_factor=1.5,
                              local_dist_factor=1.5, seed=0, job_base=0, ransac_max_iter=4000000, stale_means=True):
    """Numpy/C transcript of object_memory/object_memory.py:1000-1131 given cleaned detected clouds,
    memory clouds and the assignment list.  Returns (pose[7], per-assignment records, best index).

    Clouds are rounded to fp32 first (the product's HBM layout).  `stale_means=True` reproduces the
    reference's use of the LAST assignment's means in the final pose."""
    det32 = [np.asarray(c, dtype=np.float32) for c in det_clouds]
    mem32 = [np.asarray(c, dtype=np.float32) for c in mem_clouds]
    all_mem = np.concatenate(mem32)
    all_det = np.concatenate(det32)
    recs = []
    detected_mean = memory_mean = None
    for a_i, assn in enumerate(assns):
        cd = np.concatenate([det32[d] for d, m in assn]).astype(np.float64)
        cm = np.concatenate([mem32[m] for d, m in assn]).astype(np.float64)
        ci_d = np.concatenate([intensity(det_cols[d]) for d, m in assn])
        ci_m = np.concatenate([intensity(mem_cols[m]) for d, m in assn])
        detected_mean = np.mean(cd, axis=0)
        memory_mean = np.mean(cm, axis=0)
        src = (cd - detected_mean).astype(np.float32)
        tgt = (cm - memory_mean).astype(np.float32)
        T, rmse, fit, Tr, stats = register_point_clouds(src, ci_d, tgt, ci_m, voxel_size, global_dist_factor, local_dist_factor,
                                                        seed=seed, job_id=job_base + a_i, ransac_max_iter=ransac_max_iter,
                                                        src_raw=cd.astype(np.float32), tgt_raw=cm.astype(np.float32))
        G = T.copy()
        R = T[:3, :3]
        G[:3, 3] = T[:3, 3] + memory_mean - R @ detected_mean
        full_rmse, full_fit = evaluate(all_det, all_mem, G, 0.02)
        recs.append(dict(assn=assn, T=T, rmse=rmse, fitness=fit, full_rmse=full_rmse, full_fitness=full_fit, T_global=G,
                         T_ransac=Tr, detected_mean=detected_mean, memory_mean=memory_mean))
    order = sorted(range(len(recs)), key=lambda i: recs[i]["full_fitness"], reverse=True)
    best = order[0]
    R = recs[best]["T"][:3, :3]
    t = recs[best]["T"][:3, 3]
    if stale_means:
        tAvg = t + memory_mean - R @ detected_mean                      # means of the LAST assignment (:1127)
    else:
        tAvg = t + recs[best]["memory_mean"] - R @ recs[best]["detected_mean"]
    q = Rotation.from_matrix(R).as_quat()
    return np.concatenate((tAvg, q)), recs, best

"""Host wrappers of the registration C-ABI (device tensors in, C calls out).

Cloud batches are `(pts4, seg_off)`: pts4 a float32 CUDA tensor (N, 4) = (x, y, z, intensity) and
seg_off an int32 tensor (S + 1,) of segment boundaries (host copy kept alongside)."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


class RegContext:
    """Owns the device arena of the registration kernels (ibl_reg_ctx)."""

    def __init__(self, arena_bytes=4 << 30):
        self._h = C.c_void_p()
        _lib.check(_lib.lib.ibl_reg_ctx_create(C.byref(self._h), int(arena_bytes)), "ibl_reg_ctx_create")

    def close(self):
        if self._h:
            _lib.lib.ibl_reg_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def reset(self):
        _lib.check(_lib.lib.ibl_reg_ctx_reset(self._h), "ibl_reg_ctx_reset")

    def status(self, clear=True):
        return _lib.lib.ibl_reg_ctx_status(self._h, 1 if clear else 0)

    def high_water(self):
        return _lib.lib.ibl_reg_ctx_high_water(self._h)


class CloudBatch:
    """Packed clouds on the device."""

    def __init__(self, pts4: torch.Tensor, seg_off_host: np.ndarray):
        assert pts4.is_cuda and pts4.dtype == torch.float32 and pts4.dim() == 2 and pts4.shape[1] == 4 and pts4.is_contiguous()
        self.pts4 = pts4
        self.seg_off_host = np.ascontiguousarray(seg_off_host, dtype=np.int32)
        assert self.seg_off_host[-1] == pts4.shape[0]
        self.seg_off = torch.from_numpy(self.seg_off_host).to(pts4.device)

    @property
    def n_seg(self):
        return len(self.seg_off_host) - 1

    @property
    def n(self):
        return int(self.seg_off_host[-1])

    @staticmethod
    def from_numpy(clouds, intensities=None, device="cuda"):
        """clouds: list of (n_i, 3) arrays; intensities: list of (n_i,) or None (zeros)."""
        sizes = [len(c) for c in clouds]
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        p4 = np.zeros((int(off[-1]), 4), dtype=np.float32)
        for i, c in enumerate(clouds):
            p4[off[i]:off[i + 1], :3] = np.asarray(c, dtype=np.float32)
            if intensities is not None:
                p4[off[i]:off[i + 1], 3] = np.asarray(intensities[i], dtype=np.float32)
        return CloudBatch(torch.from_numpy(p4).to(device), off)


def unproject_masks(ctx: RegContext, depth: torch.Tensor, rgb: torch.Tensor, masks: torch.Tensor, fx: float, fy: float,
                    depth_factor: float = 1.0, want_f64: bool = False):
    """One coloured cloud per instance mask from a depth image (get_mask_coloured_pointclouds_from_depth,
    utils/depth_utils.py:176-206, before its outlier step).  depth (H, W) float32, float64 or uint16 (int16 storage is read as uint16), rgb (H, W, 3)
    uint8, masks (n, H, W) bool / uint8 -- device tensors.  Returns the clouds as a CloudBatch (x, y, z, intensity); with want_f64 also
    (points, colours) as (N, 3) float64 device tensors holding exactly the values of the reference's Open3D clouds (memory build)."""
    dev = depth.device
    assert depth.is_cuda and rgb.is_cuda and masks.is_cuda and depth.dim() == 2
    H, W = depth.shape
    if depth.dtype in (torch.uint16, torch.int16):         # int16 storage = the bits of a uint16 image
        is_u16, d = 1, depth.contiguous()
    elif depth.dtype == torch.float32:
        is_u16, d = 0, depth.contiguous()                  # numpy keeps float32 arithmetic for a float32 depth image
    else:
        is_u16, d = 2, depth.to(torch.float64).contiguous()
    rgb = rgb.to(torch.uint8).contiguous()
    m = masks.reshape(-1, H, W).to(torch.uint8).contiguous()
    n = m.shape[0]
    assert rgb.shape == (H, W, 3)
    cap = int(m.count_nonzero().item()) if n else 0           # a mask pixel yields at most one point
    pts4 = torch.empty((max(cap, 1), 4), dtype=torch.float32, device=dev)
    off_dev = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    off_host = np.zeros(n + 1, dtype=np.int32)
    p64 = torch.empty((max(cap, 1), 3), dtype=torch.float64, device=dev) if want_f64 else None
    c64 = torch.empty((max(cap, 1), 3), dtype=torch.float64, device=dev) if want_f64 else None
    st = _lib.lib.ibl_unproject_masks_f64(ctx.handle, d.data_ptr(), is_u16, rgb.data_ptr(), m.data_ptr(), n, H, W, float(fx), float(fy),
                                          float(depth_factor), pts4.data_ptr(), p64.data_ptr() if want_f64 else None,
                                          c64.data_ptr() if want_f64 else None, cap, off_dev.data_ptr(), off_host.ctypes.data, _stream())
    _lib.check(st, "ibl_unproject_masks_f64")
    batch = CloudBatch(pts4[:int(off_host[-1])].contiguous() if off_host[-1] != cap or cap == 0 else pts4, off_host)
    if want_f64:
        return batch, p64[:int(off_host[-1])], c64[:int(off_host[-1])]
    return batch


def radius_outlier_batch(ctx: RegContext, batch: CloudBatch, radius: float, nb_points: int) -> torch.Tensor:
    keep = torch.empty(max(batch.n, 1), dtype=torch.uint8, device=batch.pts4.device)
    st = _lib.lib.ibl_radius_outlier_batch(ctx.handle, batch.pts4.data_ptr(), batch.seg_off.data_ptr(),
                                           batch.seg_off_host.ctypes.data, batch.n_seg, float(radius), int(nb_points),
                                           keep.data_ptr(), _stream())
    _lib.check(st, "ibl_radius_outlier_batch")
    return keep[:batch.n]


def normals_fpfh_batch(ctx: RegContext, batch: CloudBatch, radius_normal, max_nn_normal=30, radius_feature=None,
                       max_nn_feature=100):
    dev = batch.pts4.device
    normals = torch.empty((max(batch.n, 1), 4), dtype=torch.float32, device=dev)
    fpfh = torch.empty((max(batch.n, 1), 33), dtype=torch.float32, device=dev) if radius_feature else None
    st = _lib.lib.ibl_normals_fpfh_batch(ctx.handle, batch.pts4.data_ptr(), batch.seg_off.data_ptr(),
                                         batch.seg_off_host.ctypes.data, batch.n_seg, float(radius_normal), int(max_nn_normal),
                                         float(radius_feature or 0.0), int(max_nn_feature), normals.data_ptr(),
                                         fpfh.data_ptr() if fpfh is not None else None, _stream())
    _lib.check(st, "ibl_normals_fpfh_batch")
    return normals[:batch.n], (fpfh[:batch.n] if fpfh is not None else None)


# "matching order" of the 33 FPFH bins (histogram centres outwards, interleaved): instance features store their rows in this
# order, the feature search sums its squared differences in it (csrc/reg_knn.hip FEAT_POS, oracle/oracle_reg.c FEAT_ORDER)
FEAT_ORDER = np.array([b * 11 + c for c in (5, 4, 6, 3, 7, 2, 8, 1, 9, 0, 10) for b in (1, 2, 0)], dtype=np.int64)
# the constant the fp16 search operands are centred by (csrc/reg_common.h FM_MU; matching order)
FEAT_MU = np.array([87, 46, 101, 26, 28, 17, 26, 26, 17, 14, 21, 7, 14, 19, 7, 8, 14, 6, 8, 13, 6, 5, 11, 6, 5, 10, 6, 3, 7, 14, 3, 6, 14],
                   dtype=np.float32)


class _FeatStruct(C.Structure):
    _fields_ = [("normals4", C.c_void_p), ("fpfh", C.c_void_p), ("fpfh_split", C.c_void_p), ("fpfh_norm", C.c_void_p),
                ("grad4", C.c_void_p), ("bbox", C.c_void_p),
                ("voxel_size", C.c_double), ("grad_radius", C.c_double)]


class InstanceFeatures:
    """Registration features of every cloud of a batch, resident on the device (ibl_instance_features): normals,
    FPFH and (for memory instances) colour gradients, plus host bounding boxes.  fpfh_split is None for COMPACT features (168 instead
    of 264 bytes per point): the feature search then builds its fp16 operands from the fp32 rows while it stages them."""

    def __init__(self, normals, fpfh, fpfh_split, fpfh_norm, grad, bbox, voxel_size, grad_radius):
        self.normals, self.fpfh, self.grad, self.bbox = normals, fpfh, grad, bbox
        self.fpfh_split, self.fpfh_norm = fpfh_split, fpfh_norm
        self.voxel_size, self.grad_radius = float(voxel_size), float(grad_radius)

    def as_struct(self):
        return _FeatStruct(self.normals.data_ptr(), self.fpfh.data_ptr(), self.fpfh_split.data_ptr() if self.fpfh_split is not None else None,
                           self.fpfh_norm.data_ptr(),
                           self.grad.data_ptr() if self.grad is not None else None,
                           self.bbox.ctypes.data, self.voxel_size, self.grad_radius)


def instance_features_batch(ctx: RegContext, batch: CloudBatch, voxel_size: float, grad_radius: float = 0.0,
                            compact: bool = False) -> InstanceFeatures:
    """Normals (2 voxel, 30 nn), FPFH (5 voxel, 100 nn) and, with grad_radius > 0, colour gradients (grad_radius, 30 nn) of
    every cloud on its own, in the frame it is stored in -- what ibl_register_batch_cached reuses across jobs.
    compact: do not keep the rows a second time as fp16 search operands (96 of the 264 bytes per point; same registration results,
    the matrix-core search converts on the fly -- for memories whose resident features would not fit otherwise)."""
    dev = batch.pts4.device
    n = max(batch.n, 1)
    normals = torch.empty((n, 4), dtype=torch.float32, device=dev)
    fpfh = torch.empty((n, 33), dtype=torch.float32, device=dev)
    # the rows once more as fp16 search operands (csrc/reg_featnn.hip)
    fpfh_split = None if compact else torch.empty((n, 48), dtype=torch.float16, device=dev)
    fpfh_norm = torch.empty((n,), dtype=torch.float32, device=dev)
    grad = torch.empty((n, 4), dtype=torch.float32, device=dev) if grad_radius > 0 else None
    bbox = np.zeros((max(batch.n_seg, 1), 6), dtype=np.float32)
    st = _lib.lib.ibl_instance_features_batch(ctx.handle, batch.pts4.data_ptr(), batch.seg_off.data_ptr(), batch.seg_off_host.ctypes.data,
                                              batch.n_seg, float(voxel_size), float(grad_radius), normals.data_ptr(), fpfh.data_ptr(),
                                              fpfh_split.data_ptr() if fpfh_split is not None else None, fpfh_norm.data_ptr(),
                                              grad.data_ptr() if grad is not None else None, bbox.ctypes.data, _stream())
    _lib.check(st, "ibl_instance_features_batch")
    return InstanceFeatures(normals, fpfh, fpfh_split, fpfh_norm, grad, bbox, voxel_size, grad_radius)


REG_HAVE_COLORS = 1
REG_CENTER = 2
REG_FIXED_BUDGET = 4        # benchmark only: RANSAC walks exactly ransac_max_iter hypotheses per job (no confidence exit)


def register_batch(ctx: RegContext, det: CloudBatch, mem: CloudBatch, job_src_seg, job_tgt_seg, voxel_size,
                   global_dist_factor=1.5, local_dist_factor=0.4, seed=0, job_id_base=0, ransac_max_iter=4000000,
                   have_colors=True, center=True, det_features: InstanceFeatures = None, mem_features: InstanceFeatures = None,
                   job_ids=None, fixed_budget=False):
    """Batched register_point_clouds (utils/fpfh_register.py:100-143).  job_*_seg: (J, <=3) int arrays of
    pool segment ids (-1 padded).  det_features / mem_features: instance features of the two pools
    (instance_features_batch); the results do not depend on them, only the work does.  Returns dict of host arrays:
    T (J,4,4), rmse, fitness, means (J,2,3), T_ransac (J,4,4), ransac_stats (J,3), reuse (points served by the instance
    features, points recomputed, recomputed groups, job sides, distinct matching pairs, pair uses).
    job_ids: (J,) explicit RANSAC ids instead of job_id_base + j (ibl_register_batch_ids: jobs routed between ranks keep theirs)."""
    def pad(a):
        a = np.asarray(a, dtype=np.int32)
        if a.ndim == 1:
            a = a[:, None]
        out = np.full((a.shape[0], 3), -1, dtype=np.int32)
        out[:, :a.shape[1]] = a
        return np.ascontiguousarray(out)

    js, jt = pad(job_src_seg), pad(job_tgt_seg)
    J = js.shape[0]
    assert jt.shape[0] == J
    T = np.zeros((J, 16), dtype=np.float64)
    rmse = np.zeros(J, dtype=np.float64)
    fit = np.zeros(J, dtype=np.float64)
    means = np.zeros((J, 2, 3), dtype=np.float64)
    Tr = np.zeros((J, 16), dtype=np.float64)
    stats = np.zeros((J, 3), dtype=np.int64)
    flags = (REG_HAVE_COLORS if have_colors else 0) | (REG_CENTER if center else 0) | (REG_FIXED_BUDGET if fixed_budget else 0)
    reuse = np.zeros(6, dtype=np.int64)
    df = det_features.as_struct() if det_features is not None else None
    mf = mem_features.as_struct() if mem_features is not None else None
    if job_ids is not None:
        ids = np.ascontiguousarray(job_ids, dtype=np.uint32)
        assert ids.shape == (J,)
        st = _lib.lib.ibl_register_batch_ids(ctx.handle, det.pts4.data_ptr(), det.seg_off.data_ptr(), det.seg_off_host.ctypes.data,
                                             det.n_seg, mem.pts4.data_ptr(), mem.seg_off.data_ptr(), mem.seg_off_host.ctypes.data,
                                             mem.n_seg, js.ctypes.data, jt.ctypes.data, ids.ctypes.data, J, float(voxel_size),
                                             float(global_dist_factor), float(local_dist_factor), int(seed), int(ransac_max_iter), flags,
                                             C.byref(df) if df is not None else None, C.byref(mf) if mf is not None else None,
                                             T.ctypes.data, rmse.ctypes.data, fit.ctypes.data, means.ctypes.data, Tr.ctypes.data,
                                             stats.ctypes.data, reuse.ctypes.data, _stream())
        _lib.check(st, "ibl_register_batch_ids")
        return dict(T=T.reshape(J, 4, 4), rmse=rmse, fitness=fit, means=means, T_ransac=Tr.reshape(J, 4, 4), ransac_stats=stats,
                    reuse=reuse)
    st = _lib.lib.ibl_register_batch_cached(ctx.handle, det.pts4.data_ptr(), det.seg_off.data_ptr(), det.seg_off_host.ctypes.data,
                                            det.n_seg, mem.pts4.data_ptr(), mem.seg_off.data_ptr(), mem.seg_off_host.ctypes.data,
                                            mem.n_seg, js.ctypes.data, jt.ctypes.data, J, float(voxel_size), float(global_dist_factor),
                                            float(local_dist_factor), int(seed), int(job_id_base), int(ransac_max_iter), flags,
                                            C.byref(df) if df is not None else None, C.byref(mf) if mf is not None else None,
                                            T.ctypes.data, rmse.ctypes.data, fit.ctypes.data, means.ctypes.data, Tr.ctypes.data,
                                            stats.ctypes.data, reuse.ctypes.data, _stream())
    _lib.check(st, "ibl_register_batch_cached")
    return dict(T=T.reshape(J, 4, 4), rmse=rmse, fitness=fit, means=means, T_ransac=Tr.reshape(J, 4, 4), ransac_stats=stats,
                reuse=reuse)


def register_evaluate_batch(ctx: RegContext, det: CloudBatch, q_per_frame, assns, mem: CloudBatch, mem_features: InstanceFeatures, grid,
                            voxel_size, global_dist_factor, local_dist_factor, outlier_radius=0.05, outlier_nb_points=8, eval_threshold=0.02,
                            seed=0, job_id_base=0, ransac_max_iter=4000000, have_colors=True, center=True, fixed_budget=False):
    """Stage B of localise() in one library call (`ibl_register_evaluate_batch`, csrc/localise.hip): outlier removal + compaction, the
    detections' features, registration of every candidate assignment, whole-memory evaluation, winner per frame.  assns: per frame the
    list of assignments [[det, mem], ...] (what the assignment search returns).  Returns a dict of host arrays: clean_off (S + 1), T, rmse,
    fitness, means, T_ransac, ransac_stats, reuse, T_global, full_rmse, full_fitness (per job, frames in order) and best (per frame)."""
    q = np.ascontiguousarray(q_per_frame, dtype=np.int32)
    F = len(q)
    max_assn = max(6, max((len(a) for a in assns), default=0))
    assn = np.full((F, max_assn, 6), -1, dtype=np.int32)
    alen = np.zeros((F, max_assn), dtype=np.int32)
    acnt = np.zeros(F, dtype=np.int32)
    for f, lst in enumerate(assns):
        acnt[f] = len(lst)
        for a, pairs in enumerate(lst):
            alen[f, a] = len(pairs)
            for t, (d, m) in enumerate(pairs):
                assn[f, a, 2 * t], assn[f, a, 2 * t + 1] = d, m
    J = int(acnt.sum())
    Jc = max(J, 1)
    clean_off = np.zeros(det.n_seg + 1, dtype=np.int32)
    n_jobs = C.c_int32(0)
    T = np.zeros((Jc, 16)); rmse = np.zeros(Jc); fit = np.zeros(Jc); means = np.zeros((Jc, 2, 3)); Tr = np.zeros((Jc, 16))
    stats = np.zeros((Jc, 3), dtype=np.int64); reuse = np.zeros(6, dtype=np.int64)
    G = np.zeros((Jc, 16)); frmse = np.zeros(Jc); ffit = np.zeros(Jc); best = np.full(max(F, 1), -1, dtype=np.int32)
    flags = (REG_HAVE_COLORS if have_colors else 0) | (REG_CENTER if center else 0) | (REG_FIXED_BUDGET if fixed_budget else 0)
    mf = mem_features.as_struct()
    st = _lib.lib.ibl_register_evaluate_batch(
        ctx.handle, det.pts4.data_ptr(), det.seg_off.data_ptr(), det.seg_off_host.ctypes.data, det.n_seg, q.ctypes.data, F, assn.ctypes.data,
        alen.ctypes.data, acnt.ctypes.data, max_assn, mem.pts4.data_ptr(), mem.seg_off.data_ptr(), mem.seg_off_host.ctypes.data, mem.n_seg,
        C.byref(mf), grid.handle, float(voxel_size), float(global_dist_factor), float(local_dist_factor), float(outlier_radius),
        int(outlier_nb_points), float(eval_threshold), int(seed), int(job_id_base), int(ransac_max_iter), flags, Jc, clean_off.ctypes.data,
        C.byref(n_jobs), T.ctypes.data, rmse.ctypes.data, fit.ctypes.data, means.ctypes.data, Tr.ctypes.data, stats.ctypes.data,
        reuse.ctypes.data, G.ctypes.data, frmse.ctypes.data, ffit.ctypes.data, best.ctypes.data, _stream())
    _lib.check(st, "ibl_register_evaluate_batch")
    assert n_jobs.value == J
    return dict(clean_off=clean_off, T=T[:J].reshape(J, 4, 4), rmse=rmse[:J], fitness=fit[:J], means=means[:J], T_ransac=Tr[:J].reshape(J, 4, 4),
                ransac_stats=stats[:J], reuse=reuse, T_global=G[:J].reshape(J, 4, 4), full_rmse=frmse[:J], full_fitness=ffit[:J], best=best[:F])


class MemGrid:
    """Persistent spatial hash over all memory points (lives in the context arena)."""

    def __init__(self, ctx: RegContext, mem_pts4: torch.Tensor, cell=0.04):
        assert mem_pts4.is_cuda and mem_pts4.dtype == torch.float32 and mem_pts4.shape[1] == 4 and mem_pts4.is_contiguous()
        self.ctx = ctx
        self.cell = cell
        self._h = C.c_void_p()
        st = _lib.lib.ibl_memgrid_build(ctx.handle, mem_pts4.data_ptr(), mem_pts4.shape[0], float(cell), C.byref(self._h),
                                        _stream())
        _lib.check(st, "ibl_memgrid_build")

    def close(self):
        if self._h:
            _lib.lib.ibl_memgrid_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h


def evaluate_batch(ctx: RegContext, grid: MemGrid, det_pts4: torch.Tensor, job_begin, job_end, T_global, threshold=0.02):
    """evaluate_transform (utils/fpfh_register.py:145-150) for J candidates; returns (rmse (J,), fitness (J,))."""
    jb = np.ascontiguousarray(job_begin, dtype=np.int32)
    je = np.ascontiguousarray(job_end, dtype=np.int32)
    T = np.ascontiguousarray(T_global, dtype=np.float64).reshape(-1, 16)
    J = len(jb)
    rmse = np.zeros(J, dtype=np.float64)
    fit = np.zeros(J, dtype=np.float64)
    st = _lib.lib.ibl_evaluate_batch(ctx.handle, grid.handle, det_pts4.data_ptr(), jb.ctypes.data, je.ctypes.data, T.ctypes.data,
                                     J, float(threshold), rmse.ctypes.data, fit.ctypes.data, _stream())
    _lib.check(st, "ibl_evaluate_batch")
    return rmse, fit


def evaluate_points(ctx: RegContext, grid: MemGrid, det_pts4: torch.Tensor, job_begin, job_end, T_global, threshold=0.02):
    """Per-point form of `evaluate_batch`: squared distance of every transformed detected point to its nearest point of THIS grid within
    `threshold` (+inf when none) as one float32 device tensor (jobs back to back), plus this grid's own (rmse, fitness)."""
    jb = np.ascontiguousarray(job_begin, dtype=np.int32)
    je = np.ascontiguousarray(job_end, dtype=np.int32)
    T = np.ascontiguousarray(T_global, dtype=np.float64).reshape(-1, 16)
    J = len(jb)
    d2 = torch.empty(max(int((je - jb).sum()), 1), dtype=torch.float32, device=det_pts4.device)
    rmse = np.zeros(J, dtype=np.float64)
    fit = np.zeros(J, dtype=np.float64)
    st = _lib.lib.ibl_evaluate_points(ctx.handle, grid.handle, det_pts4.data_ptr(), jb.ctypes.data, je.ctypes.data, T.ctypes.data,
                                      J, float(threshold), d2.data_ptr(), rmse.ctypes.data, fit.ctypes.data, _stream())
    _lib.check(st, "ibl_evaluate_points")
    return d2[:int((je - jb).sum())], rmse, fit

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

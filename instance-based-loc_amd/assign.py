"""Host wrapper of ibl_assign_batch (exact similarity-volume assignment search)."""
import ctypes as C
import os

import numpy as np

from . import _lib


def assign_batch(aug_half: np.ndarray, q_per_frame, num_per_length: int = 4, n_threads: int = 0):
    """aug_half: (F, Qs, M+1) float16 `[sims | 1]` rows; q_per_frame: (F,) valid rows per frame.

    Returns a list (one per frame) of assignment lists `[[det_idx, mem_idx], ...]`, identical to the
    reference's `SimVolume(...).get_top_indices_from_subvolumes(num_per_length)`
    (utils/similarity_volume.py:213-270)."""
    aug_half = np.ascontiguousarray(aug_half, dtype=np.float16)
    assert aug_half.ndim == 3
    F, Qs, M1 = aug_half.shape
    q = np.ascontiguousarray(q_per_frame, dtype=np.int32)
    assert q.shape == (F,)
    max_assn = 8
    out_assn = np.full((F, max_assn, 3, 2), -1, dtype=np.int32)
    out_len = np.zeros((F, max_assn), dtype=np.int32)
    out_cnt = np.zeros((F,), dtype=np.int32)
    if n_threads <= 0:
        n_threads = min(os.cpu_count() or 1, 16)
    st = _lib.lib.ibl_assign_batch(aug_half.ctypes.data, q.ctypes.data, F, Qs, M1 - 1, int(num_per_length),
                                   out_assn.ctypes.data, out_len.ctypes.data, out_cnt.ctypes.data, max_assn,
                                   int(n_threads))
    _lib.check(st, "ibl_assign_batch")
    res = []
    for f in range(F):
        res.append([[[int(out_assn[f, a, p, 0]), int(out_assn[f, a, p, 1])] for p in range(out_len[f, a])]
                    for a in range(out_cnt[f])])
    return res

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


K_HI, K_LO = 192, 32          # candidates per query row and memory shard: the largest / smallest entries that leave the GPU


def assign_candidates(cand_val: np.ndarray, cand_idx: np.ndarray, cand_cnt: np.ndarray, row_first, q_per_frame, M_total: int,
                      k_hi: int = K_HI, k_lo: int = K_LO, num_per_length: int = 4, n_threads: int = 0):
    """The assignment search on per-row candidate lists (`match.match_topk`, possibly gathered from several memory shards).

    cand_val (R, S) float16 / uint16 bits, cand_idx (R, S) int32 global memory indices, cand_cnt (R,) valid entries per row;
    row_first / q_per_frame (F,): the rows of each frame.  Returns (assignments per frame, exact (F,) bool): where `exact` is
    True the list equals `assign_batch` on the full rows (proved per frame by the library, csrc/assign.cpp); the other frames
    must be redone on their full rows."""
    cand_val = np.ascontiguousarray(cand_val).view(np.uint16)
    cand_idx = np.ascontiguousarray(cand_idx, dtype=np.int32)
    cand_cnt = np.ascontiguousarray(cand_cnt, dtype=np.int32)
    assert cand_val.ndim == 2 and cand_val.shape == cand_idx.shape and cand_cnt.shape == (cand_val.shape[0],)
    row_first = np.ascontiguousarray(row_first, dtype=np.int32)
    q = np.ascontiguousarray(q_per_frame, dtype=np.int32)
    F = len(q)
    assert row_first.shape == (F,) and (F == 0 or int((row_first + q).max()) <= cand_val.shape[0])
    max_assn = 8
    out_assn = np.full((F, max_assn, 3, 2), -1, dtype=np.int32)
    out_len = np.zeros((F, max_assn), dtype=np.int32)
    out_cnt = np.zeros((F,), dtype=np.int32)
    out_exact = np.zeros((F,), dtype=np.uint8)
    if n_threads <= 0:
        n_threads = min(os.cpu_count() or 1, 16)
    st = _lib.lib.ibl_assign_candidates(cand_val.ctypes.data, cand_idx.ctypes.data, cand_cnt.ctypes.data, cand_val.shape[1],
                                        row_first.ctypes.data, q.ctypes.data, F, int(M_total), int(k_hi), int(k_lo), int(num_per_length),
                                        out_assn.ctypes.data, out_len.ctypes.data, out_cnt.ctypes.data, out_exact.ctypes.data, max_assn,
                                        int(n_threads))
    _lib.check(st, "ibl_assign_candidates")
    res = [[[[int(out_assn[f, a, p, 0]), int(out_assn[f, a, p, 1])] for p in range(out_len[f, a])] for a in range(out_cnt[f])]
           for f in range(F)]
    return res, out_exact.astype(bool)

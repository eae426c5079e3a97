"""ORACLE (test infrastructure): numpy/C restatement of normalisation + closest similarity
(object_memory/object_memory.py:922-936 of the reference).  See oracle/oracle_match.c."""
import ctypes as C

import numpy as np

from .clib import lib


def normalize_rows(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    lib.oracle_normalize_rows(C.c_void_p(x.ctypes.data), C.c_void_p(out.ctypes.data), C.c_int64(x.shape[0]),
                              C.c_int(x.shape[1]))
    return out


def closest_similarity(det, mem, emb_offsets):
    det = np.ascontiguousarray(det, dtype=np.float32)
    mem = np.ascontiguousarray(mem, dtype=np.float32)
    off = np.ascontiguousarray(emb_offsets, dtype=np.int32)
    n_inst = off.shape[0] - 1
    out = np.empty((det.shape[0], n_inst), dtype=np.float32)
    lib.oracle_closest_similarity(C.c_void_p(det.ctypes.data), C.c_int64(det.shape[0]), C.c_void_p(mem.ctypes.data),
                                  C.c_void_p(off.ctypes.data), C.c_int64(n_inst), C.c_int(det.shape[1]),
                                  C.c_void_p(out.ctypes.data))
    return out


def closest_similarity_numpy(det_raw, mem_raw_per_instance):
    """Literal numpy transcript of object_memory.py:922-936 (float32, numpy's own summation order)."""
    all_memory_embs = [np.array([e / np.linalg.norm(e) for e in m]) for m in mem_raw_per_instance]
    det = det_raw / np.linalg.norm(det_raw, axis=-1, keepdims=True)
    out = np.zeros((det.shape[0], len(all_memory_embs)), dtype=np.float32)
    for i, d in enumerate(det):
        for j, m in enumerate(all_memory_embs):
            out[i][j] = np.max(np.dot(m, d))
    return out

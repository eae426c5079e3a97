"""Host wrappers (torch device tensors in, C-ABI calls out) for normalisation + closest similarity.

Mirrors object_memory/object_memory.py:922-936 of the reference."""
import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def normalize_rows(x: torch.Tensor) -> torch.Tensor:
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
    out = torch.empty_like(x)
    _lib.check(_lib.lib.ibl_normalize_rows(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], _stream()),
               "ibl_normalize_rows")
    return out


def closest_similarity(det: torch.Tensor, mem: torch.Tensor, emb_offsets: torch.Tensor, want_sims=True,
                       want_aug=True):
    """det (Nq, D), mem (R, D) fp32 L2-normalised device tensors; emb_offsets (M+1,) int32 device.

    Returns (sims fp32 (Nq, M) or None, aug fp16 (Nq, M+1) or None)."""
    for t in (det, mem):
        assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.is_contiguous()
    assert emb_offsets.is_cuda and emb_offsets.dtype == torch.int32
    nq, dim = det.shape
    nrows = mem.shape[0]
    n_inst = emb_offsets.numel() - 1
    ws_bytes = _lib.lib.ibl_closest_similarity_workspace_bytes(nq, nrows)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=det.device)
    sims = torch.empty((nq, n_inst), dtype=torch.float32, device=det.device) if want_sims else None
    aug = torch.empty((nq, n_inst + 1), dtype=torch.float16, device=det.device) if want_aug else None
    st = _lib.lib.ibl_closest_similarity(det.data_ptr(), nq, mem.data_ptr(), nrows, emb_offsets.data_ptr(), n_inst,
                                         dim, sims.data_ptr() if want_sims else None,
                                         aug.data_ptr() if want_aug else None, ws.data_ptr(), ws_bytes, _stream())
    _lib.check(st, "ibl_closest_similarity")
    return sims, aug


def match_topk(det: torch.Tensor, mem: torch.Tensor, emb_offsets: torch.Tensor, k_hi: int, k_lo: int, index_base: int = 0,
               want_aug: bool = True):
    """Closest similarity of every query row against this rank's memory rows, then the per-row two-ended candidates
    (`ibl_match_topk`): the k_hi largest and k_lo smallest fp16 `aug` entries under (value, lower index first), as
    (cand_val (Nq, S) float16, cand_idx (Nq, S) int32 = index_base + local index, cand_cnt (Nq, 2) int32 [n_hi, n_lo]), S = k_hi + k_lo.
    A row with at most S columns is returned whole (n_hi = columns, n_lo = 0).  With want_aug also the full (Nq, M + 1) fp16 rows
    (they stay on the device; only frames whose candidate search cannot be proved exact are fetched)."""
    for t in (det, mem):
        assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.is_contiguous()
    assert emb_offsets.is_cuda and emb_offsets.dtype == torch.int32
    nq, dim = det.shape
    nrows = mem.shape[0]
    n_inst = emb_offsets.numel() - 1
    S = k_hi + k_lo
    ws_bytes = _lib.lib.ibl_match_topk_workspace_bytes(nq, nrows, n_inst)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=det.device)
    val = torch.zeros((nq, S), dtype=torch.float16, device=det.device)
    idx = torch.full((nq, S), -1, dtype=torch.int32, device=det.device)
    cnt = torch.zeros((nq, 2), dtype=torch.int32, device=det.device)
    aug = torch.empty((nq, n_inst + 1), dtype=torch.float16, device=det.device) if want_aug else None
    st = _lib.lib.ibl_match_topk(det.data_ptr(), nq, mem.data_ptr(), nrows, emb_offsets.data_ptr(), n_inst, dim, int(k_hi), int(k_lo),
                                 int(index_base), val.data_ptr(), idx.data_ptr(), cnt.data_ptr(), aug.data_ptr() if want_aug else None,
                                 ws.data_ptr(), ws_bytes, _stream())
    _lib.check(st, "ibl_match_topk")
    return val, idx, cnt, aug

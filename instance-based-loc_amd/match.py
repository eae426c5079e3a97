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

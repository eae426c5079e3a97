"""ibl_linear_f16 (the encoder's fp16 MFMA GEMM on its own) against a torch fp32 reference of the same op."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(x, W, bias, epi, out0, scale):
    y = x.float() @ W.float().t()
    if bias is not None:
        y = y + bias
    if epi == 1:
        y = torch.nn.functional.gelu(y)
    if epi == 2:
        return out0 + (y * scale if scale is not None else y)
    return y


# rows cover: ragged last tile of both tile shapes (128 / 256), the M >= 4096 switch to the 256 x 256 tile, a single row, a persistent
# grid that walks three tiles per workgroup
@pytest.mark.parametrize("rows,n_out,n_in", [(1, 128, 64), (257, 384, 192), (4096, 256, 128), (5000, 768, 768), (4100, 2304, 768),
                                               (4097, 768, 3072), (300, 3072, 768),
                                               (66000, 768, 768)])       # 774 tiles on 256 persistent workgroups: several tiles per block, the
                                                                         # pipelined read-modify-write / store epilogues back to back
@pytest.mark.parametrize("epi", [0, 1, 2, 4])
def test_linear_vs_torch(rows, n_out, n_in, epi):
    from ibloc_amd import vit as V
    g = torch.Generator(device="cpu").manual_seed(rows * 7 + n_out + epi)
    x = torch.randn(rows, n_in, generator=g).to(torch.float16).cuda()
    W = (torch.randn(n_out, n_in, generator=g) / np.sqrt(n_in)).to(torch.float16).cuda()
    bias = torch.randn(n_out, generator=g).cuda()
    scale = torch.rand(n_out, generator=g).cuda() if epi == 2 else None
    out0 = torch.randn(rows, n_out, generator=g).cuda() if epi == 2 else None
    out = out0.clone() if epi == 2 else None
    got = V.linear_f16(x, W, bias, epi, out=out, scale=scale).float()
    want = _ref(x, W, bias, epi, out0, scale)
    # fp32 accumulation of exact fp16 products: only the summation order differs; fp16 outputs add one rounding (2^-12 relative)
    tol = 2e-5 if epi in (2, 4) else 6e-4
    err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
    assert err < tol, err


def test_linear_strided_and_errors():
    from ibloc_amd import _lib, vit as V
    x_full = torch.randn(300, 256, device="cuda").to(torch.float16)
    x = x_full[:, :128]                                              # row stride 256, 128 columns used
    W = torch.randn(128, 128, device="cuda").to(torch.float16)
    got = V.linear_f16(x, W, None, V.LINEAR_F32)
    want = x.float() @ W.float().t()
    assert (got - want).abs().max().item() < 1e-3
    with pytest.raises(_lib.IblError):                               # n_out not a multiple of 128
        V.linear_f16(x, W[:100], None, V.LINEAR_F32)
    assert V.linear_f16(x[:0], W, None, V.LINEAR_F32).shape == (0, 128)

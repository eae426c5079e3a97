"""-m gpu: HIP normalisation + closest-similarity (f32 MFMA) vs the fixed-order C oracle: bit-exact."""
import numpy as np
import pytest
import torch

from oracle import match_oracle as mo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_rows,dim", [(1, 768), (5, 384), (130, 768), (7, 128), (33, 512)])
def test_normalize_rows_bit_exact(n_rows, dim):
    from ibloc_amd import match
    rng = np.random.default_rng(n_rows * 1000 + dim)
    x = rng.normal(size=(n_rows, dim)).astype(np.float32) * rng.uniform(0.1, 30)
    got = match.normalize_rows(torch.from_numpy(x).cuda()).cpu().numpy()
    exp = mo.normalize_rows(x)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


@pytest.mark.parametrize("nq,M,dim,emax", [(1, 1, 768, 1), (7, 20, 384, 4), (7, 51, 768, 20), (45, 257, 768, 5),
                                           (448, 1000, 768, 4), (130, 33, 128, 3), (3, 700, 512, 2)])
def test_closest_similarity_bit_exact(nq, M, dim, emax):
    from ibloc_amd import match
    rng = np.random.default_rng(nq * 7 + M)
    counts = rng.integers(1, emax + 1, size=M)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    mem = mo.normalize_rows(rng.normal(size=(off[-1], dim)).astype(np.float32))
    det = mo.normalize_rows(rng.normal(size=(nq, dim)).astype(np.float32))
    # make some queries near-copies of stored embeddings (peaked rows, like real matches)
    for q in range(0, nq, 3):
        det[q] = mo.normalize_rows((mem[rng.integers(0, off[-1])] + 0.1 * rng.normal(size=dim).astype(np.float32) / np.sqrt(dim))[None])[0]
    sims, aug = match.closest_similarity(torch.from_numpy(det).cuda(), torch.from_numpy(mem).cuda(),
                                         torch.from_numpy(off).cuda())
    exp = mo.closest_similarity(det, mem, off)
    got = sims.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    exp_aug = np.ones((nq, M + 1), dtype=np.float16)
    exp_aug[:, :-1] = exp                      # numpy's float32 -> float16 cast (similarity_volume.py:15-16)
    assert np.array_equal(aug.cpu().numpy().view(np.uint16), exp_aug.view(np.uint16))

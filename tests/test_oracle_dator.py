"""Pins oracle/dator_oracle.py against the reference's own build_FourDNet / TransReID (tests/golden/dator_golden.npz)."""
import os

import numpy as np

from ibloc_amd import dator as D
from oracle import dator_oracle as do

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "dator_golden.npz"))


def inputs():
    rng = np.random.default_rng(304)
    rgb = rng.normal(size=(3, 3, 256, 128)).astype(np.float32)
    depth = np.repeat(rng.uniform(-1, 1, size=(3, 1, 256, 128)).astype(np.float32), 3, axis=1)
    return rgb, depth


def test_dator_forward_matches_reference_golden():
    rw, dw, hw = D.random_stream_weights(301), D.random_stream_weights(302), D.random_head_weights(303)
    rgb, depth = inputs()
    tok = do.stream_tokens(rw, D.STREAM_CFG, rgb)
    assert np.abs(tok[:, 0] - GOLD["rgb_tokens_cls"]).max() < 2e-3 * np.abs(GOLD["rgb_tokens_cls"]).max()
    assert np.abs(tok.mean(1) - GOLD["rgb_tokens_mean"]).max() < 2e-3
    emb = do.head_forward(hw, tok, do.stream_tokens(dw, D.STREAM_CFG, depth))
    assert emb.shape == (3, 128)
    assert np.abs(emb - GOLD["embedding"]).max() < 2e-4 * max(1.0, np.abs(GOLD["embedding"]).max())

"""Pins oracle/vit_oracle.py (torch fp32 restatement of the encoder forward) against outputs of
transformers' Dinov2Model / ViTModel / CLIPVisionModelWithProjection (tests/golden/vit_golden.npz)."""
import os

import numpy as np
import pytest

from oracle import vit_oracle as vo
from tests.vit_cases import CASES, build

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "vit_golden.npz"))


@pytest.mark.parametrize("case", [c for c in CASES if c[0] != "dino_b14_full"], ids=lambda c: c[0])
def test_forward_matches_hf_golden(case):
    key, cfg, w, x = build(case)
    got = vo.vit_forward(w, cfg, x)
    exp = GOLD[key]
    assert got.shape == exp.shape
    assert np.max(np.abs(got - exp)) < 2e-4 * max(1.0, np.abs(exp).max())


def test_preprocess_matches_pil_pipeline():
    # literal HF BitImageProcessor semantics on one crop: resize(256 shortest, bicubic) -> centre crop 224 ->
    # rescale -> normalise, after the BGR2RGB swap of utils/embeddings.py:64
    from PIL import Image
    from ibloc_amd import preprocess as pp
    rng = np.random.default_rng(5)
    crop = rng.integers(0, 256, size=(180, 300, 3), dtype=np.uint8)
    r = pp.RECIPES["dinov2"]
    got = vo.preprocess_crop(crop, r)
    img = Image.fromarray(np.ascontiguousarray(crop[:, :, ::-1]))
    res = np.asarray(img.resize((int(256 * 300 / 180), 256), resample=Image.BICUBIC))
    top, left = (256 - 224) // 2, (res.shape[1] - 224) // 2
    ref = res[top:top + 224, left:left + 224].astype(np.float64) * (1 / 255)
    ref = ((ref.astype(np.float32) - np.array(pp.IMAGENET_MEAN, np.float32)) / np.array(pp.IMAGENET_STD, np.float32))
    assert np.array_equal(got, ref.transpose(2, 0, 1))

"""Shared definition of the ViT golden cases (must mirror tools/gen_golden_vit.py CASES)."""
import dataclasses

import numpy as np

from ibloc_amd import vit as V

CASES = [
    ("tiny_dino", "tiny_dino", {"pos_interp": "size"}, 101, 201, 3),
    ("dino_b14_2layer", "dinov2_vitb14", {"pos_interp": "size", "depth": 2}, 102, 202, 2),
    ("dino_s14_full", "dinov2_vits14", {"pos_interp": "size"}, 103, 203, 2),
    ("dino_b14_full", "dinov2_vitb14", {"pos_interp": "size"}, 104, 204, 2),
    ("tiny_vit16", "vit_b16", {"dim": 128, "depth": 2, "heads": 2, "mlp_dim": 256}, 105, 205, 3),
    ("tiny_clip", "tiny_clip", {"patch_bias": False}, 106, 206, 3),
    ("vit_b16_full", "vit_b16", {}, 107, 207, 2),
    ("clip_b32_full", "clip_b32", {"patch_bias": False}, 108, 208, 2),
]


def build(case):
    key, name, over, wseed, iseed, batch = case
    cfg = dataclasses.replace(V.CONFIGS[name], **{k: v for k, v in over.items() if k != "patch_bias"})
    w = V.random_weights(cfg, wseed)
    if over.get("patch_bias") is False:
        w["patch.b"] = np.zeros_like(w["patch.b"])
    x = np.random.default_rng(iseed).normal(size=(batch, 3, cfg.img_h, cfg.img_w)).astype(np.float32)
    return key, cfg, w, x

#!/usr/bin/env python3
"""Golden vectors for the ViT forward restatement (oracle/vit_oracle.py).

Build container only.  The reference's utils/embeddings.py cannot be imported offline (it fetches
checkpoints by name at import time), so the forward is pinned against the model classes
the reference instantiates -- transformers' Dinov2Model / ViTModel and (for open_clip's ViT-B-32
architecture) CLIPVisionModelWithProjection -- built from local configs with the seeded random
weights of ibloc_amd.vit.random_weights.  Only outputs are stored; weights and inputs are regenerated
from their seeds.  Output: tests/golden/vit_golden.npz
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ibloc_amd import vit as V  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "vit_golden.npz")


def t(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32))


def hf_dinov2(cfg, w):
    from transformers import Dinov2Config, Dinov2Model
    c = Dinov2Config(hidden_size=cfg.dim, num_hidden_layers=cfg.depth, num_attention_heads=cfg.heads,
                     mlp_ratio=cfg.mlp_dim // cfg.dim, image_size=cfg.pos_grid[0] * cfg.patch, patch_size=cfg.patch,
                     qkv_bias=True, layerscale_value=1.0, use_swiglu_ffn=False, layer_norm_eps=cfg.ln_eps,
                     hidden_act="gelu")
    m = Dinov2Model(c).eval()
    sd = {"embeddings.cls_token": t(w["cls"]).reshape(1, 1, -1),
          "embeddings.position_embeddings": t(w["pos"]).unsqueeze(0),
          "embeddings.patch_embeddings.projection.weight": t(w["patch.w"]),
          "embeddings.patch_embeddings.projection.bias": t(w["patch.b"]),
          "layernorm.weight": t(w["ln_f.g"]), "layernorm.bias": t(w["ln_f.b"])}
    for l in range(cfg.depth):
        p, q = f"encoder.layer.{l}.", f"l{l}."
        sd[p + "norm1.weight"], sd[p + "norm1.bias"] = t(w[q + "ln1.g"]), t(w[q + "ln1.b"])
        sd[p + "norm2.weight"], sd[p + "norm2.bias"] = t(w[q + "ln2.g"]), t(w[q + "ln2.b"])
        for hf, mine in (("query", "q"), ("key", "k"), ("value", "v")):
            sd[p + f"attention.attention.{hf}.weight"] = t(w[q + mine + ".w"])
            sd[p + f"attention.attention.{hf}.bias"] = t(w[q + mine + ".b"])
        sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"] = t(w[q + "o.w"]), t(w[q + "o.b"])
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = t(w[q + "fc1.w"]), t(w[q + "fc1.b"])
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = t(w[q + "fc2.w"]), t(w[q + "fc2.b"])
        sd[p + "layer_scale1.lambda1"], sd[p + "layer_scale2.lambda1"] = t(w[q + "ls1"]), t(w[q + "ls2"])
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("mask_token" in k for k in missing), (missing, unexpected)
    return lambda x: m(pixel_values=x).last_hidden_state[:, 0]


def hf_vit(cfg, w):
    from transformers import ViTConfig, ViTModel
    c = ViTConfig(hidden_size=cfg.dim, num_hidden_layers=cfg.depth, num_attention_heads=cfg.heads,
                  intermediate_size=cfg.mlp_dim, image_size=cfg.img_h, patch_size=cfg.patch, layer_norm_eps=cfg.ln_eps,
                  hidden_act="gelu", qkv_bias=True)
    m = ViTModel(c, add_pooling_layer=False).eval()
    sd = {"embeddings.cls_token": t(w["cls"]).reshape(1, 1, -1),
          "embeddings.position_embeddings": t(w["pos"]).unsqueeze(0),
          "embeddings.patch_embeddings.projection.weight": t(w["patch.w"]),
          "embeddings.patch_embeddings.projection.bias": t(w["patch.b"]),
          "layernorm.weight": t(w["ln_f.g"]), "layernorm.bias": t(w["ln_f.b"])}
    for l in range(cfg.depth):
        p, q = f"layers.{l}.", f"l{l}."     # transformers 5.x parameter names of ViTModel
        sd[p + "layernorm_before.weight"], sd[p + "layernorm_before.bias"] = t(w[q + "ln1.g"]), t(w[q + "ln1.b"])
        sd[p + "layernorm_after.weight"], sd[p + "layernorm_after.bias"] = t(w[q + "ln2.g"]), t(w[q + "ln2.b"])
        for hf, mine in (("q_proj", "q"), ("k_proj", "k"), ("v_proj", "v"), ("o_proj", "o")):
            sd[p + f"attention.{hf}.weight"], sd[p + f"attention.{hf}.bias"] = t(w[q + mine + ".w"]), t(w[q + mine + ".b"])
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = t(w[q + "fc1.w"]), t(w[q + "fc1.b"])
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = t(w[q + "fc2.w"]), t(w[q + "fc2.b"])
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return lambda x: m(pixel_values=x).last_hidden_state[:, 0, :]


def hf_clip(cfg, w):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    c = CLIPVisionConfig(hidden_size=cfg.dim, num_hidden_layers=cfg.depth, num_attention_heads=cfg.heads,
                         intermediate_size=cfg.mlp_dim, image_size=cfg.img_h, patch_size=cfg.patch,
                         layer_norm_eps=cfg.ln_eps, hidden_act="gelu", projection_dim=cfg.proj_dim)
    m = CLIPVisionModelWithProjection(c).eval()
    sd = {"vision_model.embeddings.class_embedding": t(w["cls"]),
          "vision_model.embeddings.position_embedding.weight": t(w["pos"]),
          "vision_model.embeddings.patch_embedding.weight": t(w["patch.w"]),
          "vision_model.pre_layrnorm.weight": t(w["ln_pre.g"]), "vision_model.pre_layrnorm.bias": t(w["ln_pre.b"]),
          "vision_model.post_layernorm.weight": t(w["ln_f.g"]), "vision_model.post_layernorm.bias": t(w["ln_f.b"]),
          "visual_projection.weight": t(w["proj.w"])}
    for l in range(cfg.depth):
        p, q = f"vision_model.encoder.layers.{l}.", f"l{l}."
        sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"] = t(w[q + "ln1.g"]), t(w[q + "ln1.b"])
        sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"] = t(w[q + "ln2.g"]), t(w[q + "ln2.b"])
        for hf, mine in (("q_proj", "q"), ("k_proj", "k"), ("v_proj", "v"), ("out_proj", "o")):
            sd[p + f"self_attn.{hf}.weight"], sd[p + f"self_attn.{hf}.bias"] = t(w[q + mine + ".w"]), t(w[q + mine + ".b"])
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = t(w[q + "fc1.w"]), t(w[q + "fc1.b"])
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = t(w[q + "fc2.w"]), t(w[q + "fc2.b"])
    missing, unexpected = m.load_state_dict(sd, strict=False)
    missing = [k for k in missing if "position_ids" not in k]
    assert not unexpected and not missing, (missing, unexpected)
    return lambda x: m(pixel_values=x).image_embeds


CASES = [
    # (golden key, config name, overrides, hf builder, weight seed, input seed, batch)
    ("tiny_dino", "tiny_dino", {"pos_interp": "size"}, hf_dinov2, 101, 201, 3),
    ("dino_b14_2layer", "dinov2_vitb14", {"pos_interp": "size", "depth": 2}, hf_dinov2, 102, 202, 2),
    ("dino_s14_full", "dinov2_vits14", {"pos_interp": "size"}, hf_dinov2, 103, 203, 2),
    ("dino_b14_full", "dinov2_vitb14", {"pos_interp": "size"}, hf_dinov2, 104, 204, 2),
    ("tiny_vit16", "vit_b16", {"dim": 128, "depth": 2, "heads": 2, "mlp_dim": 256}, hf_vit, 105, 205, 3),
    ("tiny_clip", "tiny_clip", {"patch_bias": False}, hf_clip, 106, 206, 3),
    # full-size a2 / a3 configurations (utils/embeddings.py:74-98 google/vit-base-patch16-224-in21k, :31-50 open_clip ViT-B-32)
    ("vit_b16_full", "vit_b16", {}, hf_vit, 107, 207, 2),
    ("clip_b32_full", "clip_b32", {"patch_bias": False}, hf_clip, 108, 208, 2),
]


def make_cfg(name, over):
    import dataclasses
    over = {k: v for k, v in over.items() if k != "patch_bias"}
    return dataclasses.replace(V.CONFIGS[name], **over)


def main():
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    out = {}
    with torch.no_grad():
        for key, name, over, builder, wseed, iseed, batch in CASES:
            cfg = make_cfg(name, over)
            w = V.random_weights(cfg, wseed)
            if over.get("patch_bias") is False:
                w["patch.b"] = np.zeros_like(w["patch.b"])
            x = np.random.default_rng(iseed).normal(size=(batch, 3, cfg.img_h, cfg.img_w)).astype(np.float32)
            y = builder(cfg, w)(torch.from_numpy(x)).numpy()
            out[key] = y.astype(np.float32)
            print(key, y.shape, float(np.abs(y).mean()))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()

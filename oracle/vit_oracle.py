"""ORACLE (test infrastructure; never imported by the product path).

CPU restatement (torch fp32 on the host) of the embedding path of the reference:
  * utils/embeddings.py:53-71  get_all_dino_embeddings: cv2 BGR2RGB swap, HF BitImageProcessor
    (shortest edge 256 bicubic, centre crop 224, /255, ImageNet mean/std), Dinov2Model,
    last_hidden_state[:, 0] (CLS after the final LayerNorm), un-normalised.
  * utils/embeddings.py:74-98  get_all_vit_embeddings (ViTModel, 224x224 bilinear, mean=std=0.5).
  * utils/embeddings.py:31-50  get_all_clip_embeddings (open_clip ViT-B-32: ln_pre, ln_post, proj,
    L2-normalised output).
The resize goes through PIL itself -- the library the reference's processors call -- so the
preprocessing oracle is the reference's own dependency, not a re-implementation.  The transformer
forward is pinned against transformers' Dinov2Model / ViTModel / CLIPVisionModelWithProjection with
seeded random weights (tests/golden/vit_golden.npz, generator tools/gen_golden_vit.py); pretrained
checkpoints are not available offline, so pretrained-weight parity is unpinned.
"""
import math

import numpy as np
import torch
from PIL import Image

_PIL_FILTER = {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR}


def preprocess_crop(crop: np.ndarray, recipe) -> np.ndarray:
    """HxWx3 uint8 (as handed to the embedding function) -> (3, out_h, out_w) float32 model input.
    `recipe` is an ibloc_amd.preprocess.PreprocessRecipe (plain data: sizes, filter, mean/std)."""
    u8 = preprocess_crop_u8(crop, recipe)
    x = (u8.astype(np.float64) * (1 / 255)).astype(np.float32)
    x = (x - np.asarray(recipe.mean, dtype=np.float32)) / np.asarray(recipe.std, dtype=np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def preprocess_crop_u8(crop: np.ndarray, recipe) -> np.ndarray:
    img = np.ascontiguousarray(crop, dtype=np.uint8)
    if recipe.swap_rb:
        img = img[:, :, ::-1]            # cv2.cvtColor(img, cv2.COLOR_BGR2RGB), embeddings.py:41,64,86
    h, w = img.shape[:2]
    if recipe.resize_mode == "exact":
        rh, rw = recipe.out_h, recipe.out_w
    else:
        short, long = (h, w) if h <= w else (w, h)
        ns, nl = recipe.shortest, int(recipe.shortest * long / short)
        rh, rw = (ns, nl) if h <= w else (nl, ns)
    res = np.asarray(Image.fromarray(np.ascontiguousarray(img)).resize((rw, rh), resample=_PIL_FILTER[recipe.filt]))
    if recipe.crop_rounding == "floor":
        top, left = (rh - recipe.out_h) // 2, (rw - recipe.out_w) // 2
    else:
        top, left = int(round((rh - recipe.out_h) / 2.0)), int(round((rw - recipe.out_w) / 2.0))
    return res[top:top + recipe.out_h, left:left + recipe.out_w]


def interpolate_pos(pos: torch.Tensor, stored_grid, grid, mode: str) -> torch.Tensor:
    gh, gw = stored_grid
    th, tw = grid
    if (gh, gw) == (th, tw):
        return pos
    t = pos[1:].reshape(1, gh, gw, -1).permute(0, 3, 1, 2)
    if mode == "hf-4.44":      # transformers 4.44.0 modeling_dinov2.py interpolate_pos_encoding
        sf = (float((th + 0.1) / math.sqrt(gh * gw)), float((tw + 0.1) / math.sqrt(gh * gw)))
        t = torch.nn.functional.interpolate(t, scale_factor=sf, mode="bicubic", align_corners=False)
    else:
        t = torch.nn.functional.interpolate(t, size=(th, tw), mode="bicubic", align_corners=False)
    return torch.cat([pos[:1], t.permute(0, 2, 3, 1).reshape(th * tw, -1)], dim=0)


@torch.no_grad()
def vit_forward(weights: dict, cfg, pixel_values: np.ndarray, all_tokens=False, device="cpu") -> np.ndarray:
    """pixel_values (B, 3, H, W) float32 -> (B, out_dim) float32 (or (B, T, D) when all_tokens).
    cfg is an ibloc_amd.vit.VitConfig (plain data).  device: where torch evaluates this fp32 restatement ("cuda" lets the parity
    tests embed tens of thousands of crops; gfx950 has no reduced-precision fp32 matmul mode, the arithmetic stays fp32)."""
    F = torch.nn.functional
    w = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v, dtype=np.float32))).to(device) for k, v in weights.items()}
    if isinstance(pixel_values, torch.Tensor):       # already a float32 tensor (the parity tests normalise u8 crops on the device)
        x = pixel_values.to(device=device, dtype=torch.float32)
    else:
        x = torch.from_numpy(np.asarray(pixel_values, dtype=np.float32)).to(device)
    B = x.shape[0]
    x = F.conv2d(x, w["patch.w"], w.get("patch.b"), stride=cfg.patch)           # (B, D, gh, gw)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([w["cls"].reshape(1, 1, -1).expand(B, -1, -1), x], dim=1)
    x = x + interpolate_pos(w["pos"], cfg.pos_grid, cfg.grid, cfg.pos_interp).unsqueeze(0)
    if cfg.pre_ln:
        x = F.layer_norm(x, (cfg.dim,), w["ln_pre.g"], w["ln_pre.b"], cfg.ln_eps)
    hd = cfg.dim // cfg.heads
    nrun = cfg.depth if cfg.n_blocks_run < 0 else cfg.n_blocks_run
    for l in range(nrun):
        p = f"l{l}."
        h = F.layer_norm(x, (cfg.dim,), w[p + "ln1.g"], w[p + "ln1.b"], cfg.ln_eps)
        q = F.linear(h, w[p + "q.w"], w[p + "q.b"]).view(B, -1, cfg.heads, hd).transpose(1, 2)
        k = F.linear(h, w[p + "k.w"], w[p + "k.b"]).view(B, -1, cfg.heads, hd).transpose(1, 2)
        v = F.linear(h, w[p + "v.w"], w[p + "v.b"]).view(B, -1, cfg.heads, hd).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v
        a = F.linear(a.transpose(1, 2).reshape(B, -1, cfg.dim), w[p + "o.w"], w[p + "o.b"])
        if cfg.layerscale:
            a = a * w[p + "ls1"]
        x = x + a
        h = F.layer_norm(x, (cfg.dim,), w[p + "ln2.g"], w[p + "ln2.b"], cfg.ln_eps)
        h = F.linear(F.gelu(F.linear(h, w[p + "fc1.w"], w[p + "fc1.b"])), w[p + "fc2.w"], w[p + "fc2.b"])
        if cfg.layerscale:
            h = h * w[p + "ls2"]
        x = x + h
    if all_tokens:
        if cfg.final_ln:
            x = F.layer_norm(x, (cfg.dim,), w["ln_f.g"], w["ln_f.b"], cfg.ln_eps)
        return x.cpu().numpy()
    c = x[:, 0]
    if cfg.final_ln:
        c = F.layer_norm(c, (cfg.dim,), w["ln_f.g"], w["ln_f.b"], cfg.ln_eps)
    if cfg.proj_dim:
        c = c @ w["proj.w"].t()
    return c.cpu().numpy()


def embed_crops(weights: dict, cfg, recipe, crops, l2_normalize=False, device="cpu") -> np.ndarray:
    px = np.stack([preprocess_crop(c, recipe) for c in crops])
    out = vit_forward(weights, cfg, px, device=device)
    if l2_normalize:                       # clip_features /= clip_features.norm(...), embeddings.py:48
        out = out / np.linalg.norm(out, axis=-1, keepdims=True)
    return out

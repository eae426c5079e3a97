"""Drop-in for the reference's `utils/embeddings.py` on the MI355X build.

The reference loads four encoders at import time by fetching checkpoints by name
(/root/reference/utils/embeddings.py:13-28, 101-103) and embeds ONE crop per call.  Here the encoders are
`ibloc_amd.vit.VitEncoder` objects (HIP kernels behind the C-ABI) registered once with `set_encoder(kind, encoder)`
or built from a converted checkpoint with `load_encoder(kind, state_dict)` (kinds "dino" | "vit" | "clip" | "dator"); the four reference entry points keep
their names and `(**kwargs) -> torch.Tensor` signature:

    get_all_clip_embeddings   :31-50   L2-normalised 512-d
    get_all_dino_embeddings   :53-71   CLS after the final LayerNorm, 768-d, un-normalised
    get_all_vit_embeddings    :74-98   CLS, 768-d
    get_dator_embeddings      :105-121 RGB-D 128-d (ibloc_amd.dator.DatorEncoder registered as kind "dator")

`embed_batch(kind, crops)` is the batched form the localisation engine uses (one launch sequence for many crops).
"""
import numpy as np
import torch

from ibloc_amd import match
from ibloc_amd import vit as V

_ENCODERS = {}
_KIND_TO_CONFIG = {"dino": "dinov2_vitb14", "vit": "vit_b16", "clip": "clip_b32"}


def set_encoder(kind: str, encoder: V.VitEncoder) -> None:
    _ENCODERS[kind] = encoder


def hf_dinov2_to_weights(sd: dict, depth: int) -> dict:
    """transformers Dinov2Model.state_dict() (4.44 naming) -> the weight dict VitEncoder takes."""
    g = lambda k: np.asarray(sd[k].float().cpu() if hasattr(sd[k], "float") else sd[k], dtype=np.float32)
    w = {"patch.w": g("embeddings.patch_embeddings.projection.weight"), "patch.b": g("embeddings.patch_embeddings.projection.bias"),
         "cls": g("embeddings.cls_token").reshape(-1), "pos": g("embeddings.position_embeddings")[0],
         "ln_f.g": g("layernorm.weight"), "ln_f.b": g("layernorm.bias")}
    for l in range(depth):
        p, q = f"encoder.layer.{l}.", f"l{l}."
        w[q + "ln1.g"], w[q + "ln1.b"] = g(p + "norm1.weight"), g(p + "norm1.bias")
        w[q + "ln2.g"], w[q + "ln2.b"] = g(p + "norm2.weight"), g(p + "norm2.bias")
        for hf, mine in (("query", "q"), ("key", "k"), ("value", "v")):
            w[q + mine + ".w"], w[q + mine + ".b"] = g(p + f"attention.attention.{hf}.weight"), g(p + f"attention.attention.{hf}.bias")
        w[q + "o.w"], w[q + "o.b"] = g(p + "attention.output.dense.weight"), g(p + "attention.output.dense.bias")
        w[q + "fc1.w"], w[q + "fc1.b"] = g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias")
        w[q + "fc2.w"], w[q + "fc2.b"] = g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias")
        w[q + "ls1"], w[q + "ls2"] = g(p + "layer_scale1.lambda1"), g(p + "layer_scale2.lambda1")
    return w


def hf_vit_to_weights(sd: dict, depth: int) -> dict:
    """transformers ViTModel.state_dict() -> weight dict.  Both parameter namings are read: transformers 4.44 (the version the reference
    pins, environment.yml:275: `encoder.layer.N.attention.attention.query`, `intermediate.dense`, `output.dense`) and 5.x
    (`layers.N.attention.q_proj`, `mlp.fc1`, `mlp.fc2`)."""
    g = lambda k: np.asarray(sd[k].float().cpu() if hasattr(sd[k], "float") else sd[k], dtype=np.float32)
    w = {"patch.w": g("embeddings.patch_embeddings.projection.weight"), "patch.b": g("embeddings.patch_embeddings.projection.bias"),
         "cls": g("embeddings.cls_token").reshape(-1), "pos": g("embeddings.position_embeddings")[0],
         "ln_f.g": g("layernorm.weight"), "ln_f.b": g("layernorm.bias")}
    new_names = "layers.0.attention.q_proj.weight" in sd
    for l in range(depth):
        p, q = (f"layers.{l}." if new_names else f"encoder.layer.{l}."), f"l{l}."
        w[q + "ln1.g"], w[q + "ln1.b"] = g(p + "layernorm_before.weight"), g(p + "layernorm_before.bias")
        w[q + "ln2.g"], w[q + "ln2.b"] = g(p + "layernorm_after.weight"), g(p + "layernorm_after.bias")
        if new_names:
            for hf, mine in (("q_proj", "q"), ("k_proj", "k"), ("v_proj", "v"), ("o_proj", "o")):
                w[q + mine + ".w"], w[q + mine + ".b"] = g(p + f"attention.{hf}.weight"), g(p + f"attention.{hf}.bias")
            w[q + "fc1.w"], w[q + "fc1.b"] = g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias")
            w[q + "fc2.w"], w[q + "fc2.b"] = g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias")
        else:
            for hf, mine in (("query", "q"), ("key", "k"), ("value", "v")):
                w[q + mine + ".w"], w[q + mine + ".b"] = g(p + f"attention.attention.{hf}.weight"), g(p + f"attention.attention.{hf}.bias")
            w[q + "o.w"], w[q + "o.b"] = g(p + "attention.output.dense.weight"), g(p + "attention.output.dense.bias")
            w[q + "fc1.w"], w[q + "fc1.b"] = g(p + "intermediate.dense.weight"), g(p + "intermediate.dense.bias")
            w[q + "fc2.w"], w[q + "fc2.b"] = g(p + "output.dense.weight"), g(p + "output.dense.bias")
    return w


def open_clip_visual_to_weights(sd: dict, depth: int) -> dict:
    """open_clip `model.visual.state_dict()` (VisionTransformer) -> weight dict (fused in_proj split into q/k/v)."""
    g = lambda k: np.asarray(sd[k].float().cpu() if hasattr(sd[k], "float") else sd[k], dtype=np.float32)
    D = g("class_embedding").shape[0]
    w = {"patch.w": g("conv1.weight"), "patch.b": np.zeros(D, np.float32), "cls": g("class_embedding"),
         "pos": g("positional_embedding"), "ln_pre.g": g("ln_pre.weight"), "ln_pre.b": g("ln_pre.bias"),
         "ln_f.g": g("ln_post.weight"), "ln_f.b": g("ln_post.bias"), "proj.w": g("proj").T.copy()}
    for l in range(depth):
        p, q = f"transformer.resblocks.{l}.", f"l{l}."
        w[q + "ln1.g"], w[q + "ln1.b"] = g(p + "ln_1.weight"), g(p + "ln_1.bias")
        w[q + "ln2.g"], w[q + "ln2.b"] = g(p + "ln_2.weight"), g(p + "ln_2.bias")
        wi, bi = g(p + "attn.in_proj_weight"), g(p + "attn.in_proj_bias")
        for i, mine in enumerate("qkv"):
            w[q + mine + ".w"], w[q + mine + ".b"] = wi[i * D:(i + 1) * D], bi[i * D:(i + 1) * D]
        w[q + "o.w"], w[q + "o.b"] = g(p + "attn.out_proj.weight"), g(p + "attn.out_proj.bias")
        w[q + "fc1.w"], w[q + "fc1.b"] = g(p + "mlp.c_fc.weight"), g(p + "mlp.c_fc.bias")
        w[q + "fc2.w"], w[q + "fc2.b"] = g(p + "mlp.c_proj.weight"), g(p + "mlp.c_proj.bias")
    return w


def load_encoder(kind: str, state_dict, device="cuda", cfg: V.VitConfig = None):
    """Build + register the encoder of `kind` ("dino" | "vit" | "clip" | "dator") from a checkpoint state dict.  cfg: another
    architecture / position-embedding rule than the checkpoint the reference loads (default: V.CONFIGS of the kind).

    "dator": `state_dict` is a `build_FourDNet` checkpoint as the reference's `load_model('.../dator_best_tum.pth')` reads it
    (utils/embeddings.py:101-103, make_model.py:620-626) -- the dict itself or the path of the `.pth` file (loaded with
    `weights_only=True`: a checkpoint is data) -> `ibloc_amd.dator.DatorEncoder` (`module.` prefix stripped, `classifier*` skipped, LoRA
    folded into the QKV weights)."""
    if kind == "dator":
        from ibloc_amd import dator as D
        if isinstance(state_dict, (str, bytes)) or hasattr(state_dict, "__fspath__"):
            state_dict = torch.load(state_dict, map_location="cpu", weights_only=True)
        rw, dw, hw = D.fourdnet_state_dict_to_weights(state_dict)
        enc = D.DatorEncoder(rw, dw, hw, device=device)
        set_encoder(kind, enc)
        return enc
    cfg = cfg or V.CONFIGS[_KIND_TO_CONFIG[kind]]
    conv = {"dino": hf_dinov2_to_weights, "vit": hf_vit_to_weights, "clip": open_clip_visual_to_weights}[kind]
    enc = V.VitEncoder(cfg, conv(state_dict, cfg.depth), device=device)
    set_encoder(kind, enc)
    return enc


def _encoder(kind):
    if kind not in _ENCODERS:
        raise RuntimeError(
            f"no '{kind}' encoder registered: call utils.embeddings.load_encoder('{kind}', state_dict) with the checkpoint the "
            "reference loads at import time (utils/embeddings.py:13-28), or set_encoder(kind, VitEncoder) -- checkpoints cannot be "
            "fetched by name on an offline MI355X box")
    return _ENCODERS[kind]


def embed_batch(kind: str, crops) -> torch.Tensor:
    """crops: list of HxWx3 uint8 arrays (or a uint8 tensor N x H x W x 3) -> (N, D) device tensor."""
    out = _encoder(kind).embed(crops)
    if kind == "clip":
        out = match.normalize_rows(out)          # clip_features /= clip_features.norm(dim=-1, keepdim=True), :48
    return out


def get_all_clip_embeddings(**kwargs) -> torch.Tensor:
    return embed_batch("clip", [kwargs["current_obj_grounded_img"]])[0]


def get_all_dino_embeddings(**kwargs) -> torch.Tensor:
    return embed_batch("dino", [kwargs["current_obj_grounded_img"]])[0]


def get_all_vit_embeddings(**kwargs) -> torch.Tensor:
    return embed_batch("vit", [kwargs["current_obj_grounded_img"]])[0]


def get_dator_embeddings(**kwargs) -> torch.Tensor:
    """utils/embeddings.py:105-121: crop the full depth image with the bounding box (xyxy), embed (RGB crop, depth crop)."""
    enc = _encoder("dator")
    bb = kwargs["current_obj_bounding_box"]
    depth = kwargs["full_depth_image"][int(bb[1]):int(bb[3]), int(bb[0]):int(bb[2])]
    return enc.embed([kwargs["current_obj_grounded_img"]], [np.asarray(depth, dtype=np.float32)])[0]

"""The checkpoint converters of utils/embeddings.py -- the only way real weights enter the HIP encoders (they replace the import-time
`from_pretrained` loaders of /root/reference/utils/embeddings.py:13-28).  Models of the classes the reference instantiates are built
from local configs with seeded random weights (no checkpoint exists offline), their state_dict goes through the converter, and the
converted weights must reproduce the model's own forward through the oracle (CPU, fp32).  open_clip is not installed: its
VisionTransformer state dict is assembled from the architecturally identical transformers CLIP vision tower (open_clip's own
parameter names; fused in_proj, `proj` stored (width, out))."""
import dataclasses

import numpy as np
import pytest
import torch

from ibloc_amd import vit as V
from ibloc_amd.utils import embeddings as E
from oracle import vit_oracle as vo


def _close(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def _perturb(model, seed):
    """default inits leave LayerNorm / LayerScale / biases at constants: draw everything, so that a swapped pair cannot hide"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.05)


def test_hf_dinov2_converter():
    from transformers import Dinov2Config, Dinov2Model
    cfg = dataclasses.replace(V.CONFIGS["tiny_dino"], pos_interp="size")      # the installed transformers resamples with size=
    torch.manual_seed(0)
    m = Dinov2Model(Dinov2Config(hidden_size=cfg.dim, num_hidden_layers=cfg.depth, num_attention_heads=cfg.heads, mlp_ratio=cfg.mlp_dim // cfg.dim,
                                 image_size=cfg.pos_grid[0] * cfg.patch, patch_size=cfg.patch, qkv_bias=True, layerscale_value=1.0,
                                 use_swiglu_ffn=False, layer_norm_eps=cfg.ln_eps, hidden_act="gelu")).eval()
    _perturb(m, 1)
    x = np.random.default_rng(2).normal(size=(2, 3, cfg.img_h, cfg.img_w)).astype(np.float32)
    with torch.no_grad():
        want = m(pixel_values=torch.from_numpy(x)).last_hidden_state[:, 0].numpy()
    got = vo.vit_forward(E.hf_dinov2_to_weights(m.state_dict(), cfg.depth), cfg, x)
    assert _close(got, want) < 2e-5


def _vit_444_names(sd):
    """transformers 5.x ViTModel names -> the 4.44 names the reference's checkpoints carry"""
    out = {}
    for k, v in sd.items():
        k2 = k.replace("layers.", "encoder.layer.")
        for a, b in (("attention.q_proj", "attention.attention.query"), ("attention.k_proj", "attention.attention.key"),
                     ("attention.v_proj", "attention.attention.value"), ("attention.o_proj", "attention.output.dense"),
                     ("mlp.fc1", "intermediate.dense"), ("mlp.fc2", "output.dense")):
            k2 = k2.replace(a, b)
        out[k2] = v
    return out


@pytest.mark.parametrize("naming", ["5.x", "4.44"])
def test_hf_vit_converter(naming):
    from transformers import ViTConfig, ViTModel
    cfg = dataclasses.replace(V.CONFIGS["vit_b16"], dim=128, depth=2, heads=2, mlp_dim=256)
    torch.manual_seed(3)
    m = ViTModel(ViTConfig(hidden_size=cfg.dim, num_hidden_layers=cfg.depth, num_attention_heads=cfg.heads, intermediate_size=cfg.mlp_dim,
                           image_size=cfg.img_h, patch_size=cfg.patch, layer_norm_eps=cfg.ln_eps, hidden_act="gelu", qkv_bias=True),
                 add_pooling_layer=False).eval()
    _perturb(m, 4)
    x = np.random.default_rng(5).normal(size=(2, 3, cfg.img_h, cfg.img_w)).astype(np.float32)
    with torch.no_grad():
        want = m(pixel_values=torch.from_numpy(x)).last_hidden_state[:, 0, :].numpy()
    sd = m.state_dict()
    if naming == "4.44":
        sd = _vit_444_names(sd)
        assert "encoder.layer.0.attention.attention.query.weight" in sd and "encoder.layer.1.intermediate.dense.bias" in sd
    got = vo.vit_forward(E.hf_vit_to_weights(sd, cfg.depth), cfg, x)
    assert _close(got, want) < 2e-5


def _open_clip_visual_state_dict(hf_sd, depth):
    """open_clip VisionTransformer parameter names from the transformers CLIP vision tower (same architecture)"""
    v = "vision_model."
    sd = {"conv1.weight": hf_sd[v + "embeddings.patch_embedding.weight"], "class_embedding": hf_sd[v + "embeddings.class_embedding"],
          "positional_embedding": hf_sd[v + "embeddings.position_embedding.weight"],
          "ln_pre.weight": hf_sd[v + "pre_layrnorm.weight"], "ln_pre.bias": hf_sd[v + "pre_layrnorm.bias"],
          "ln_post.weight": hf_sd[v + "post_layernorm.weight"], "ln_post.bias": hf_sd[v + "post_layernorm.bias"],
          "proj": hf_sd["visual_projection.weight"].T.contiguous()}
    for l in range(depth):
        p, q = v + f"encoder.layers.{l}.", f"transformer.resblocks.{l}."
        sd[q + "ln_1.weight"], sd[q + "ln_1.bias"] = hf_sd[p + "layer_norm1.weight"], hf_sd[p + "layer_norm1.bias"]
        sd[q + "ln_2.weight"], sd[q + "ln_2.bias"] = hf_sd[p + "layer_norm2.weight"], hf_sd[p + "layer_norm2.bias"]
        sd[q + "attn.in_proj_weight"] = torch.cat([hf_sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"])
        sd[q + "attn.in_proj_bias"] = torch.cat([hf_sd[p + f"self_attn.{n}_proj.bias"] for n in "qkv"])
        sd[q + "attn.out_proj.weight"], sd[q + "attn.out_proj.bias"] = hf_sd[p + "self_attn.out_proj.weight"], hf_sd[p + "self_attn.out_proj.bias"]
        sd[q + "mlp.c_fc.weight"], sd[q + "mlp.c_fc.bias"] = hf_sd[p + "mlp.fc1.weight"], hf_sd[p + "mlp.fc1.bias"]
        sd[q + "mlp.c_proj.weight"], sd[q + "mlp.c_proj.bias"] = hf_sd[p + "mlp.fc2.weight"], hf_sd[p + "mlp.fc2.bias"]
    return sd


def test_open_clip_visual_converter():
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    cfg = V.CONFIGS["tiny_clip"]
    torch.manual_seed(6)
    m = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=cfg.dim, num_hidden_layers=cfg.depth, num_attention_heads=cfg.heads,
                                                       intermediate_size=cfg.mlp_dim, image_size=cfg.img_h, patch_size=cfg.patch,
                                                       layer_norm_eps=cfg.ln_eps, hidden_act="gelu", projection_dim=cfg.proj_dim)).eval()
    _perturb(m, 7)
    x = np.random.default_rng(8).normal(size=(2, 3, cfg.img_h, cfg.img_w)).astype(np.float32)
    with torch.no_grad():
        want = m(pixel_values=torch.from_numpy(x)).image_embeds.numpy()
    got = vo.vit_forward(E.open_clip_visual_to_weights(_open_clip_visual_state_dict(m.state_dict(), cfg.depth), cfg.depth), cfg, x)
    assert _close(got, want) < 2e-5


def _bicubic_scale_factor_np(t, sf_h, sf_w):
    """torch.nn.functional.interpolate(t, scale_factor=(sf_h, sf_w), mode="bicubic", align_corners=False) restated with numpy:
    output size floor(in * scale), source coordinate (dst + 0.5) / scale - 0.5 (the given scale, not out / in), cubic convolution
    with A = -0.75 over the four taps floor(src) - 1 .. floor(src) + 2, indices clamped to the border."""
    def axis(n, sf):
        m = int(np.floor(n * sf))
        src = (np.arange(m) + 0.5) / sf - 0.5
        i0 = np.floor(src).astype(np.int64)
        f = src - i0
        A = -0.75
        def c1(x): return ((A + 2) * x - (A + 3)) * x * x + 1            # |x| <= 1
        def c2(x): return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A      # 1 < |x| < 2
        wts = np.stack([c2(f + 1), c1(f), c1(1 - f), c2(2 - f)], 1)
        idx = np.clip(i0[:, None] + np.arange(-1, 3)[None], 0, n - 1)
        return idx, wts
    iy, wy = axis(t.shape[0], sf_h)
    ix, wx = axis(t.shape[1], sf_w)
    rows = np.einsum("ok,okxc->oxc", wy, t[iy])                          # (out_h, in_w, C)
    return np.einsum("pk,opkc->opc", wx, rows[:, ix])


def test_hf_444_position_embedding_rule():
    """the shipped default `pos_interp="hf-4.44"`: transformers 4.44.0 Dinov2Embeddings.interpolate_pos_encoding resamples the 37 x 37
    table with scale_factor = (16 + 0.1) / 37, not size=(16, 16) -- the two differ in the sampling grid.  Checked against an
    independent numpy restatement of torch's bicubic kernel."""
    cfg = V.CONFIGS["dinov2_vitb14"]
    rng = np.random.default_rng(9)
    pos = rng.normal(size=(1 + 37 * 37, 24)).astype(np.float32)
    small = dataclasses.replace(cfg, dim=24)
    got = V.interpolate_pos_embed(pos, small)
    sf = (16 + 0.1) / 37.0
    want = _bicubic_scale_factor_np(pos[1:].reshape(37, 37, 24).astype(np.float64), sf, sf).reshape(256, 24)
    assert got.shape == (257, 24) and np.array_equal(got[0], pos[0])
    assert np.abs(got[1:] - want).max() < 2e-5
    other = V.interpolate_pos_embed(pos, dataclasses.replace(small, pos_interp="size"))
    assert np.abs(other[1:] - want).max() > 1e-3            # the size= rule is a different resampling


def _fourdnet_checkpoint(rw, dw, hw, prefix="module."):
    """A checkpoint with EXACTLY the entries of the reference model's own state_dict() (names and shapes recorded from the live
    `build_FourDNet` in the build container: tests/golden/dator_state_keys.json, tools/gen_golden_dator.py), filled with the given weights
    through the reference's parameter names; entries the forward never reads (classifier, the streams' norm / fc) get noise."""
    import importlib.util
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("gen_golden_dator", os.path.join(here, "..", "tools", "gen_golden_dator.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    keys = json.load(open(os.path.join(here, "golden", "dator_state_keys.json")))
    sd = {}
    gen.stream_state("base.", rw, sd)
    gen.stream_state("base2.", dw, sd)
    gen.head_state(hw, sd)
    assert set(sd) <= set(keys), sorted(set(sd) - set(keys))[:5]
    g = torch.Generator().manual_seed(11)
    for k, shape in keys.items():
        if k in sd:
            assert list(sd[k].shape) == shape, (k, list(sd[k].shape), shape)
        else:
            assert any(a in k for a in ("classifier", ".fc.", ".norm.")), k          # what load_param copies but forward() never reads
            sd[k] = torch.randn(shape, generator=g)
    return {prefix + k: v for k, v in sd.items()}


def test_fourdnet_checkpoint_converter():
    """load_encoder("dator", ...)'s converter on a checkpoint shaped like the reference's dator_best_tum.pth (make_model.py:620-626):
    `module.` prefix stripped, classifier skipped, LoRA factors kept for folding -- and the converted weights reproduce the embedding the
    reference's own build_FourDNet computed from the same parameters (tests/golden/dator_golden.npz)."""
    import os
    from ibloc_amd import dator as D
    from oracle import dator_oracle as do
    rw, dw, hw = D.random_stream_weights(301), D.random_stream_weights(302), D.random_head_weights(303)
    ck = _fourdnet_checkpoint(rw, dw, hw)
    rw2, dw2, hw2 = D.fourdnet_state_dict_to_weights(ck)
    for got, want in ((rw2, rw), (dw2, dw), (hw2, hw)):
        assert set(want) - set(got) <= {"ln_f.g", "ln_f.b"} and set(got) <= set(want)
        for k in got:
            assert np.array_equal(got[k], np.asarray(want[k], dtype=np.float32).reshape(got[k].shape)), k
    assert "l10.lora_down" in rw2 and "l11.lora_up" in dw2 and "l9.lora_down" not in rw2
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "dator_golden.npz"))
    rng = np.random.default_rng(304)
    rgb = rng.normal(size=(3, 3, 256, 128)).astype(np.float32)
    depth = np.repeat(rng.uniform(-1, 1, size=(3, 1, 256, 128)).astype(np.float32), 3, axis=1)
    emb = do.forward(rw2, dw2, hw2, D.STREAM_CFG, rgb[:1], depth[:1])
    assert np.abs(emb - gold["embedding"][:1]).max() < 2e-4 * max(1.0, np.abs(gold["embedding"]).max())
    # no prefix (a plain checkpoint) reads the same; a truncated checkpoint names the missing entry
    plain = {k[len("module."):]: v for k, v in ck.items()}
    assert np.array_equal(D.fourdnet_state_dict_to_weights(plain)[2]["Q_r.w"], hw2["Q_r.w"])
    del plain["base2.blocks.3.mlp.fc2.weight"]
    with pytest.raises(KeyError, match="base2.blocks.3.mlp.fc2.weight"):
        D.fourdnet_state_dict_to_weights(plain)

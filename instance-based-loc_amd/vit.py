"""Host side of the ViT encoders (DINOv2 / ViT-B/16 / CLIP ViT-B/32 / TransReID streams):
configuration, weight packing into the C-ABI structs of include/ibloc.h, and the batched
preprocess + forward call.  Replaces the model objects utils/embeddings.py builds at import time
(/root/reference/utils/embeddings.py:13-28) and the per-crop calls at :31-98.
"""
import ctypes as C
import math
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from . import preprocess as pp

FLAG_LAYERSCALE = 1
FLAG_PRE_LN = 2
FLAG_FINAL_LN = 4
FLAG_QUICK_GELU = 8
FLAG_PROJ = 16
FLAG_OUT_ALL_TOKENS = 32
FLAG_ACT_TERMS2 = 64         # IBL_VIT_ACT_TERMS2 / 3: some block takes the input of a residual GEMM as K-extended rows of 2 / 3 terms
FLAG_ACT_TERMS3 = 128
MAX_LAYERS = 32


class VitDesc(C.Structure):
    _fields_ = [("dim", C.c_int32), ("depth", C.c_int32), ("heads", C.c_int32), ("mlp_dim", C.c_int32),
                ("patch", C.c_int32), ("img_h", C.c_int32), ("img_w", C.c_int32), ("n_tokens", C.c_int32),
                ("patch_k_pad", C.c_int32), ("flags", C.c_int32), ("n_blocks_run", C.c_int32),
                ("out_dim", C.c_int32), ("ln_eps", C.c_float)]


class VitLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1_g", "ln1_b", "w_qkv", "b_qkv", "w_o", "b_o", "ls1", "ln2_g", "ln2_b",
                                          "w_fc1", "b_fc1", "w_fc2", "b_fc2", "ls2",
                                          "w_qkv_x", "w_o_lo", "ls1_lo", "w_fc1_x", "w_fc2_lo", "ls2_lo")] + \
               [("qkv_terms", C.c_int32), ("fc1_terms", C.c_int32), ("o_terms", C.c_int32), ("fc2_terms", C.c_int32),
                ("w_o_x", C.c_void_p), ("w_fc2_x", C.c_void_p)]


class VitWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_patch", "b_patch", "cls_pos", "pos_patch", "ln_pre_g", "ln_pre_b",
                                          "ln_f_g", "ln_f_b", "w_proj", "w_patch_lo", "w_proj_x")] + [("layers", VitLayer * MAX_LAYERS)]


SPLIT_SCALE = 64.0          # IBL_VIT_SPLIT_SCALE (include/ibloc.h)

# Which operands of which blocks get a second fp16 term ("p<terms>;<layer>:<qkv><o><fc1><fc2>;...", layer "*" = every other block; qkv / fc1: 1, 2 (weights) or 3
# (weights + LayerNorm output), o / fc2: 1, 2 (weights) or 3 (weights + the attention output / the GELU hidden layer in two terms: one more
# accumulating launch each, round 4), p(atch embedding): 1 or 2 (weights)).  Measured on the 12-layer ViT-B/14 with the seeded
# random-init weights (DESIGN (c), tools/sim_vit_rounding.py): the residual stream is small in the first blocks, so the patch
# embedding (18 % of the error variance at 1.3 % of the FLOPs), block 0 (46 %) and block 1 (14 %) carry most of the fp16 rounding
# error of the embedding; the default gives them exact weights where that is cheap.  "plain" = one term everywhere (rounds 1-2).
# Round 4 (3 584 crops, forward of 224 crops): round 3's "p2;0:3222;1:2211" 7.60e-4 mean / 8.97e-4 max, 15.38 ms; this default (block 0's second
# LayerNorm output and block 2's QKV weights in two terms as well -- both ride in K-extended launches, no extra launch) 7.26e-4 / 8.59e-4,
# 15.76 ms; "p2;0:3232;1:2222" 7.14e-4 / 8.49e-4 but 16.14 ms (the second-term launch of an fc2 costs a whole read-modify-write pass).
DEFAULT_PRECISION = "p2;0:3232;1:2211;2:2111"
# per-model defaults where they differ: CLIP ViT-B/32 runs 50 tokens per crop (4 ms per 224 crops), so two-term weights in every block cost
# 2 ms and take its worst crop of 896 from 9.7e-4 -- too close to SURVEY 8d's 1e-3 gate -- to 8.1e-4 (mean 7.1e-4 -> 6.1e-4)
MODEL_PRECISION = {"clip_b32": "p2;*:2222;0:3232"}


def parse_precision(spec):
    """-> (patch_terms, {layer: (qkv, o, fc1, fc2)})"""
    if spec in (None, "", "default"):
        spec = DEFAULT_PRECISION
    if spec == "plain":
        return 1, {}
    patch, layers = 1, {}
    for part in spec.split(";"):
        part = part.strip()
        if not part:
            continue
        if part[0] == "p":
            patch = int(part[1:])
        else:
            l, t = part.split(":")
            t = tuple(int(c) for c in t)
            if len(t) != 4 or not all(1 <= v <= 3 for v in t):
                raise ValueError(f"bad precision entry {part!r}")
            layers["*" if l.strip() == "*" else int(l)] = t      # "*": every block without an entry of its own
    if patch not in (1, 2):
        raise ValueError("patch terms: 1 or 2")
    return patch, layers


def split_terms(w, terms):
    """fp32 weight (N, K) -> K-extended fp16 operand rows [W_hi | W_lo * S] (terms 2) or [W_hi | W_hi / S | W_lo * S] (terms 3)"""
    w = np.ascontiguousarray(w, dtype=np.float32)
    hi = w.astype(np.float16).astype(np.float32)
    lo = (w - hi) * SPLIT_SCALE
    parts = [hi, lo] if terms == 2 else [hi, hi / SPLIT_SCALE, lo]
    return np.concatenate(parts, axis=1)


def weight_lo(w):
    w = np.ascontiguousarray(w, dtype=np.float32)
    return (w - w.astype(np.float16).astype(np.float32)) * SPLIT_SCALE


@dataclass
class VitConfig:
    name: str
    dim: int
    depth: int
    heads: int
    mlp_dim: int
    patch: int
    img_h: int
    img_w: int
    pos_grid: tuple            # grid of the stored position embeddings (37, 37) for DINOv2 @518
    layerscale: bool = False
    pre_ln: bool = False
    final_ln: bool = True
    proj_dim: int = 0
    ln_eps: float = 1e-6
    n_blocks_run: int = -1
    out_all_tokens: bool = False
    recipe: str = "dinov2"
    pos_interp: str = "hf-4.44"    # how stored position embeddings are resampled to the run grid

    @property
    def grid(self):
        return self.img_h // self.patch, self.img_w // self.patch

    @property
    def n_tokens(self):
        return 1 + self.grid[0] * self.grid[1]

    @property
    def patch_k(self):
        return 3 * self.patch * self.patch

    @property
    def patch_k_pad(self):
        return (self.patch_k + 63) // 64 * 64

    @property
    def out_dim(self):
        return self.proj_dim if self.proj_dim else self.dim


CONFIGS = {
    # facebook/dinov2-base, as loaded at utils/embeddings.py:18 (image_size 518 -> 37x37 stored pos-embed)
    "dinov2_vitb14": VitConfig("dinov2_vitb14", 768, 12, 12, 3072, 14, 224, 224, (37, 37), layerscale=True,
                               recipe="dinov2"),
    # facebook/dinov2-small (BASELINE config 1)
    "dinov2_vits14": VitConfig("dinov2_vits14", 384, 12, 6, 1536, 14, 224, 224, (37, 37), layerscale=True,
                               recipe="dinov2"),
    # google/vit-base-patch16-224-in21k (utils/embeddings.py:26), layer_norm_eps 1e-12
    "vit_b16": VitConfig("vit_b16", 768, 12, 12, 3072, 16, 224, 224, (14, 14), ln_eps=1e-12, recipe="vit"),
    # open_clip ViT-B-32 laion2b_s34b_b79k (utils/embeddings.py:13-16): ln_pre, ln_post, 512-d projection
    "clip_b32": VitConfig("clip_b32", 768, 12, 12, 3072, 32, 224, 224, (7, 7), pre_ln=True, proj_dim=512,
                          ln_eps=1e-5, recipe="clip"),
    # tiny configurations used by the parity tests
    "tiny_dino": VitConfig("tiny_dino", 128, 2, 2, 256, 14, 224, 224, (37, 37), layerscale=True, recipe="dinov2"),
    "tiny_clip": VitConfig("tiny_clip", 128, 2, 2, 256, 32, 224, 224, (7, 7), pre_ln=True, proj_dim=128,
                           ln_eps=1e-5, recipe="clip"),
}


def random_weights(cfg: VitConfig, seed: int):
    """Seeded synthetic weights (no pretrained checkpoints exist offline): trunc-normal-ish N(0, 0.02)
    matrices, LayerNorm gains around 1, LayerScale around 1 -- numpy fp32, keyed like the oracle expects."""
    rng = np.random.default_rng(seed)

    def mat(*shape, std=0.02):
        return np.clip(rng.normal(0, std, size=shape), -2 * std, 2 * std).astype(np.float32)

    w = {
        "patch.w": mat(cfg.dim, 3, cfg.patch, cfg.patch),
        "patch.b": mat(cfg.dim),
        "cls": mat(cfg.dim),
        "pos": mat(1 + cfg.pos_grid[0] * cfg.pos_grid[1], cfg.dim),
        "ln_f.g": (1 + mat(cfg.dim, std=0.1)),
        "ln_f.b": mat(cfg.dim, std=0.1),
    }
    if cfg.pre_ln:
        w["ln_pre.g"] = 1 + mat(cfg.dim, std=0.1)
        w["ln_pre.b"] = mat(cfg.dim, std=0.1)
    if cfg.proj_dim:
        w["proj.w"] = mat(cfg.proj_dim, cfg.dim, std=cfg.dim ** -0.5)
    for l in range(cfg.depth):
        p = f"l{l}."
        w[p + "ln1.g"] = 1 + mat(cfg.dim, std=0.1)
        w[p + "ln1.b"] = mat(cfg.dim, std=0.1)
        # larger q/k scale than 0.02 so that the attention is not uniform (exercises the softmax)
        for n in ("q", "k", "v", "o"):
            w[p + n + ".w"] = mat(cfg.dim, cfg.dim, std=0.06 if n in "qk" else 0.03)
            w[p + n + ".b"] = mat(cfg.dim, std=0.02)
        w[p + "ln2.g"] = 1 + mat(cfg.dim, std=0.1)
        w[p + "ln2.b"] = mat(cfg.dim, std=0.1)
        w[p + "fc1.w"] = mat(cfg.mlp_dim, cfg.dim, std=0.03)
        w[p + "fc1.b"] = mat(cfg.mlp_dim)
        w[p + "fc2.w"] = mat(cfg.dim, cfg.mlp_dim, std=0.02)
        w[p + "fc2.b"] = mat(cfg.dim)
        if cfg.layerscale:
            w[p + "ls1"] = 1 + mat(cfg.dim, std=0.1)
            w[p + "ls2"] = 1 + mat(cfg.dim, std=0.1)
    return w


def interpolate_pos_embed(pos: np.ndarray, cfg: VitConfig) -> np.ndarray:
    """Stored (1 + gh*gw, D) position embeddings -> (n_tokens, D) for the run grid.  One-time load step.

    "hf-4.44": transformers 4.44.0 Dinov2Embeddings.interpolate_pos_encoding (the version the reference
    pins, environment.yml:275): bicubic, align_corners=False, scale_factor = (grid + 0.1) / stored_grid.
    "size": torch's size= form (transformers >= 4.46)."""
    gh, gw = cfg.pos_grid
    th, tw = cfg.grid
    if (gh, gw) == (th, tw):
        return pos.astype(np.float32)
    cls_pos, patch_pos = pos[:1], pos[1:]
    t = torch.from_numpy(patch_pos.astype(np.float32)).reshape(1, gh, gw, -1).permute(0, 3, 1, 2)
    if cfg.pos_interp == "hf-4.44":
        sf = (float((th + 0.1) / math.sqrt(gh * gw)), float((tw + 0.1) / math.sqrt(gh * gw)))
        t = torch.nn.functional.interpolate(t, scale_factor=sf, mode="bicubic", align_corners=False)
        if tuple(t.shape[-2:]) != (th, tw):
            raise ValueError("pos-embed interpolation produced an unexpected grid")
    else:
        t = torch.nn.functional.interpolate(t, size=(th, tw), mode="bicubic", align_corners=False)
    patch_new = t.permute(0, 2, 3, 1).reshape(th * tw, -1).numpy()
    return np.concatenate([cls_pos, patch_new], axis=0).astype(np.float32)


class PackedCrops:
    """A list of differently sized HxWx3 uint8 crops as ONE device-resident byte string + their shapes: what `VitEncoder.preprocess`
    uploads for a list of arrays, done once (a query batch whose crops already sit in HBM).  Slices like a list."""

    def __init__(self, crops, device="cuda", _src=None, _shapes=None, _offs=None):
        if _src is not None:
            self.src, self.shapes, self.offs = _src, _shapes, _offs
            return
        self.shapes = [tuple(c.shape[:2]) for c in crops]
        self.offs = np.concatenate([[0], np.cumsum([h * w * 3 for h, w in self.shapes])]).astype(np.int64)
        flat = np.concatenate([np.ascontiguousarray(c, dtype=np.uint8).reshape(-1) for c in crops]) if crops else np.zeros(0, np.uint8)
        self.src = torch.from_numpy(flat).to(device)

    def __len__(self):
        return len(self.shapes)

    def __getitem__(self, sl):
        if not isinstance(sl, slice):
            raise TypeError("PackedCrops supports slices only")
        lo, hi, step = sl.indices(len(self.shapes))
        assert step == 1
        return PackedCrops(None, _src=self.src[int(self.offs[lo]):int(self.offs[hi])], _shapes=self.shapes[lo:hi],
                           _offs=self.offs[lo:hi + 1] - self.offs[lo])


class VitEncoder:
    """Device-resident weights + batched forward through the C-ABI."""

    def __init__(self, cfg: VitConfig, weights: dict, device="cuda", precision=None, fold_layerscale=None):
        """precision: operand-term plan (see DEFAULT_PRECISION; None = $IBL_VIT_PREC or the default, "plain" = fp16 operands only);
        fold_layerscale: LayerScale multiplied into the output-projection weights at load, so that the library may run the residual GEMMs
        with the residual tile preloaded (lab: with IBL_GEMM_RESID_PRE=1; measured 2 % slower than the read-modify-write epilogue, so the
        default -- also with $IBL_VIT_FOLD_LS unset -- keeps the LayerScale vectors and round 3's arithmetic)"""
        if cfg.depth > MAX_LAYERS:
            raise ValueError("too many layers")
        import os
        self.precision = precision if precision is not None else os.environ.get("IBL_VIT_PREC", MODEL_PRECISION.get(cfg.name, DEFAULT_PRECISION))
        patch_terms, layer_terms = parse_precision(self.precision)
        self.fold_layerscale = (os.environ.get("IBL_VIT_FOLD_LS", "0") == "1") if fold_layerscale is None else bool(fold_layerscale)
        self.cfg = cfg
        self.device = torch.device(device)
        self._keep = []                     # device tensors referenced by the structs

        def dev_f32(a):
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)
            self._keep.append(t)
            return t.data_ptr()

        def dev_f16(a):
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device).to(torch.float16).contiguous()
            self._keep.append(t)
            return t.data_ptr()

        pos = interpolate_pos_embed(weights["pos"], cfg)
        wp = np.zeros((cfg.dim, cfg.patch_k_pad), dtype=np.float32)
        wp[:, :cfg.patch_k] = weights["patch.w"].reshape(cfg.dim, -1)
        W = VitWeights()
        W.w_patch = dev_f16(wp)
        if patch_terms == 2:
            W.w_patch_lo = dev_f16(weight_lo(wp))
        W.b_patch = dev_f32(weights["patch.b"]) if "patch.b" in weights else None
        W.cls_pos = dev_f32(weights["cls"].reshape(-1) + pos[0])
        W.pos_patch = dev_f32(pos[1:])
        if cfg.pre_ln:
            W.ln_pre_g, W.ln_pre_b = dev_f32(weights["ln_pre.g"]), dev_f32(weights["ln_pre.b"])
        if cfg.final_ln:
            W.ln_f_g, W.ln_f_b = dev_f32(weights["ln_f.g"]), dev_f32(weights["ln_f.b"])
        if cfg.proj_dim:
            W.w_proj = dev_f16(weights["proj.w"])
            if self.precision != "plain":      # the last arithmetic before the output: three-term operands (free at batch x dim x out_dim)
                W.w_proj_x = dev_f16(split_terms(weights["proj.w"], 3))
        for l in range(cfg.depth):
            p = f"l{l}."
            L = W.layers[l]
            L.ln1_g, L.ln1_b = dev_f32(weights[p + "ln1.g"]), dev_f32(weights[p + "ln1.b"])
            L.w_qkv = dev_f16(np.concatenate([weights[p + "q.w"], weights[p + "k.w"], weights[p + "v.w"]], axis=0))
            L.b_qkv = dev_f32(np.concatenate([weights[p + "q.b"], weights[p + "k.b"], weights[p + "v.b"]], axis=0))
            # fold_layerscale (lab, off by default): x += ls * (a W^T + b) = a (diag(ls) W)^T + ls * b.  Without a scale vector in the
            # epilogue the library can preload the residual tile into the accumulators (EPI_RESID_PRE_F32, csrc/vit.hip).
            w_o, b_o, w_fc2, b_fc2 = weights[p + "o.w"], weights[p + "o.b"], weights[p + "fc2.w"], weights[p + "fc2.b"]
            fold = self.fold_layerscale              # (without LayerScale: "folded" = no 1 / S vectors for the second terms either)
            if fold and cfg.layerscale:
                ls1, ls2 = np.asarray(weights[p + "ls1"], np.float32), np.asarray(weights[p + "ls2"], np.float32)
                w_o, b_o = ls1[:, None] * np.asarray(w_o, np.float32), ls1 * np.asarray(b_o, np.float32)
                w_fc2, b_fc2 = ls2[:, None] * np.asarray(w_fc2, np.float32), ls2 * np.asarray(b_fc2, np.float32)
            L.w_o, L.b_o = dev_f16(w_o), dev_f32(b_o)
            L.ln2_g, L.ln2_b = dev_f32(weights[p + "ln2.g"]), dev_f32(weights[p + "ln2.b"])
            L.w_fc1, L.b_fc1 = dev_f16(weights[p + "fc1.w"]), dev_f32(weights[p + "fc1.b"])
            L.w_fc2, L.b_fc2 = dev_f16(w_fc2), dev_f32(b_fc2)
            if cfg.layerscale and not fold:
                L.ls1, L.ls2 = dev_f32(weights[p + "ls1"]), dev_f32(weights[p + "ls2"])
            nrun_ = cfg.depth if cfg.n_blocks_run < 0 else cfg.n_blocks_run
            if (l in layer_terms or "*" in layer_terms) and not (l == nrun_ - 1 and not cfg.out_all_tokens):     # the CLS-only last block stays plain
                tq, to, t1, t2 = layer_terms[l] if l in layer_terms else layer_terms["*"]
                ones = np.ones(cfg.dim, dtype=np.float32)
                if tq > 1:
                    L.w_qkv_x = dev_f16(split_terms(np.concatenate([weights[p + "q.w"], weights[p + "k.w"], weights[p + "v.w"]], axis=0), tq))
                    L.qkv_terms = tq
                kext = os.environ.get("IBL_VIT_RESID_KEXT", "1") != "0"       # 0: second weight terms of o / fc2 as launches of their own (rounds 3-4a)
                self._act_terms = max(getattr(self, "_act_terms", 1), to if kext else 1, t2 if kext else 1)
                if to > 1 and kext:
                    L.w_o_x, L.o_terms = dev_f16(split_terms(w_o, to)), to
                elif to > 1:
                    if to > 2:
                        raise ValueError("o = 3 needs the K-extended form")
                    L.w_o_lo = dev_f16(weight_lo(w_o))
                    if not fold:                             # (no vector: the library adds the term with the factor 1 / S, residual preloaded)
                        L.ls1_lo = dev_f32((weights[p + "ls1"] if cfg.layerscale else ones) / SPLIT_SCALE)
                if t1 > 1:
                    L.w_fc1_x = dev_f16(split_terms(weights[p + "fc1.w"], t1))
                    L.fc1_terms = t1
                if t2 > 1 and kext:
                    L.w_fc2_x, L.fc2_terms = dev_f16(split_terms(w_fc2, t2)), t2
                elif t2 > 1:
                    if t2 > 2:
                        raise ValueError("fc2 = 3 needs the K-extended form")
                    L.w_fc2_lo = dev_f16(weight_lo(w_fc2))
                    if not fold:
                        L.ls2_lo = dev_f32((weights[p + "ls2"] if cfg.layerscale else ones) / SPLIT_SCALE)
        self.W = W
        flags = 0
        flags |= FLAG_LAYERSCALE if cfg.layerscale else 0
        flags |= FLAG_PRE_LN if cfg.pre_ln else 0
        flags |= FLAG_FINAL_LN if cfg.final_ln else 0
        flags |= FLAG_PROJ if cfg.proj_dim else 0
        flags |= FLAG_OUT_ALL_TOKENS if cfg.out_all_tokens else 0
        flags |= {1: 0, 2: FLAG_ACT_TERMS2, 3: FLAG_ACT_TERMS3}[getattr(self, "_act_terms", 1)]
        nrun = cfg.depth if cfg.n_blocks_run < 0 else cfg.n_blocks_run
        self.desc = VitDesc(cfg.dim, cfg.depth, cfg.heads, cfg.mlp_dim, cfg.patch, cfg.img_h, cfg.img_w, cfg.n_tokens,
                            cfg.patch_k_pad, flags, nrun, cfg.out_dim, cfg.ln_eps)
        self.recipe = pp.RECIPES[cfg.recipe]
        self._ws = None
        self._plan_cache = {}

    # ---- preprocessing -------------------------------------------------------------------------
    def preprocess(self, crops, want_u8=False):
        """crops: list of HxWx3 uint8 numpy arrays (RGB as the detector hands them over), or a single
        uint8 device/host tensor (N, H, W, 3) of equally sized crops.  Returns fp16 patch matrix (device)."""
        r = self.recipe
        if isinstance(crops, PackedCrops):
            shapes, src = crops.shapes, crops.src.to(self.device)
        elif isinstance(crops, torch.Tensor):
            n, h, w, _ = crops.shape
            shapes = [(h, w)] * n
            src = crops.to(self.device).contiguous().view(-1)
        else:
            shapes = [c.shape[:2] for c in crops]
            src = torch.from_numpy(np.concatenate([np.ascontiguousarray(c, dtype=np.uint8).reshape(-1) for c in crops]))
            src = src.to(self.device)
        key = tuple(shapes) if len(set(shapes)) > 1 else (shapes[0], len(shapes))
        plan = self._plan_cache.get(key)
        if plan is None:
            descs, tables, src_bytes, tmp_bytes, max_h = pp.plan_batch(r, shapes)
            d_descs = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.device)
            d_tables = torch.from_numpy(tables).to(self.device)
            plan = (d_descs, d_tables, src_bytes, tmp_bytes, max_h)
            if len(self._plan_cache) < 64:
                self._plan_cache[key] = plan
        d_descs, d_tables, src_bytes, tmp_bytes, max_h = plan
        assert src.numel() == src_bytes
        n = len(shapes)
        P = self.cfg.n_tokens - 1
        tmp = torch.empty(max(tmp_bytes, 16), dtype=torch.uint8, device=self.device)
        patches = torch.empty((n * P, self.cfg.patch_k_pad), dtype=torch.float16, device=self.device)
        out_u8 = torch.empty((n, r.out_h, r.out_w, 3), dtype=torch.uint8, device=self.device) if want_u8 else None
        mean = (C.c_float * 3)(*r.mean)
        std = (C.c_float * 3)(*r.std)
        st = _lib.lib.ibl_preprocess_crops(src.data_ptr(), d_descs.data_ptr(), n, max_h, d_tables.data_ptr(),
                                           tmp.data_ptr(), r.out_h, r.out_w, self.cfg.patch, self.cfg.patch_k_pad,
                                           1 if r.swap_rb else 0, mean, std, patches.data_ptr(),
                                           out_u8.data_ptr() if want_u8 else None,
                                           torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "ibl_preprocess_crops")
        return (patches, out_u8) if want_u8 else patches

    # ---- forward -------------------------------------------------------------------------------
    def forward_patches(self, patches: torch.Tensor, lane: int = 0) -> torch.Tensor:
        """lane: which of the encoder's workspaces to use (concurrent forwards on different streams need their own)."""
        P = self.cfg.n_tokens - 1
        assert patches.dtype == torch.float16 and patches.is_cuda and patches.shape[1] == self.cfg.patch_k_pad
        batch = patches.shape[0] // P
        ws_bytes = _lib.lib.ibl_vit_workspace_bytes(C.byref(self.desc), batch)
        if not hasattr(self, "_ws_lanes"):
            self._ws_lanes = {}
        ws = self._ws_lanes.get(lane)
        if ws is None or ws.numel() < ws_bytes:
            ws = self._ws_lanes[lane] = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
        if self.cfg.out_all_tokens:
            out = torch.empty((batch, self.cfg.n_tokens, self.cfg.dim), dtype=torch.float32, device=self.device)
        else:
            out = torch.empty((batch, self.cfg.out_dim), dtype=torch.float32, device=self.device)
        st = _lib.lib.ibl_vit_forward(C.byref(self.desc), C.byref(self.W), patches.data_ptr(), batch, out.data_ptr(),
                                      ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "ibl_vit_forward")
        return out

    def patches_from_pixels(self, x: torch.Tensor) -> torch.Tensor:
        """(B, 3, H, W) float model input (already normalised) -> fp16 patch matrix.  Data-layout plumbing
        only (used by tests and by callers that hold pre-normalised tensors)."""
        cfg = self.cfg
        B = x.shape[0]
        gh, gw = cfg.grid
        p = cfg.patch
        t = x.to(self.device, torch.float32).reshape(B, 3, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5)
        t = t.reshape(B * gh * gw, 3 * p * p)
        out = torch.zeros((B * gh * gw, cfg.patch_k_pad), dtype=torch.float16, device=self.device)
        out[:, :cfg.patch_k] = t.to(torch.float16)
        return out

    def embed(self, crops, max_batch=512, streams=1, min_split=128, lane=0) -> torch.Tensor:
        """crops -> (N, out_dim) fp32 device tensor (un-normalised CLS embedding, as the reference returns).

        streams > 1: a batch of >= min_split crops is embedded as that many micro-batches on their own HIP streams.  The layers of
        a ViT are a strict chain, so on one stream the MFMA-bound K loops of a GEMM never overlap its own HBM-bound epilogue
        nor the LayerNorm / attention kernels around it; two independent chains fill each other's gaps (measured: embed 20.9 ->
        20.2 ms for 224 crops).  Off by default: the kernels of the two chains share the GPU, so per-kernel timings (the bench's
        roofline line) no longer describe one kernel.  Same arithmetic per crop either way."""
        n = crops.shape[0] if isinstance(crops, torch.Tensor) else len(crops)
        if n == 0:
            return torch.empty((0, self.cfg.out_dim), dtype=torch.float32, device=self.device)
        if streams > 1 and min_split <= n <= max_batch:
            if not hasattr(self, "_streams") or len(self._streams) != streams:
                self._streams = [torch.cuda.Stream(device=self.device) for _ in range(streams)]
            cur = torch.cuda.current_stream(self.device)
            bounds = [n * k // streams for k in range(streams + 1)]
            outs = []
            for k, st in enumerate(self._streams):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    outs.append(self.forward_patches(self.preprocess(crops[bounds[k]:bounds[k + 1]]), lane=k))
            for st in self._streams:
                cur.wait_stream(st)
            for o in outs:
                o.record_stream(cur)
            return torch.cat(outs, dim=0)
        outs = []
        for i in range(0, n, max_batch):
            outs.append(self.forward_patches(self.preprocess(crops[i:i + max_batch]), lane=lane))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)


LINEAR_F16, LINEAR_GELU_F16, LINEAR_RESID_F32, LINEAR_F32 = 0, 1, 2, 4


def linear_f16(x: torch.Tensor, W: torch.Tensor, bias=None, epilogue=LINEAR_F16, out=None, scale=None) -> torch.Tensor:
    """out = epilogue(x W^T + bias) through `ibl_linear_f16` (the encoder's GEMM kernel on its own).
    x (rows, n_in) fp16, W (n_out, n_in) fp16 (nn.Linear layout), bias / scale fp32 (n_out).  LINEAR_RESID_F32
    accumulates into `out` (fp32), the other epilogues allocate it when it is not given."""
    assert x.dtype == torch.float16 and W.dtype == torch.float16 and x.is_cuda and W.is_cuda
    assert x.stride(-1) == 1 and W.stride(-1) == 1
    rows, n_in = x.shape
    n_out = W.shape[0]
    if out is None:
        if epilogue == LINEAR_RESID_F32:
            raise ValueError("LINEAR_RESID_F32 accumulates into `out`")
        out = torch.empty((rows, n_out), device=x.device, dtype=torch.float32 if epilogue == LINEAR_F32 else torch.float16)
    want = torch.float32 if epilogue in (LINEAR_RESID_F32, LINEAR_F32) else torch.float16
    assert out.dtype == want and out.shape == (rows, n_out) and out.stride(-1) == 1
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == n_out and bias.is_contiguous()
    if scale is not None:
        assert scale.dtype == torch.float32 and scale.numel() == n_out and scale.is_contiguous()
    st = _lib.lib.ibl_linear_f16(x.data_ptr(), x.stride(0), W.data_ptr(), W.stride(0), bias.data_ptr() if bias is not None else None,
                                  scale.data_ptr() if scale is not None else None, rows, n_out, n_in, epilogue, out.data_ptr(),
                                  out.stride(0), torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "ibl_linear_f16")
    return out

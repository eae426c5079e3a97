"""DATOR RGB-D dual-stream encoder: two TransReID ViT-B/16 streams at 256x128 (11 of 12 blocks,
no final norm, LoRA folded into the QKV weights at load) on the shared ViT kernels, and the fusion head
(/root/reference/dator/model/make_model.py:629-843) as dedicated HIP kernels behind `ibl_dator_head_forward`.

The reference's `dator_wrapper.get_model_input` is missing from its repository; the preprocessing follows the only
surviving record, dator/get_embeds.py:80-87 (RGB: resize to 256x128, ToTensor, mean = std = 0.5) and :129-136
(depth: bilinear resize to 256x128, tiled to 3 channels, clipped to [0, 50], /50, (x - 0.5) / 0.5)."""
import ctypes as C
import dataclasses

import numpy as np
import torch

from . import _lib
from . import vit as V

MIN_DEPTH, MAX_DEPTH = 0.0, 50.0

STREAM_CFG = V.VitConfig("transreid_b16", 768, 12, 12, 3072, 16, 256, 128, (16, 8), layerscale=False, final_ln=False,
                         n_blocks_run=11, out_all_tokens=True, recipe="dator_rgb")

# Operand-term plan of the two TransReID streams (ibloc_amd.vit syntax).  Round 4: every GEMM of every block with three-term operands --
# weights hi + lo, and the GEMM's input hi + lo: the LayerNorm outputs, the attention output and the GELU hidden layer, all as K-extended
# rows [a_hi | a_lo S | a_hi / S] written by the producing kernel (one accumulation of K' = 3 K per GEMM).  SURVEY 8d's gate is 1e-3 per embedding; with the ViT
# plan (FAST_PRECISION) DATOR sits at 1.2e-3 mean / 3.0e-3 max over 448 u8 crops because the fusion head amplifies the streams' token
# error 1.5x (3x for single crops); this plan: 3.1e-4 mean / 5.9e-4 max, at 1.9x the encoder time (38 against 20 ms per 224 crops).
# What is left is the fp16 rounding of q / k / v, of the softmax probabilities and of the input pixels (tools/sim_dator_rounding.py).
DEFAULT_PRECISION = "p2;*:3333"
FAST_PRECISION = V.DEFAULT_PRECISION          # the ViT encoders' plan: 1.9x faster, over the 1e-3 gate

HEAD_LINEARS = ["proj_local_rgb", "proj_global_rgb", "merge_rgb", "proj_local_depth", "proj_global_depth", "merge_depth",
                "Q_r", "V_r", "Q_d", "V_d"]
ATTN_OPS = ["r2r", "d2d", "d2r", "r2d"]


def random_head_weights(seed: int, reduced=128, in_planes=768):
    rng = np.random.default_rng(seed)

    def mat(*shape, std=0.05):
        return rng.normal(0, std, size=shape).astype(np.float32)

    w = {}
    for n in HEAD_LINEARS:
        k = in_planes if n.startswith("proj_") else (2 * reduced if n.startswith("merge") else reduced)
        w[n + ".w"], w[n + ".b"] = mat(reduced, k, std=k ** -0.5), mat(reduced, std=0.05)
    for op in ATTN_OPS:
        w[op + ".sel.w"], w[op + ".sel.b"] = mat(48, reduced, std=0.3), mat(48, std=0.3)
        w[op + ".aw.w"], w[op + ".aw.b"] = mat(24, reduced, std=0.3), mat(24, std=0.1)
        w[op + ".ffn.w"], w[op + ".ffn.b"] = mat(reduced, reduced, std=reduced ** -0.5), mat(reduced, std=0.05)
        w[op + ".norm.g"], w[op + ".norm.b"] = 1 + mat(reduced, std=0.1), mat(reduced, std=0.1)
    for i, (co, ci) in enumerate([(128, 2 * reduced), (32, 128), (8, 32), (2, 8)]):
        w[f"hyper.{i}.w"], w[f"hyper.{i}.b"] = mat(co, ci, 3, 3, std=(9 * ci) ** -0.5), mat(co, std=0.05)
    return w


def random_stream_weights(seed: int):
    """TransReID stream weights in the VitEncoder dict layout (+ LoRA factors on blocks 10 and 11, vit_pytorch.py:381-391)."""
    w = V.random_weights(STREAM_CFG, seed)
    rng = np.random.default_rng(seed + 7)
    for l in (10, 11):
        w[f"l{l}.lora_down"] = rng.normal(0, 1.0, size=(768, 4)).astype(np.float32)
        w[f"l{l}.lora_up"] = rng.normal(0, 0.01, size=(4, 2304)).astype(np.float32)     # zeros at init in the reference
    return w


def fold_lora(w: dict) -> dict:
    """W_eff = W_qkv + (A @ B)^T  (AttentionWithLoRA.forward, vit_pytorch.py:182-185), folded once at load."""
    out = dict(w)
    for l in range(STREAM_CFG.depth):
        if f"l{l}.lora_down" in w:
            delta = (w[f"l{l}.lora_down"].astype(np.float64) @ w[f"l{l}.lora_up"].astype(np.float64)).T.astype(np.float32)   # (2304, 768)
            for i, n in enumerate("qkv"):
                out[f"l{l}.{n}.w"] = w[f"l{l}.{n}.w"] + delta[i * 768:(i + 1) * 768]
            del out[f"l{l}.lora_down"], out[f"l{l}.lora_up"]          # folded: nothing downstream may add the term again
    return out


_HEAD_NAMES = {"proj_local_rgb": "project_local_rgb", "proj_global_rgb": "project_global_rgb", "merge_rgb": "merge_local_global_rgb",
               "proj_local_depth": "project_local_depth", "proj_global_depth": "project_global_depth",
               "merge_depth": "merge_local_global_depth", "Q_r": "Q_r", "V_r": "V_r", "Q_d": "Q_d", "V_d": "V_d"}


def fourdnet_state_dict_to_weights(state_dict: dict):
    """A `build_FourDNet` checkpoint (what the reference's `load_model('.../dator_best_tum.pth')` reads,
    utils/embeddings.py:101-103 -> build_FourDNet.load_param, dator/model/make_model.py:620-626) -> (rgb stream weights, depth stream
    weights, head weights) in the dict layouts DatorEncoder takes.

    Follows load_param: a `module.` prefix (DataParallel checkpoints) is stripped, `classifier*` entries are skipped.  `base.*` is the
    RGB TransReID stream, `base2.*` the depth stream (make_model.py:459-463); their LoRA factors (`attn.qkv_lora_down_matrix` /
    `_up_matrix`, blocks 10-11, vit_pytorch.py:176-177) stay separate entries here and are folded by DatorEncoder.  Parameters the
    forward never reads are ignored: the streams' own `norm` / `fc` (local_feature=True returns before them, vit_pytorch.py:437-443)
    and `sie_embed`.  A missing parameter raises KeyError naming the checkpoint key."""
    sd = {}
    for k, v in state_dict.items():
        k = k.replace("module.", "")
        if k.find("classifier") != -1:
            continue
        sd[k] = v

    def g(k):
        if k not in sd:
            raise KeyError(f"DATOR checkpoint lacks '{k}'")
        v = sd[k]
        v = v.detach().float().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        return np.ascontiguousarray(v, dtype=np.float32)

    def stream(prefix):
        w = {"cls": g(prefix + "cls_token").reshape(-1), "pos": g(prefix + "pos_embed")[0],
             "patch.w": g(prefix + "patch_embed.proj.weight"), "patch.b": g(prefix + "patch_embed.proj.bias")}
        n_pos = 1 + STREAM_CFG.pos_grid[0] * STREAM_CFG.pos_grid[1]
        if w["pos"].shape != (n_pos, STREAM_CFG.dim):
            raise ValueError(f"{prefix}pos_embed has shape {w['pos'].shape}, the 256x128 / stride-16 stream expects {(n_pos, STREAM_CFG.dim)}")
        dim = STREAM_CFG.dim
        for l in range(STREAM_CFG.depth):
            p, q = f"{prefix}blocks.{l}.", f"l{l}."
            w[q + "ln1.g"], w[q + "ln1.b"] = g(p + "norm1.weight"), g(p + "norm1.bias")
            w[q + "ln2.g"], w[q + "ln2.b"] = g(p + "norm2.weight"), g(p + "norm2.bias")
            wq, bq = g(p + "attn.qkv.weight"), g(p + "attn.qkv.bias")
            for i, n in enumerate("qkv"):
                w[q + n + ".w"], w[q + n + ".b"] = wq[i * dim:(i + 1) * dim].copy(), bq[i * dim:(i + 1) * dim].copy()
            w[q + "o.w"], w[q + "o.b"] = g(p + "attn.proj.weight"), g(p + "attn.proj.bias")
            w[q + "fc1.w"], w[q + "fc1.b"] = g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias")
            w[q + "fc2.w"], w[q + "fc2.b"] = g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias")
            if p + "attn.qkv_lora_down_matrix" in sd:
                w[q + "lora_down"], w[q + "lora_up"] = g(p + "attn.qkv_lora_down_matrix"), g(p + "attn.qkv_lora_up_matrix")
        # VitEncoder reads a final-LayerNorm entry only when cfg.final_ln; the streams run without it
        return w

    hw = {}
    for mine, ref in _HEAD_NAMES.items():
        hw[mine + ".w"], hw[mine + ".b"] = g(ref + ".weight"), g(ref + ".bias")
    for op in ATTN_OPS:
        hw[op + ".sel.w"], hw[op + ".sel.b"] = g(f"{op}_selector.0.weight"), g(f"{op}_selector.0.bias")
        hw[op + ".aw.w"], hw[op + ".aw.b"] = g(f"{op}_attn_weights.0.weight"), g(f"{op}_attn_weights.0.bias")
        hw[op + ".ffn.w"], hw[op + ".ffn.b"] = g(f"{op}_ffn.weight"), g(f"{op}_ffn.bias")
        hw[op + ".norm.g"], hw[op + ".norm.b"] = g(f"{op}_norm.weight"), g(f"{op}_norm.bias")
    for i in range(4):
        hw[f"hyper.{i}.w"], hw[f"hyper.{i}.b"] = g(f"hypernet.{2 * i}.weight"), g(f"hypernet.{2 * i}.bias")
    return stream("base."), stream("base2."), hw


class DatorHeadWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ["proj_local_rgb_w", "proj_local_rgb_b", "proj_global_rgb_w", "proj_global_rgb_b", "merge_rgb_w", "merge_rgb_b",
                 "proj_local_depth_w", "proj_local_depth_b", "proj_global_depth_w", "proj_global_depth_b", "merge_depth_w",
                 "merge_depth_b", "Q_r_w", "Q_r_b", "V_r_w", "V_r_b", "Q_d_w", "Q_d_b", "V_d_w", "V_d_b"]
                + [f"{op}_{p}" for op in ATTN_OPS for p in ("sel_w", "sel_b", "aw_w", "aw_b", "ffn_w", "ffn_b", "norm_g", "norm_b")]
                + [f"hyper{i}_{p}" for i in range(4) for p in ("w", "b")]]


class DatorEncoder:
    def __init__(self, rgb_weights: dict, depth_weights: dict, head_weights: dict, device="cuda", precision=None):
        """precision: operand-term plan of the two streams (ibloc_amd.vit syntax; None = $IBL_VIT_PREC or DEFAULT_PRECISION below)"""
        import os
        self.device = torch.device(device)
        self.precision = precision if precision is not None else os.environ.get("IBL_VIT_PREC", DEFAULT_PRECISION)
        self.rgb = V.VitEncoder(STREAM_CFG, fold_lora(rgb_weights), device=device, precision=self.precision)
        self.depth = V.VitEncoder(STREAM_CFG, fold_lora(depth_weights), device=device, precision=self.precision)
        self._keep = []
        H = DatorHeadWeights()
        for name, _ in DatorHeadWeights._fields_:
            key = name
            for a, b in (("_w", ".w"), ("_b", ".b"), ("_g", ".g")):
                if key.endswith(a):
                    key = key[:-len(a)] + b
                    break
            for op in ATTN_OPS:
                for part in ("sel", "aw", "ffn", "norm"):
                    key = key.replace(f"{op}_{part}", f"{op}.{part}")
            key = key.replace("hyper0", "hyper.0").replace("hyper1", "hyper.1").replace("hyper2", "hyper.2").replace("hyper3", "hyper.3")
            arr = np.ascontiguousarray(head_weights[key], dtype=np.float32)
            if key.startswith("hyper") and key.endswith(".w"):
                # conv weights [co][ci][3][3] -> [tap][ci][co] (output channels contiguous: one per thread)
                arr = np.ascontiguousarray(arr.reshape(arr.shape[0], arr.shape[1], 9).transpose(2, 1, 0))
            t = torch.from_numpy(arr).to(self.device)
            self._keep.append(t)
            setattr(H, name, t.data_ptr())
        self.H = H
        self._ws = None

    def preprocess_depth(self, depth_crops) -> torch.Tensor:
        """list of (h, w) float depth crops, or one float32 tensor (N, h, w) of equally sized crops (host or device) -> fp16 patch
        matrix of the depth stream (bilinear resize to 256x128, 3 identical channels, clip, scale, normalise)."""
        if isinstance(depth_crops, torch.Tensor):
            n, h, w = depth_crops.shape
            sizes = np.tile(np.array([[h, w]], dtype=np.int32), (n, 1))
            flat = depth_crops.to(self.device, torch.float32).contiguous().view(-1)
        else:
            n = len(depth_crops)
            sizes = np.array([[c.shape[0], c.shape[1]] for c in depth_crops], dtype=np.int32)
            flat = torch.from_numpy(np.concatenate([np.ascontiguousarray(c, dtype=np.float32).reshape(-1) for c in depth_crops])).to(self.device)
        offs = np.concatenate([[0], np.cumsum(sizes[:, 0].astype(np.int64) * sizes[:, 1])]).astype(np.int64)
        d_sizes = torch.from_numpy(sizes).to(self.device)
        d_offs = torch.from_numpy(offs).to(self.device)
        cfg = STREAM_CFG
        patches = torch.empty((n * (cfg.n_tokens - 1), cfg.patch_k_pad), dtype=torch.float16, device=self.device)
        st = _lib.lib.ibl_preprocess_depth(flat.data_ptr(), d_offs.data_ptr(), d_sizes.data_ptr(), n, cfg.img_h, cfg.img_w, cfg.patch,
                                           cfg.patch_k_pad, float(MIN_DEPTH), float(MAX_DEPTH), patches.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "ibl_preprocess_depth")
        return patches

    def head(self, rgb_tokens: torch.Tensor, depth_tokens: torch.Tensor, lane: int = 0) -> torch.Tensor:
        B = rgb_tokens.shape[0]
        ws_bytes = _lib.lib.ibl_dator_head_workspace_bytes(B)
        if self._ws is None:
            self._ws = {}
        ws = self._ws.get(lane)
        if ws is None or ws.numel() < ws_bytes:
            ws = self._ws[lane] = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
        out = torch.empty((B, 128), dtype=torch.float32, device=self.device)
        st = _lib.lib.ibl_dator_head_forward(C.byref(self.H), rgb_tokens.data_ptr(), depth_tokens.data_ptr(), B, out.data_ptr(),
                                             ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "ibl_dator_head_forward")
        return out

    def forward_pixels(self, rgb: torch.Tensor, depth: torch.Tensor) -> torch.Tensor:
        """(B, 3, 256, 128) pre-normalised tensors -> (B, 128)."""
        rt = self.rgb.forward_patches(self.rgb.patches_from_pixels(rgb))
        dt = self.depth.forward_patches(self.depth.patches_from_pixels(depth))
        return self.head(rt, dt)

    def embed(self, rgb_crops, depth_crops=None, lane: int = 0, max_batch: int = 512) -> torch.Tensor:
        """rgb_crops: list of HxWx3 uint8 (RGB order, no channel swap -- get_embeds.py feeds RGB) or a uint8 tensor (N, H, W, 3);
        depth_crops: list of (h, w) float or a float32 tensor (N, h, w).  `embed((rgb, depth))` is the form LocaliseEngine calls
        (its `crops` argument is handed through as is).  lane: workspace index for concurrent forwards on different streams."""
        if depth_crops is None:
            rgb_crops, depth_crops = rgb_crops
        n = rgb_crops.shape[0] if isinstance(rgb_crops, torch.Tensor) else len(rgb_crops)
        outs = []
        for i in range(0, n, max_batch):
            rt = self.rgb.forward_patches(self.rgb.preprocess(rgb_crops[i:i + max_batch]), lane=lane)
            dt = self.depth.forward_patches(self.preprocess_depth(depth_crops[i:i + max_batch]), lane=lane)
            outs.append(self.head(rt, dt, lane=lane))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

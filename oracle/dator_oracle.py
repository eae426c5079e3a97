"""ORACLE (test infrastructure; never imported by the product path).

Torch-fp32 restatement of DATOR's forward:
  * TransReID stream with local_feature=True: 11 of 12 blocks, no final norm, LoRA added to the QKV projection of
    the last two blocks (dator/model/backbones/vit_pytorch.py:182-185, 381-391, 422-443);
  * fusion head build_FourDNet.forward (dator/model/make_model.py:629-843): global/local projections and merge,
    hyper-network gates, Q/V, four deformable-sampling attentions (grid_sample, align_corners=True, bilinear, zero
    padding; x = first 24 selector channels, y = last 24), residual + LayerNorm (eps 1e-5), gated mean.
Pinned against the reference's own build_FourDNet run in the build container with the same seeded weights
(tests/golden/dator_golden.npz, tools/gen_golden_dator.py).  Trained weights (dator_best_tum.pth) and the reference's
`dator_wrapper.get_model_input` are not in its repository: preprocessing follows dator/get_embeds.py:80-87,129-136
and pretrained-weight parity is unpinned."""
import numpy as np
import torch
from PIL import Image

from . import vit_oracle as vo

F = torch.nn.functional


def stream_tokens(w: dict, cfg, pixels: np.ndarray) -> np.ndarray:
    """(B, 3, 256, 128) -> (B, 129, 768): vit_oracle forward with the LoRA term folded exactly as F.linear(x, (A@B).T)."""
    w2 = dict(w)
    for l in range(cfg.depth):
        if f"l{l}.lora_down" in w:
            delta = (torch.from_numpy(w[f"l{l}.lora_down"]) @ torch.from_numpy(w[f"l{l}.lora_up"])).T.numpy()
            for i, n in enumerate("qkv"):
                w2[f"l{l}.{n}.w"] = w[f"l{l}.{n}.w"] + delta[i * cfg.dim:(i + 1) * cfg.dim]
    return vo.vit_forward(w2, cfg, pixels, all_tokens=True)


@torch.no_grad()
def head_forward(hw: dict, rgb_tokens: np.ndarray, depth_tokens: np.ndarray) -> np.ndarray:
    h = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in hw.items()}
    xr, xd = torch.from_numpy(rgb_tokens), torch.from_numpy(depth_tokens)
    B, N = xr.shape[0], xr.shape[1] - 1

    def lin(name, x):
        return F.linear(x, h[name + ".w"], h[name + ".b"])

    def merged(x, side):
        g = lin(f"proj_global_{side}", x[:, 0])
        loc = lin(f"proj_local_{side}", x[:, 1:])
        return lin(f"merge_{side}", torch.cat((g.unsqueeze(1).repeat(1, N, 1), loc), -1))

    fr, fd = merged(xr, "rgb"), merged(xd, "depth")
    dsp = fd.reshape(B, 16, 8, 128).permute(0, 3, 1, 2)
    rsp = fr.reshape(B, 16, 8, 128).permute(0, 3, 1, 2)
    x = torch.cat((dsp, rsp), dim=1)
    for i in range(4):
        x = F.conv2d(x, h[f"hyper.{i}.w"], h[f"hyper.{i}.b"], padding=1)
        if i < 3:
            x = F.relu(x)
    filt = F.softmax(x.permute(0, 2, 3, 1), dim=-1)
    rgb_f, depth_f = filt[..., 0].reshape(B, 128, 1), filt[..., 1].reshape(B, 128, 1)
    q_r, v_r, q_d, v_d = lin("Q_r", fr), lin("V_r", fr), lin("Q_d", fd), lin("V_d", fd)

    def deform(op, q, v):
        sel = torch.sigmoid(lin(op + ".sel", q))
        aw = F.softmax(lin(op + ".aw", q), dim=-1)
        grid = torch.stack((sel[:, :, :24], sel[:, :, 24:]), -1) * 2 - 1
        vm = v.permute(0, 2, 1).reshape(B, 128, 16, 8)
        samp = F.grid_sample(vm, grid, align_corners=True).permute(0, 2, 3, 1)
        return lin(op + ".ffn", torch.sum(samp * aw.unsqueeze(-1), dim=-2))

    def ln(op, x):
        return F.layer_norm(x, (128,), h[op + ".norm.g"], h[op + ".norm.b"], 1e-5)

    fr = ln("r2r", fr + deform("r2r", q_r, v_r))
    fd = ln("d2d", fd + deform("d2d", q_d, v_d))
    fr = ln("d2r", fr + deform("d2r", q_d, v_r) * rgb_f)
    fd = ln("r2d", fd + deform("r2d", q_r, v_d) * depth_f)
    return torch.mean(fd * depth_f + fr * rgb_f, dim=-2).numpy()


def forward(rgb_w, depth_w, head_w, cfg, rgb_pixels, depth_pixels):
    return head_forward(head_w, stream_tokens(rgb_w, cfg, rgb_pixels), stream_tokens(depth_w, cfg, depth_pixels))


def preprocess_rgb(crop: np.ndarray) -> np.ndarray:
    """dator/get_embeds.py:80-87: ToPILImage, Resize([256, 128]) (bilinear), ToTensor, Normalize(0.5, 0.5)."""
    res = np.asarray(Image.fromarray(np.ascontiguousarray(crop, dtype=np.uint8)).resize((128, 256), resample=Image.BILINEAR))
    x = (res.astype(np.float64) * (1 / 255)).astype(np.float32)
    return np.ascontiguousarray(((x - np.float32(0.5)) / np.float32(0.5)).transpose(2, 0, 1))


def preprocess_depth(depth_crop: np.ndarray, dmin=0.0, dmax=50.0) -> np.ndarray:
    """dator/get_embeds.py:129-136: cv2.resize(depth, (128, 256)) (INTER_LINEAR: half-pixel centres, edge clamp, no
    antialiasing), tile to 3 channels, clip, (d - min) / (max - min), (x - 0.5) / 0.5."""
    d = np.ascontiguousarray(depth_crop, dtype=np.float32)
    h, w = d.shape
    oh, ow = 256, 128

    def coords(o, n):
        f = (np.arange(o, dtype=np.float32) + np.float32(0.5)) * np.float32(n / o) - np.float32(0.5)
        i0 = np.floor(f).astype(np.int64)
        fr = f - i0.astype(np.float32)
        lo = i0 < 0
        hi = i0 >= n - 1
        i0 = np.clip(i0, 0, n - 1)
        i1 = np.clip(i0 + 1, 0, n - 1)
        fr = np.where(lo | hi, np.float32(0), fr).astype(np.float32)
        i1 = np.where(hi, i0, i1)
        return i0, i1, fr

    y0, y1, fy = coords(oh, h)
    x0, x1, fx = coords(ow, w)
    top = d[y0][:, x0] * (1 - fx)[None, :] + d[y0][:, x1] * fx[None, :]
    bot = d[y1][:, x0] * (1 - fx)[None, :] + d[y1][:, x1] * fx[None, :]
    r = (top * (1 - fy)[:, None] + bot * fy[:, None]).astype(np.float32)
    r = np.clip(r, np.float32(dmin), np.float32(dmax))
    r = (r - np.float32(dmin)) / np.float32(dmax - dmin)
    r = (r - np.float32(0.5)) / np.float32(0.5)
    return np.repeat(r[None], 3, axis=0).astype(np.float32)


def embed(rgb_w, depth_w, head_w, rgb_crops, depth_crops):
    """RGB crops (HxWx3 u8) + depth crops (h x w float, metres) -> (n, 128): preprocessing of dator/get_embeds.py:80-87,129-136,
    then build_FourDNet.forward.  Stream weights with the LoRA factors already folded (ibloc_amd.dator.fold_lora)."""
    from ibloc_amd.dator import STREAM_CFG
    rgb = np.stack([preprocess_rgb(c) for c in rgb_crops])
    dep = np.stack([preprocess_depth(c) for c in depth_crops])
    return forward(rgb_w, depth_w, head_w, STREAM_CFG, rgb, dep)

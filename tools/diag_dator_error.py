#!/usr/bin/env python3
"""GPU diagnostic: where DATOR's embedding error comes from, crop by crop.  For u8 crops of the bench generator and one operand-term plan:
the device embedding against the fp32 oracle (torch on the device) AND against the same restatement evaluated in fp64 -- how uncertain
is the fp32 reference itself? -- then the two halves apart: device streams + oracle head, oracle streams + device head.
    python tools/diag_dator_error.py [plan] [n_crops]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ibloc_amd import dator as D  # noqa: E402
from oracle import dator_oracle as do  # noqa: E402
import bench  # noqa: E402

F = torch.nn.functional


def stream(w, cfg, x, dt):
    """the oracle's TransReID stream (oracle/vit_oracle.py:68-104, all tokens, no final norm) in dtype dt"""
    w = {k: v.to(dt) for k, v in w.items()}
    x = x.to(dt)
    B = x.shape[0]
    x = F.conv2d(x, w["patch.w"], w.get("patch.b"), stride=cfg.patch).flatten(2).transpose(1, 2)
    x = torch.cat([w["cls"].reshape(1, 1, -1).expand(B, -1, -1), x], 1) + w["pos"].unsqueeze(0)
    hd = cfg.dim // cfg.heads
    for l in range(cfg.n_blocks_run):
        p = f"l{l}."
        h = F.layer_norm(x, (cfg.dim,), w[p + "ln1.g"], w[p + "ln1.b"], cfg.ln_eps)
        q, k, v = (F.linear(h, w[p + n + ".w"], w[p + n + ".b"]).view(B, -1, cfg.heads, hd).transpose(1, 2) for n in "qkv")
        a = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v
        x = x + F.linear(a.transpose(1, 2).reshape(B, -1, cfg.dim), w[p + "o.w"], w[p + "o.b"])
        h = F.layer_norm(x, (cfg.dim,), w[p + "ln2.g"], w[p + "ln2.b"], cfg.ln_eps)
        x = x + F.linear(F.gelu(F.linear(h, w[p + "fc1.w"], w[p + "fc1.b"])), w[p + "fc2.w"], w[p + "fc2.b"])
    return x


def head(hw, xr, xd, dt):
    """the oracle's fusion head (oracle/dator_oracle.py:34-74) in dtype dt, on the tokens' device"""
    h = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)).to(xr.device).to(dt) for k, v in hw.items()}
    xr, xd = xr.to(dt), xd.to(dt)
    B, N = xr.shape[0], xr.shape[1] - 1
    lin = lambda name, x: F.linear(x, h[name + ".w"], h[name + ".b"])

    def merged(x, side):
        g = lin(f"proj_global_{side}", x[:, 0])
        return lin(f"merge_{side}", torch.cat((g.unsqueeze(1).repeat(1, N, 1), lin(f"proj_local_{side}", x[:, 1:])), -1))
    fr, fd = merged(xr, "rgb"), merged(xd, "depth")
    x = torch.cat((fd.reshape(B, 16, 8, 128).permute(0, 3, 1, 2), fr.reshape(B, 16, 8, 128).permute(0, 3, 1, 2)), dim=1)
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
        samp = F.grid_sample(v.permute(0, 2, 1).reshape(B, 128, 16, 8), grid, align_corners=True).permute(0, 2, 3, 1)
        return lin(op + ".ffn", torch.sum(samp * aw.unsqueeze(-1), dim=-2))
    ln = lambda op, x: F.layer_norm(x, (128,), h[op + ".norm.g"], h[op + ".norm.b"], 1e-5)
    fr = ln("r2r", fr + deform("r2r", q_r, v_r))
    fd = ln("d2d", fd + deform("d2d", q_d, v_d))
    fr = ln("d2r", fr + deform("d2r", q_d, v_r) * rgb_f)
    fd = ln("r2d", fd + deform("r2d", q_r, v_d) * depth_f)
    return torch.mean(fd * depth_f + fr * rgb_f, dim=-2)


def rel(a, b):
    a, b = a.double(), b.double()
    return (torch.linalg.norm((a - b).flatten(1), dim=1) / torch.linalg.norm(b.flatten(1), dim=1)).cpu().numpy()


def main():
    plan = sys.argv[1] if len(sys.argv) > 1 else "default"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 224
    rw, dw, hw = D.random_stream_weights(20), D.random_stream_weights(21), D.random_head_weights(22)
    frw = {k: torch.from_numpy(v).cuda() for k, v in D.fold_lora(rw).items()}
    fdw = {k: torch.from_numpy(v).cuda() for k, v in D.fold_lora(dw).items()}
    crops = bench.Crops("dator", 21)
    rng = np.random.default_rng(3)
    rgb, dep = crops.variants(list(rng.integers(0, 100000, size=n)), rng, "cuda")
    cfg = D.STREAM_CFG
    enc = D.DatorEncoder(rw, dw, hw, precision=None if plan == "default" else plan)
    R = {k: [] for k in ("dev_vs_32", "dev_vs_64", "ref32_vs_64", "tok_dev_r", "tok_dev_d", "tok32_r", "streams_only", "head_only", "head32_vs_64_same_tokens")}
    with torch.no_grad():
        for i in range(0, n, 56):
            pr, img = enc.rgb.preprocess(rgb[i:i + 56], want_u8=True)
            pd = enc.preprocess_depth(dep[i:i + 56])
            rt, dt = enc.rgb.forward_patches(pr).clone(), enc.depth.forward_patches(pd).clone()
            emb = enc.head(rt, dt).clone()
            mean = torch.tensor(enc.rgb.recipe.mean, dtype=torch.float32, device="cuda")
            std = torch.tensor(enc.rgb.recipe.std, dtype=torch.float32, device="cuda")
            xr = (((img.to(torch.float64) * (1 / 255)).to(torch.float32) - mean) / std).permute(0, 3, 1, 2).contiguous()
            xd = torch.from_numpy(np.stack([do.preprocess_depth(d) for d in dep[i:i + 56].cpu().numpy()])).cuda()
            t32r, t32d = stream(frw, cfg, xr, torch.float32), stream(fdw, cfg, xd, torch.float32)
            t64r, t64d = stream(frw, cfg, xr, torch.float64), stream(fdw, cfg, xd, torch.float64)
            e32, e64 = head(hw, t32r, t32d, torch.float32), head(hw, t64r, t64d, torch.float64)
            R["dev_vs_32"].append(rel(emb, e32))
            R["dev_vs_64"].append(rel(emb, e64))
            R["ref32_vs_64"].append(rel(e32, e64))
            R["tok_dev_r"].append(rel(rt, t64r))
            R["tok_dev_d"].append(rel(dt, t64d))
            R["tok32_r"].append(rel(t32r, t64r))
            R["streams_only"].append(rel(head(hw, rt, dt, torch.float64), e64))                 # device tokens through the fp64 head
            R["head_only"].append(rel(enc.head(t32r.contiguous(), t32d.contiguous()), head(hw, t32r, t32d, torch.float64)))   # fp32 oracle tokens through the device head
            R["head32_vs_64_same_tokens"].append(rel(head(hw, t32r, t32d, torch.float32), head(hw, t32r, t32d, torch.float64)))
    R = {k: np.concatenate(v) for k, v in R.items()}
    print(f"plan {enc.rgb.precision}, {n} crops")
    for k, v in R.items():
        print(f"  {k:26s} mean {v.mean():.3e}  p90 {np.percentile(v, 90):.3e}  max {v.max():.3e}")
    worst = np.argsort(-R["dev_vs_32"])[:5]
    print("  worst crops (dev_vs_32):", [(int(j), *(f"{R[k][j]:.2e}" for k in ("dev_vs_32", "dev_vs_64", "ref32_vs_64", "tok_dev_r", "tok_dev_d", "streams_only", "head_only"))) for j in worst])


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GPU: embedding error (rel-L2 vs the fp32 oracle evaluated by torch on the same device) and encoder time of ViT-B/14 for a list of
operand-term plans (ibloc_amd.vit.DEFAULT_PRECISION syntax).  python tools/perf_vit_precision.py [plan ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ibloc_amd import vit as V  # noqa: E402
from oracle import vit_oracle as vo  # noqa: E402
import bench  # noqa: E402


def oracle_input(u8, recipe):
    mean = torch.tensor(recipe.mean, dtype=torch.float32, device=u8.device)
    std = torch.tensor(recipe.std, dtype=torch.float32, device=u8.device)
    x = (u8.to(torch.float64) * (1 / 255)).to(torch.float32)
    return ((x - mean) / std).permute(0, 3, 1, 2).contiguous()


def main():
    plans = sys.argv[1:] or ["plain", V.DEFAULT_PRECISION]
    cfg = V.CONFIGS["dinov2_vitb14"]
    w = V.random_weights(cfg, 20)
    wt = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)).cuda() for k, v in w.items()}
    crops = bench.Crops("dinov2_vitb14", 21)
    rng = np.random.default_rng(3)
    n = int(os.environ.get("N_CROPS", "1792"))
    u8 = crops.variants(list(rng.integers(0, 100000, size=n)), rng, "cuda")
    ref = None
    for plan in plans:
        enc = V.VitEncoder(cfg, w, precision=plan)
        outs, refs = [], []
        for i in range(0, n, 224):
            patches, img = enc.preprocess(u8[i:i + 224], want_u8=True)
            outs.append(enc.forward_patches(patches).clone())
            if ref is None:
                refs.append(torch.from_numpy(vo.vit_forward(wt, cfg, oracle_input(img, enc.recipe), device="cuda")).cuda())
        if ref is None:
            ref = torch.cat(refs)
        out = torch.cat(outs)
        rel = (torch.linalg.norm(out - ref, dim=1) / torch.linalg.norm(ref, dim=1)).cpu().numpy()
        patches = enc.preprocess(u8[:224])
        for _ in range(3):
            enc.forward_patches(patches)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            enc.forward_patches(patches)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100
        print(f"{plan:28s} rel-L2 mean {rel.mean():.3e} max {rel.max():.3e}   forward(224 crops) {ms:.2f} ms", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Quick timing of the ViT forward / preprocess on the GPU box (not the bench)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from ibloc_amd import vit as V

name = sys.argv[1] if len(sys.argv) > 1 else "dinov2_vitb14"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 448
cfg = V.CONFIGS[name]
enc = V.VitEncoder(cfg, V.random_weights(cfg, 0))
rng = np.random.default_rng(0)
crops = torch.from_numpy(rng.integers(0, 256, size=(B, 224, 224, 3), dtype=np.uint8)).cuda()
for _ in range(2):
    p = enc.preprocess(crops)
    o = enc.forward_patches(p)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
n = 5
e[0].record()
for _ in range(n):
    p = enc.preprocess(crops)
e[1].record()
for _ in range(n):
    o = enc.forward_patches(p)
e[2].record()
torch.cuda.synchronize()
tp, tf = e[0].elapsed_time(e[1]) / n, e[1].elapsed_time(e[2]) / n
flops = {"dinov2_vitb14": 46.32e9, "dinov2_vits14": 12.25e9, "vit_b16": 35.13e9, "clip_b32": 8.82e9}.get(name, 0) * B
print(f"{name} B={B}: preprocess {tp:.3f} ms, forward {tf:.3f} ms -> {flops / tf / 1e9:.1f} TFLOP/s, "
      f"{B / (tp + tf) * 1e3:.0f} crops/s")

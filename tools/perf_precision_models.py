#!/usr/bin/env python3
"""GPU: embedding error (rel-L2 per crop vs the fp32 oracle evaluated by torch on the same device) and encoder time for ViT-B/16,
CLIP ViT-B/32, DINOv2-S/14 and DATOR (two TransReID streams + head) under a list of operand-term plans
(ibloc_amd.vit.DEFAULT_PRECISION syntax), on u8 crops of the bench generator.
    python tools/perf_precision_models.py [model ...] -- [plan ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ibloc_amd import dator as D  # noqa: E402
from ibloc_amd import vit as V  # noqa: E402
from oracle import dator_oracle as do  # noqa: E402
from oracle import vit_oracle as vo  # noqa: E402
import bench  # noqa: E402


def oracle_input(u8, recipe):
    mean = torch.tensor(recipe.mean, dtype=torch.float32, device=u8.device)
    std = torch.tensor(recipe.std, dtype=torch.float32, device=u8.device)
    x = (u8.to(torch.float64) * (1 / 255)).to(torch.float32)
    return ((x - mean) / std).permute(0, 3, 1, 2).contiguous()


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def run_vit(name, plans, n):
    cfg = V.CONFIGS[name]
    w = V.random_weights(cfg, 20)
    wt = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)).cuda() for k, v in w.items()}
    crops = bench.Crops(name, 21)
    rng = np.random.default_rng(3)
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
        ms = timed(lambda: enc.forward_patches(patches))
        print(f"{name:14s} {plan:32s} rel-L2 mean {rel.mean():.3e} max {rel.max():.3e}   forward(224 crops) {ms:.2f} ms", flush=True)


def run_dator(plans, n):
    rw, dw, hw = D.random_stream_weights(20), D.random_stream_weights(21), D.random_head_weights(22)
    frw = {k: torch.from_numpy(v).cuda() for k, v in D.fold_lora(rw).items()}
    fdw = {k: torch.from_numpy(v).cuda() for k, v in D.fold_lora(dw).items()}
    crops = bench.Crops("dator", 21)
    rng = np.random.default_rng(3)
    rgb, dep = crops.variants(list(rng.integers(0, 100000, size=n)), rng, "cuda")
    cfg = D.STREAM_CFG
    ref = None
    for plan in plans:
        os.environ["IBL_VIT_PREC"] = plan
        enc = D.DatorEncoder(rw, dw, hw)
        outs, refs = [], []
        for i in range(0, n, 112):
            pr, img = enc.rgb.preprocess(rgb[i:i + 112], want_u8=True)
            pd = enc.preprocess_depth(dep[i:i + 112])
            rt, dt = enc.rgb.forward_patches(pr), enc.depth.forward_patches(pd)
            outs.append(enc.head(rt, dt).clone())
            if ref is None:
                # oracle: fp32 streams on the device from the same u8 image / the fp16 depth patches' own pixel values are NOT used:
                # the depth model input is recomputed in fp32 by the oracle's restatement (host), the streams by torch on the device
                dpx = np.stack([do.preprocess_depth(d) for d in dep[i:i + 112].cpu().numpy()])
                ot_r = vo.vit_forward(frw, cfg, oracle_input(img, enc.rgb.recipe), all_tokens=True, device="cuda")
                ot_d = vo.vit_forward(fdw, cfg, torch.from_numpy(dpx), all_tokens=True, device="cuda")
                refs.append(torch.from_numpy(do.head_forward(hw, ot_r, ot_d)).cuda())
        if ref is None:
            ref = torch.cat(refs)
        out = torch.cat(outs)
        rel = (torch.linalg.norm(out - ref, dim=1) / torch.linalg.norm(ref, dim=1)).cpu().numpy()
        tot = float(torch.linalg.norm(out - ref) / torch.linalg.norm(ref))
        ms = timed(lambda: enc.embed(rgb[:224], dep[:224]))
        print(f"{'dator':14s} {plan:32s} rel-L2 mean {rel.mean():.3e} max {rel.max():.3e} whole {tot:.3e}   embed(224 crops) {ms:.2f} ms", flush=True)
    os.environ.pop("IBL_VIT_PREC", None)


def main():
    argv = sys.argv[1:]
    if "--" in argv:
        k = argv.index("--")
        models, plans = argv[:k], argv[k + 1:]
    else:
        models, plans = argv, []
    models = models or ["vit_b16", "clip_b32", "dinov2_vits14", "dator"]
    plans = plans or ["plain", V.DEFAULT_PRECISION, "p2;0:3222;1:2222", "p2;0:3222;1:3222;2:2211", "p2;0:3222;1:3222;2:2222;3:2211"]
    n = int(os.environ.get("N_CROPS", "896"))
    for m in models:
        if m == "dator":
            run_dator(plans, min(n, 448))
        else:
            run_vit(m, plans, n)


if __name__ == "__main__":
    main()

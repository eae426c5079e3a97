#!/usr/bin/env python3
"""Randomised cross-check of the registration fast paths on the GPU: for several seeded scenes the outputs of
(instance features + matrix-core search) must equal, bit for bit, those of (no instance features + VALU search)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibloc_amd.registration import CloudBatch, RegContext, instance_features_batch, radius_outlier_batch, register_batch
from ibloc_amd.synth import SynthWorld
from ibloc_amd.engine import intensity_from_colors

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = RegContext(12 << 30)
bad = 0
for seed in range(n_scenes):
    rng = np.random.default_rng(1000 + seed)
    spacing = float(rng.choice([1.2, 1.8, 2.5]))
    pts = int(rng.choice([1500, 3000, 5000]))
    w = SynthWorld(16, pts_per_object=pts, E=1, D=8, seed=seed, spacing=spacing)
    frames = [w.make_frame(rng, q=int(rng.integers(2, 6)), pts_per_object=pts) for _ in range(3)]
    clouds = [c[0] for f in frames for c in f["clouds"]]
    ints = [intensity_from_colors(c[1]) for f in frames for c in f["clouds"]]
    det0 = CloudBatch.from_numpy(clouds, ints)
    if seed % 2:                                   # outlier-cleaned (sparse, noisy) detections every other scene
        keep = radius_outlier_batch(ctx, det0, 0.05, 8).cpu().numpy().astype(bool)
        off = det0.seg_off_host
        clouds = [c[keep[off[i]:off[i + 1]]] for i, c in enumerate(clouds)]
        ints = [c[keep[off[i]:off[i + 1]]] for i, c in enumerate(ints)]
    det = CloudBatch.from_numpy(clouds, ints)
    mem = CloudBatch.from_numpy(w.points, [intensity_from_colors(c) for c in w.colors])
    js, jt, base = [], [], 0
    for f in frames:
        ids, q = f["ids"], len(f["ids"])
        for _ in range(5):
            n = int(rng.integers(1, min(3, q) + 1))
            ds = sorted(rng.choice(q, size=n, replace=False).tolist())
            ms = [ids[d] if rng.random() < 0.7 else int(rng.integers(0, w.M)) for d in ds]
            if len(set(ms)) < n:
                continue
            js.append([base + d for d in ds] + [-1] * (3 - n))
            jt.append(ms + [-1] * (3 - n))
        base += q
    fd = instance_features_batch(ctx, det, 0.05)
    fm = instance_features_batch(ctx, mem, 0.05, grad_radius=0.15)
    fast = register_batch(ctx, det, mem, js, jt, 0.05, 1.5, 1.5, seed=seed, job_id_base=3, det_features=fd, mem_features=fm)
    os.environ["IBL_FEAT_VALU"] = "1"
    try:
        slow = register_batch(ctx, det, mem, js, jt, 0.05, 1.5, 1.5, seed=seed, job_id_base=3)
    finally:
        os.environ.pop("IBL_FEAT_VALU", None)
    same = all(np.array_equal(fast[k], slow[k]) for k in ("T", "rmse", "fitness", "T_ransac", "ransac_stats", "means"))
    print(f"scene {seed}: spacing {spacing} pts {pts} jobs {len(js)} reuse {fast['reuse'].tolist()} -> {'identical' if same else 'DIFFERENT'}")
    bad += 0 if same else 1
print("all identical" if bad == 0 else f"{bad} scenes differ")
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""GPU: randomised cross-check of the fp32-guard-band SPFH path (pair_bins_f32 + fp64 queue, csrc/reg_knn.hip) against the all-fp64 evaluation
(IBL_SPFH_F64=1) on clouds of many shapes: the FPFH rows (and normals) must be identical, bit for bit.  python tools/stress_spfh.py [rounds]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibloc_amd.registration import CloudBatch, RegContext, instance_features_batch  # noqa: E402


def shapes(rng):
    n = int(rng.integers(200, 6000))
    kind = rng.integers(0, 7)
    off = rng.uniform(-250, 250, size=3) * rng.choice([0.0, 1.0])
    if kind == 0:      # noisy plane
        p = np.concatenate([rng.uniform(-0.3, 0.3, size=(n, 2)), rng.normal(0, rng.choice([0, 1e-4, 3e-3]), size=(n, 1))], 1)
    elif kind == 1:    # sphere shell
        v = rng.normal(size=(n, 3))
        p = v / np.linalg.norm(v, axis=1, keepdims=True) * rng.uniform(0.1, 0.4)
    elif kind == 2:    # cylinder
        t = rng.uniform(0, 2 * np.pi, n)
        p = np.stack([0.15 * np.cos(t), 0.15 * np.sin(t), rng.uniform(-0.3, 0.3, n)], 1)
    elif kind == 3:    # lattice (exact ties, equal normals)
        g = np.stack(np.meshgrid(*[np.arange(int(round(n ** (1 / 3))) + 1)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n]
        p = g * rng.choice([0.01, 0.02, 0.03125])
    elif kind == 4:    # line + noise (d parallel to everything)
        p = np.outer(rng.uniform(-0.4, 0.4, n), rng.normal(size=3)) + rng.normal(0, 1e-3, size=(n, 3))
    elif kind == 5:    # volume noise
        p = rng.uniform(-0.12, 0.12, size=(n, 3))
    else:              # two touching boxes' faces
        p = rng.uniform(-0.2, 0.2, size=(n, 3))
        p[np.arange(n), rng.integers(0, 3, n)] = rng.choice([-0.2, 0.2], n)
    return (p + off).astype(np.float32)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ctx = RegContext(8 << 30)
    bad = 0
    pts = 0
    for r in range(rounds):
        rng = np.random.default_rng(5000 + r)
        b = CloudBatch.from_numpy([shapes(rng) for _ in range(24)])
        os.environ.pop("IBL_SPFH_F64", None)
        a = instance_features_batch(ctx, b, 0.05)
        os.environ["IBL_SPFH_F64"] = "1"
        c = instance_features_batch(ctx, b, 0.05)
        torch.cuda.synchronize()
        same = torch.equal(a.fpfh[:b.n], c.fpfh[:b.n]) and torch.equal(a.normals[:b.n], c.normals[:b.n])
        pts += b.n
        if not same:
            bad += 1
            d = (a.fpfh[:b.n] != c.fpfh[:b.n]).any(1).sum().item()
            print(f"round {r}: {d} of {b.n} FPFH rows differ")
    os.environ.pop("IBL_SPFH_F64", None)
    print(f"{rounds} rounds, {pts} points: {'all identical' if not bad else str(bad) + ' rounds differ'} (status {ctx.status()})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

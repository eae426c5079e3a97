#!/usr/bin/env python3
"""Times ibl_instance_features_batch (normals + SPFH + FPFH [+ colour gradients]) on the detections of one bench step: 224 clouds of
~5 000 points (synth.SynthWorld frames after the radius-outlier removal).  IBLOC_LIB=path selects another build of the library
(tools/_lab variants).  Prints ms per call."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ibloc_amd.engine import intensity_from_colors  # noqa: E402
from ibloc_amd.registration import CloudBatch, RegContext, instance_features_batch, radius_outlier_batch  # noqa: E402
from ibloc_amd.synth import SynthWorld  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    pts = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    grad = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    w = SynthWorld(64, pts_per_object=pts, E=1, D=8, seed=21)
    rng = np.random.default_rng(5)
    clouds, ints = [], []
    for _ in range(frames):
        f = w.make_frame(rng, q=7, pts_per_object=pts)
        for p, c in f["clouds"]:
            clouds.append(p)
            ints.append(intensity_from_colors(c))
    det = CloudBatch.from_numpy(clouds, ints)
    ctx = RegContext(12 << 30)
    keep = radius_outlier_batch(ctx, det, 0.05, 8).bool()
    off = torch.cat([torch.zeros(1, dtype=torch.int32, device="cuda"), torch.cumsum(keep.to(torch.int32), 0)])[det.seg_off.long()]
    clean = CloudBatch(det.pts4[keep].contiguous(), off.cpu().numpy().astype(np.int32))
    for _ in range(2):
        instance_features_batch(ctx, clean, 0.05, grad)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        instance_features_batch(ctx, clean, 0.05, grad)
    torch.cuda.synchronize()
    print(f"{os.environ.get('IBLOC_LIB', 'default'):40s} {clean.n} points in {clean.n_seg} clouds: {(time.perf_counter() - t0) / n * 1e3:7.2f} ms per call")


if __name__ == "__main__":
    main()

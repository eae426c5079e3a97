#!/usr/bin/env python3
"""Times stage B of one bench step (outlier removal, detection features, register, evaluate) on synthetic frames, without the encoder:
the detections' embeddings come from the generator.  IBL_TIMING=1 prints the host-synchronised phases of ibl_register_batch_cached;
IBLOC_LIB=path selects a lab build; IBL_COMPACT=1 keeps the memory's instance features without their fp16 operand rows.  usage: perf_register.py [frames] [memory]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ibloc_amd.engine import LocaliseEngine, MemoryShard, intensity_from_colors  # noqa: E402
from ibloc_amd.registration import CloudBatch, RegContext  # noqa: E402
from ibloc_amd.synth import SynthWorld  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    w = SynthWorld(M, pts_per_object=5000, E=4, D=64, seed=21)
    ctx = RegContext(16 << 30)
    eng = LocaliseEngine(MemoryShard(ctx, list(w.embeddings), w.points, colors=w.colors, compact_features=bool(int(os.environ.get("IBL_COMPACT", "0")))))
    rng = np.random.default_rng(5)
    batches = []
    for _ in range(4):
        clouds, ints, embs, qs = [], [], [], []
        for _ in range(frames):
            f = w.make_frame(rng, q=7, pts_per_object=5000)
            for p, c in f["clouds"]:
                clouds.append(p)
                ints.append(intensity_from_colors(c))
            embs.append(f["det_emb"])
            qs.append(len(f["clouds"]))
        batches.append((CloudBatch.from_numpy(clouds, ints), qs, np.concatenate(embs)))
    kw = dict(fpfh_voxel_size=0.05, fpfh_global_dist_factor=1.5, fpfh_local_dist_factor=1.5, seed=7)
    eng.localise_batch(batches[0][0], batches[0][1], det_emb=batches[0][2], **kw)
    torch.cuda.synchronize()
    tm = {}
    t0 = time.perf_counter()
    for det, qs, emb in batches[1:]:
        eng.localise_batch(det, qs, det_emb=emb, timings=tm, **kw)
    torch.cuda.synchronize()
    n = len(batches) - 1
    print(f"{os.environ.get('IBLOC_LIB', 'default'):36s} {(time.perf_counter() - t0) / n * 1e3:7.2f} ms per step;",
          {k: round(v / n, 2) for k, v in tm.items() if isinstance(v, float)})


if __name__ == "__main__":
    main()

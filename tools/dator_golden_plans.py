import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from ibloc_amd import dator as D
GOLD = np.load('/root/repo/tests/golden/dator_golden.npz')
rw, dw, hw = D.random_stream_weights(301), D.random_stream_weights(302), D.random_head_weights(303)
rng = np.random.default_rng(304)
rgb = rng.normal(size=(3, 3, 256, 128)).astype(np.float32)
depth = np.repeat(rng.uniform(-1, 1, size=(3, 1, 256, 128)).astype(np.float32), 3, axis=1)
ref = GOLD["embedding"]
for plan in sys.argv[1:]:
    e = D.DatorEncoder(rw, dw, hw, precision=plan)
    got = e.forward_pixels(torch.from_numpy(rgb), torch.from_numpy(depth)).cpu().numpy()
    rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    per = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
    print(plan, "whole", rel, "per crop", per)

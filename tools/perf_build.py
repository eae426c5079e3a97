#!/usr/bin/env python3
"""Timing of the memory-build kernels beside their CPU restatements on a bounded sample.

voxel down-sampling: M objects x P surface points (fp64 xyz + rgb resident in HBM), voxel 0.005 as the driver uses
(tum_localisation_trial.py:139); algorithmic traffic = 48 B read per input point + 48 B written per output voxel.
DBSCAN: one group of N points of surface density, eps 0.05 / min_points 50 (the driver's :148)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ibloc_amd import _lib
from ibloc_amd.build import dbscan_batch, voxel_downsample_batch
from ibloc_amd.registration import RegContext, _stream
from oracle import build_oracle as bo

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
rng = np.random.default_rng(0)
ctx = RegContext(16 << 30)


def surface(n, c):
    u = rng.uniform(-1, 1, size=(n, 3))
    ax = rng.integers(0, 3, size=n)
    u[np.arange(n), ax] = np.sign(u[np.arange(n), ax])
    return c + u * np.array([0.25, 0.2, 0.3])


pts = [surface(P, rng.uniform(-20, 20, size=3)) for _ in range(M)]
cols = [rng.uniform(size=(P, 3)) for _ in range(M)]
n = M * P
# device-resident timing through the C-ABI (the python wrapper above also pays the host <-> device copies)
off = np.arange(M + 1, dtype=np.int32) * P
dP = torch.from_numpy(np.concatenate(pts)).cuda()
dC = torch.from_numpy(np.concatenate(cols)).cuda()
oP, oC = torch.empty_like(dP), torch.empty_like(dC)
out_off = np.zeros(M + 1, dtype=np.int32)
for voxel in (0.005, 0.02):
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(_lib.lib.ibl_voxel_downsample_batch(ctx.handle, dP.data_ptr(), dC.data_ptr(), off.ctypes.data, M, voxel, oP.data_ptr(), oC.data_ptr(), None,
                                                       out_off.ctypes.data, _stream()), "vox")
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    dt = min(ts)
    nv = int(out_off[-1])
    print(f"voxel {voxel}: {n / 1e6:.1f} M points -> {nv / 1e6:.2f} M voxels in {dt * 1e3:.1f} ms = {n / dt / 1e6:.0f} M points/s, "
          f"algorithmic {(48 * n + 48 * nv) / dt / 1e9:.0f} GB/s (HBM peak 8000)")
    k = max(1, min(M, 200000 // P))
    t0 = time.perf_counter()
    for i in range(k):
        bo.voxel_down_sample_with_colors(pts[i], cols[i], voxel)
    cdt = time.perf_counter() - t0
    print(f"    CPU restatement (python dict, 1 core): {k * P / cdt / 1e6:.3f} M points/s on {k} objects ({cdt:.1f} s) -> device is {n / dt / (k * P / cdt):.0f}x")
# DBSCAN
N = 2000000
S = np.concatenate([surface(N // 40, rng.uniform(-8, 8, size=3)) for _ in range(40)])
dS = torch.from_numpy(S).cuda()
lab = torch.empty(len(S), dtype=torch.int32, device="cuda")
goff = np.array([0, len(S)], dtype=np.int32)
ncl = np.zeros(1, dtype=np.int32)
for eps, mp in ((0.05, 50), (0.02, 10)):
    ts = []
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(_lib.lib.ibl_dbscan_batch(ctx.handle, dS.data_ptr(), goff.ctypes.data, 1, eps, mp, lab.data_ptr(), ncl.ctypes.data, _stream()), "dbscan")
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"dbscan eps {eps} min_points {mp}: {len(S) / 1e6:.1f} M points, {int(ncl[0])} clusters, noise {(lab == -1).float().mean().item():.3f}: {min(ts) * 1e3:.0f} ms = {len(S) / min(ts) / 1e6:.1f} M points/s")
from sklearn.cluster import DBSCAN
sub = S[:60000]
t0 = time.perf_counter()
want = DBSCAN(eps=0.05, min_samples=50).fit(sub).labels_
cdt = time.perf_counter() - t0
got, _ = dbscan_batch(ctx, [sub], 0.05, 50)
print(f"    CPU scikit-learn DBSCAN (kd-tree, 1 core) on 60 k points: {cdt:.1f} s = {len(sub) / cdt / 1e6:.3f} M points/s; labels equal: {np.array_equal(got[0], want)}")

import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from ibloc_amd.engine import LocaliseEngine, MemoryShard, intensity_from_colors
from ibloc_amd.registration import CloudBatch, RegContext
from ibloc_amd.synth import SynthWorld
from oracle import reg_oracle as ro
world = SynthWorld(6, pts_per_object=2000, E=2, D=64, seed=3)
f = world.make_frame(np.random.default_rng(4), q=2, pts_per_object=2000, anchor=1)
ctx = RegContext(2 << 30)
eng = LocaliseEngine(MemoryShard(ctx, list(world.embeddings), world.points, colors=world.colors))
det = CloudBatch.from_numpy([c[0] for c in f["clouds"]], [intensity_from_colors(c[1]) for c in f["clouds"]])
res = eng.localise_batch(det, [2], det_emb=f["det_emb"], fpfh_global_dist_factor=1.5, fpfh_local_dist_factor=1.5, seed=9)[0]
cleaned, ccols = [], []
for (p, c) in f["clouds"]:
    k = ro.radius_outlier(p.astype(np.float32), 0.05, 8)
    cleaned.append(p[k]); ccols.append(c[k])
pose, recs, best = ro.localise_from_assignments(cleaned, ccols, world.points, world.colors, res.assignments, 0.05, 1.5, 1.5, seed=9)
print("assns", res.assignments, "best gpu", res.best, "oracle", best)
for a, b in zip(res.records, recs):
    print("gpu fit %.4f full %.5f rstats %s | oracle fit %.4f full %.5f | dT %.2e  dTr %.2e" % (a["fitness"], a["full_fitness"], a["ransac_stats"], b["fitness"], b["full_fitness"], np.abs(a["T"]-b["T"]).max(), 0))
print(res.pose, pose)

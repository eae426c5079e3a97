"""-m gpu: registration with the memory clouds sharded by instance range (routing.py) against the unsharded engine on the same frames:
two ranks on this box's one GPU (gloo transport with host staging -- RCCL refuses two ranks on one device; on a node the same code runs
over "nccl"), every rank localises its own frames, jobs travel to the owner of their targets or fetch the instances they miss, and the
poses, transforms and selected assignments must be bit-identical (object_memory.py:1020-1106)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["IBL_ROOT"])
import numpy as np, torch, torch.distributed as dist
from ibloc_amd.engine import LocaliseEngine, MemoryShard, intensity_from_colors
from ibloc_amd.registration import CloudBatch, RegContext
from ibloc_amd.synth import SynthWorld
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
w = SynthWorld(12, pts_per_object=2500, E=2, D=32, seed=61, spacing=1.6)            # the same memory on every rank
rng = np.random.default_rng(900 + rank)                                             # its own frames
frames = [w.make_frame(rng, q=3, pts_per_object=2500, anchor=int(a)) for a in rng.integers(0, 12, size=3)]
clouds, ints, embs, qs = [], [], [], []
for f in frames:
    for (p, c) in f["clouds"]:
        clouds.append(p); ints.append(intensity_from_colors(c))
    embs.append(f["det_emb"]); qs.append(len(f["clouds"]))
kw = dict(det_emb=np.concatenate(embs), fpfh_voxel_size=0.05, fpfh_global_dist_factor=1.5, fpfh_local_dist_factor=1.5, seed=5,
          job_id_base=100 * rank)

ctx = RegContext(3 << 30)
ref_mem = MemoryShard(ctx, list(w.embeddings), w.points, colors=w.colors)
ref = LocaliseEngine(ref_mem).localise_batch(CloudBatch.from_numpy(clouds, ints), qs, **kw)
ref_mem.close()

ctx2 = RegContext(3 << 30)
mem = MemoryShard(ctx2, list(w.embeddings), w.points, colors=w.colors, shard=(rank, world), shard_clouds=True)
assert mem.clouds.n_seg == mem.hi - mem.lo
eng = LocaliseEngine(mem)
res = eng.localise_batch(CloudBatch.from_numpy(clouds, ints), qs, **kw)
n_jobs = 0
for a, b in zip(res, ref):
    assert a.assignments == b.assignments and a.best == b.best and a.n_clean == b.n_clean
    assert np.array_equal(a.pose, b.pose) and np.array_equal(a.pose_corrected, b.pose_corrected)
    for ra, rb in zip(a.records, b.records):
        n_jobs += 1
        assert np.array_equal(ra["T"], rb["T"]) and ra["fitness"] == rb["fitness"] and ra["rmse"] == rb["rmse"]
        assert np.array_equal(ra["ransac_stats"], rb["ransac_stats"]) and np.array_equal(ra["T_global"], rb["T_global"])
        assert ra["full_fitness"] == rb["full_fitness"] and abs(ra["full_rmse"] - rb["full_rmse"]) <= 1e-9 * max(1.0, rb["full_rmse"])
st = eng.route_stats
print(f"rank {rank}: {n_jobs} jobs bit-identical; shipped {st['jobs_shipped']}, instances fetched {st['instances_fetched']}, "
      f"bytes sent {st['bytes_sent']}", flush=True)
# ranks that disagree on a registration parameter (a shipped job would silently run with its executor's seed): EVERY rank must raise,
# none may be left waiting in a collective (ADVICE r2)
raised = False
try:
    eng.localise_batch(CloudBatch.from_numpy(clouds, ints), qs, **dict(kw, seed=5 + rank))
except ValueError as e:
    raised = "different registration parameters" in str(e)
assert raised, "ranks with different seeds must raise on every rank"
tot = [None] * world
dist.all_gather_object(tot, (st["jobs_shipped"], st["instances_fetched"], n_jobs))
dist.barrier()
if rank == 0:
    assert sum(t[0] for t in tot) > 0 and sum(t[1] for t in tot) > 0 and sum(t[2] for t in tot) > 0, tot      # both routes ran
    print("GPU_ROUTING_OK", tot)
'''


def test_sharded_cloud_registration_is_bit_identical(tmp_path):
    script = tmp_path / "gpu_routing_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, IBL_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", str(script)], env=env, capture_output=True, text=True, timeout=800)
    print(out.stdout[-3000:])
    assert out.returncode == 0 and "GPU_ROUTING_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-4000:]

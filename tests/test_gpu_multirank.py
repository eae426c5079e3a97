"""-m gpu, needs >= 2 visible devices (skipped on a one-GPU box): the sharded layouts over REAL RCCL -- the first thing a multi-GPU lease
should run, before `bench.py --gpus N` does (VERDICT r3 #6a).  Two child ranks (fresh processes, one per device, started before this
process makes any HIP call of its own) check, on the library's communicator (`ibl_comm_*`, parallel.RcclComm):
  1. `ibl_alltoall` / `ibl_allgather_topk` / `ibl_allreduce_max_i32` / `ibl_allreduce_min` against known patterns and against
     torch.distributed's own nccl collectives on the same buffers,
  2. `ShardExchange.agree`: a rank that reports an error / runs out of batches makes every rank see it,
  3. stage A with the embedding memory sharded by instance range (all-gather of query rows, per-shard top-k, all-to-all of the candidate
     lists, assignment search at the owner): assignment lists identical to the unsharded engine, frame by frame,
  4. the routed registration with sharded clouds (routing.py): poses, transforms and winners bit-identical to the unsharded engine
     (the test of tests/test_gpu_routing.py, here with one device per rank and RCCL instead of gloo with host staging)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["IBL_ROOT"])
import numpy as np, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
REHEARSE = os.environ.get("IBL_MULTIRANK_REHEARSE", "") == "1"      # one-GPU box: both ranks on device 0 over gloo, no RCCL communicator
local = 0 if REHEARSE else int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("gloo" if REHEARSE else "nccl")
from ibloc_amd.engine import LocaliseEngine, MemoryShard, intensity_from_colors
from ibloc_amd.parallel import RcclComm, ShardExchange
from ibloc_amd.registration import CloudBatch, RegContext
from ibloc_amd.synth import SynthWorld

comm = None if REHEARSE else RcclComm()
assert REHEARSE or (comm.world == world and comm.rank == rank)
# 1. collectives against patterns and against torch's nccl group
n = 1 << 16
if not REHEARSE:
  if True:
    mine = (torch.arange(world * n, device=dev, dtype=torch.int32) + 1000003 * rank).contiguous()
    got = torch.empty_like(mine)
    comm.all_to_all(got, mine)
    want = torch.empty_like(mine)
    dist.all_to_all_single(want, mine)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    for r in range(world):                       # block r came from rank r and is that rank's block `rank`
        assert torch.equal(got[r * n:(r + 1) * n], torch.arange(rank * n, (rank + 1) * n, device=dev, dtype=torch.int32) + 1000003 * r)
    ag = torch.empty(world * n, device=dev, dtype=torch.int32)
    comm.all_gather(ag, mine[:n].contiguous())
    torch.cuda.synchronize()
    for r in range(world):
        assert torch.equal(ag[r * n:(r + 1) * n], torch.arange(n, device=dev, dtype=torch.int32) + 1000003 * r)
    mx = torch.tensor([rank, 7 - rank, 3], device=dev, dtype=torch.int32)
    comm.all_reduce_max_i32(mx)
    mn = torch.tensor([float(rank), float("inf") if rank else 2.5, 1.0 + rank], device=dev)
    comm.all_reduce_min(mn)
    torch.cuda.synchronize()
    assert mx.tolist() == [world - 1, 7, 3] and mn.tolist() == [0.0, 2.5, 1.0]
# 2. agree: one rank's error / end of input reaches every rank
ex = ShardExchange(rows_cap=32, comm=comm)
assert ex.agree(False, False, dev) == (False, False, True)
assert ex.agree(rank == 1, False, dev) == (True, False, True)
assert ex.agree(False, rank == 0, dev) == (False, True, True)
assert ex.any_flag(rank == world - 1, dev) is True and ex.any_flag(False, dev) is False

# 3. stage A with a sharded embedding memory == the unsharded engine (assignment lists, frame by frame)
w = SynthWorld(60, pts_per_object=16, E=3, D=64, seed=71, sample_points=False)
rng = np.random.default_rng(800 + rank)
frames = [w.make_frame(rng, q=int(q), pts_per_object=16, with_clouds=False) for q in rng.integers(1, 8, size=9)]
qs = [len(f["ids"]) for f in frames]
emb = np.concatenate([f["det_emb"] for f in frames])
ctx = RegContext(256 << 20)
ref = LocaliseEngine(MemoryShard(ctx, list(w.embeddings), device=str(dev))).localise_batch(None, qs, det_emb=emb, register=False)
ctx2 = RegContext(256 << 20)
eng = LocaliseEngine(MemoryShard(ctx2, list(w.embeddings), device=str(dev), shard=(rank, world)), rows_cap=64, comm=comm)
assert eng.exchange is not None and eng.exchange.comm is comm and eng.exchange.world == world
for rep in range(3):                         # several steps: the collectives of consecutive steps must pair up
    res = eng.localise_batch(None, qs, det_emb=emb, register=False)
    assert [r.assignments for r in res] == [r.assignments for r in ref], rank
# the pipelined form issues its collectives from the stage-A thread
items = [dict(det=None, q_per_frame=qs, det_emb=emb) for _ in range(4)]
for res in eng.localise_stream(items, register=False):
    assert [r.assignments for r in res] == [r.assignments for r in ref], rank
print(f"rank {rank}: sharded stage A == unsharded on {len(frames)} frames x 7 steps", flush=True)

# 4. routed registration with sharded clouds == the unsharded engine, bit for bit (the one-GPU rehearsal of this part is
#    tests/test_gpu_routing.py)
if REHEARSE:
    dist.barrier()
    if rank == 0:
        print("GPU_MULTIRANK_OK (rehearsal: gloo, both ranks on device 0)", flush=True)
    dist.destroy_process_group()
    sys.exit(0)
w2 = SynthWorld(12, pts_per_object=2500, E=2, D=32, seed=61, spacing=1.6)
rng = np.random.default_rng(900 + rank)
frames = [w2.make_frame(rng, q=3, pts_per_object=2500, anchor=int(a)) for a in rng.integers(0, 12, size=3)]
clouds, ints, embs, qs = [], [], [], []
for f in frames:
    for (p, c) in f["clouds"]:
        clouds.append(p); ints.append(intensity_from_colors(c))
    embs.append(f["det_emb"]); qs.append(len(f["clouds"]))
kw = dict(det_emb=np.concatenate(embs), fpfh_voxel_size=0.05, fpfh_global_dist_factor=1.5, fpfh_local_dist_factor=1.5, seed=5,
          job_id_base=100 * rank)
ctx3 = RegContext(3 << 30)
ref_mem = MemoryShard(ctx3, list(w2.embeddings), w2.points, colors=w2.colors, device=str(dev))
ref = LocaliseEngine(ref_mem).localise_batch(CloudBatch.from_numpy(clouds, ints, device=str(dev)), qs, **kw)
ref_mem.close()
ctx4 = RegContext(3 << 30)
mem = MemoryShard(ctx4, list(w2.embeddings), w2.points, colors=w2.colors, device=str(dev), shard=(rank, world), shard_clouds=True)
eng2 = LocaliseEngine(mem)
res = eng2.localise_batch(CloudBatch.from_numpy(clouds, ints, device=str(dev)), qs, **kw)
n_jobs = 0
for a, b in zip(res, ref):
    assert a.assignments == b.assignments and a.best == b.best and a.n_clean == b.n_clean
    assert np.array_equal(a.pose, b.pose) and np.array_equal(a.pose_corrected, b.pose_corrected)
    for ra, rb in zip(a.records, b.records):
        n_jobs += 1
        assert np.array_equal(ra["T"], rb["T"]) and ra["fitness"] == rb["fitness"] and ra["rmse"] == rb["rmse"]
        assert np.array_equal(ra["ransac_stats"], rb["ransac_stats"]) and ra["full_fitness"] == rb["full_fitness"]
st = eng2.route_stats
tot = [None] * world
dist.all_gather_object(tot, (st["jobs_shipped"], st["instances_fetched"], n_jobs))
torch.cuda.synchronize()
comm.close()
dist.barrier()
if rank == 0:
    assert sum(t[0] for t in tot) > 0 and sum(t[2] for t in tot) > 0, tot
    print("GPU_MULTIRANK_OK", tot, flush=True)
dist.destroy_process_group()
'''


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device (the one-GPU rehearsals are "
                    "tests/test_gpu_routing.py and the gloo world-2 / world-3 tests of the CPU suite)")
def test_sharded_paths_over_rccl_with_two_ranks(tmp_path):
    script = tmp_path / "gpu_multirank_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, IBL_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29547", str(script)], env=env, capture_output=True, text=True, timeout=900)
    print(out.stdout[-3000:])
    assert out.returncode == 0 and "GPU_MULTIRANK_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-4000:]


def test_worker_rehearsal_on_one_gpu(tmp_path):
    """the same worker script with both ranks on device 0 over gloo (no RCCL communicator): parts 2 and 3 -- agreement flags, sharded stage A
    over several steps and through the pipelined scheduler -- so that a multi-GPU lease adds RCCL itself and nothing else"""
    script = tmp_path / "gpu_multirank_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, IBL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", IBL_MULTIRANK_REHEARSE="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29548", str(script)], env=env, capture_output=True, text=True, timeout=600)
    print(out.stdout[-3000:])
    assert out.returncode == 0 and "GPU_MULTIRANK_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-4000:]

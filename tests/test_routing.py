"""Sharded-cloud registration routing (routing.py, SURVEY §8e / VERDICT item 8) on the CPU: the plan, the exchange and the way home.

The registration itself needs the GPU (tests/test_gpu_routing.py runs it); here the executor's `compute` is a digest of exactly what a
registration would read -- the rows of its detected segments, the rows of every array of its target instances, its RANSAC id -- so a job
that reaches the wrong rank, sees another instance's rows, or loses its id changes the digest.  world-size 2 and 3 over gloo."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_plan_routes_rules():
    from ibloc_amd.routing import owners_of, plan_routes
    M, W = 10, 2                                           # rank 0 owns 0-4, rank 1 owns 5-9
    assert owners_of([0, 4, 5, 9, -1], M, W).tolist() == [0, 0, 1, 1, -1]
    src0 = [[0, 1, -1], [2, -1, -1], [0, 2, 3]]
    tgt0 = [[1, 2, -1], [7, -1, -1], [3, 8, 9]]            # local / all on rank 1 / mixed
    src1 = [[0, -1, -1], [1, 2, -1]]
    tgt1 = [[2, -1, -1], [6, 1, -1]]                       # all on rank 0 / mixed
    plan = plan_routes([src0, src1], [tgt0, tgt1], M, W)
    assert plan.executor[0].tolist() == [0, 1, 0] and plan.executor[1].tolist() == [0, 1]
    assert plan.det_send[(0, 1)].tolist() == [2] and plan.det_send[(1, 0)].tolist() == [0]
    assert plan.inst_send[(1, 0)].tolist() == [8, 9] and plan.inst_send[(0, 1)].tolist() == [1]
    assert set(plan.det_send) == {(0, 1), (1, 0)} and set(plan.inst_send) == {(1, 0), (0, 1)}


def test_plan_routes_edge_cases():
    from ibloc_amd.routing import plan_routes
    M, W = 9, 3                                            # ranks own 0-2, 3-5, 6-8
    none = np.zeros((0, 3), np.int64)
    # rank 1 has no jobs at all; rank 0 has a job without targets (stays home, moves nothing) and two jobs sharing a remote segment
    src0 = [[0, -1, -1], [1, 2, -1], [2, 1, -1]]
    tgt0 = [[-1, -1, -1], [7, 8, -1], [6, -1, -1]]
    src2 = [[0, 1, 2]]
    tgt2 = [[0, 4, 8]]                                     # three owners: runs at home (rank 2), fetches 0 from rank 0 and 4 from rank 1
    plan = plan_routes([src0, none, src2], [tgt0, none, tgt2], M, W)
    assert plan.executor[0].tolist() == [0, 2, 2] and len(plan.executor[1]) == 0 and plan.executor[2].tolist() == [2]
    assert plan.det_send == {} or set(plan.det_send) == {(0, 2)}
    assert plan.det_send[(0, 2)].tolist() == [1, 2]        # segments 1 and 2 once, although two jobs use them
    assert {k: v.tolist() for k, v in plan.inst_send.items()} == {(0, 2): [0], (1, 2): [4]}


WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["IBL_ROOT"])
import numpy as np, torch, torch.distributed as dist
from ibloc_amd.parallel import shard_range
from ibloc_amd.routing import InstanceStore, Transport, routed_evaluate, routed_register
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
M = 23
rng = np.random.default_rng(5)                                   # the same memory on every rank
sizes = rng.integers(3, 40, size=M)
off_all = np.concatenate([[0], np.cumsum(sizes)])
full = {"pts": rng.standard_normal((off_all[-1], 4)).astype(np.float32),
        "fpfh": rng.standard_normal((off_all[-1], 33)).astype(np.float32),
        "split": rng.standard_normal((off_all[-1], 48)).astype(np.float16),
        "norm": rng.standard_normal(off_all[-1]).astype(np.float32)}
bbox_all = rng.standard_normal((M, 6)).astype(np.float32)

def store_for(lo, hi):
    return InstanceStore(lo, off_all[lo:hi + 1] - off_all[lo], {k: torch.from_numpy(v[off_all[lo]:off_all[hi]]) for k, v in full.items()},
                         {"bbox": bbox_all[lo:hi]})

def batch_of(r):                                                  # rank r's detections and jobs (any rank can rebuild them)
    g = np.random.default_rng(100 + r)
    S = 9 + r
    seg = g.integers(2, 30, size=S)
    det_off = np.concatenate([[0], np.cumsum(seg)])
    det = g.standard_normal((det_off[-1], 4)).astype(np.float32)
    J = 14
    src = np.full((J, 3), -1, np.int64); tgt = np.full((J, 3), -1, np.int64)
    for j in range(J):
        L = int(g.integers(1, 4))
        src[j, :L] = g.choice(S, size=L, replace=False)
        if j % 3 == 0:                                            # every third job: all targets inside one shard
            lo, hi = shard_range(M, int(g.integers(0, world)), world)
            tgt[j, :L] = g.choice(np.arange(lo, hi), size=L, replace=False)
        else:
            tgt[j, :L] = g.choice(M, size=L, replace=False)
    ids = (1000 * (r + 1) + np.arange(J)).astype(np.uint32)
    return det, det_off, src, tgt, ids

def compute(det_pts, det_off, n_home, arrays, per_inst, mem_off, js, jt, ids):
    d = np.zeros((len(ids), 4), dtype=np.float64)
    for k in range(len(ids)):
        for c in range(3):
            if js[k, c] >= 0:
                d[k, 0] += det_pts[det_off[js[k, c]]:det_off[js[k, c] + 1]].double().sum().item() * (c + 1)
            if jt[k, c] >= 0:
                a, b = int(mem_off[jt[k, c]]), int(mem_off[jt[k, c] + 1])
                for name in sorted(arrays):
                    d[k, 1] += arrays[name][a:b].double().sum().item() * (c + 1)
                d[k, 2] += float(per_inst["bbox"][jt[k, c]].astype(np.float64).sum()) * (c + 1)
        d[k, 3] = float(ids[k])
    return {"digest": d, "echo": np.asarray(ids, dtype=np.int64)}

tr = Transport()
lo, hi = shard_range(M, rank, world)
det, det_off, src, tgt, ids = batch_of(rank)
stats = {}
got, plan = routed_register(tr, M, torch.from_numpy(det), det_off, src, tgt, ids, store_for(lo, hi), sizes, compute, stats)
# the unsharded run of the same batch: everything local, identity mapping
ref = compute(torch.from_numpy(det), det_off, len(det_off) - 1, store_for(0, M).arrays, {"bbox": bbox_all}, off_all, src.astype(np.int32),
              tgt.astype(np.int32), ids)
assert np.array_equal(got["digest"], ref["digest"]), (rank, got["digest"] - ref["digest"])
assert np.array_equal(got["echo"], ids.astype(np.int64))
assert stats["jobs_shipped"] > 0 and stats["instances_fetched"] > 0, stats       # both routes were exercised

# whole-memory evaluation: nearest own point by brute force, MIN over the ranks == brute force over all points
thr2 = 0.8 ** 2
def eval_local_on(points):
    def f(pts, jb, je, G):
        out = []
        for b, e, T in zip(jb, je, G):
            p = pts[b:e, :3].double().numpy() @ T[:3, :3].T + T[:3, 3]
            d = ((p[:, None, :] - points[None, :, :3].astype(np.float64)) ** 2).sum(-1).min(1) if len(points) else np.full(len(p), np.inf)
            out.append(np.where(d < thr2, d, np.inf).astype(np.float32))
        return torch.from_numpy(np.concatenate(out))
    return f
jb = det_off[src[:, 0]]; je = det_off[src[:, 0] + 1]
G = np.tile(np.eye(4), (len(jb), 1, 1)); G[:, :3, 3] = 0.05 * (rank + 1)
fit, rmse = routed_evaluate(tr, torch.from_numpy(det), jb, je, G, eval_local_on(full["pts"][off_all[lo]:off_all[hi]]))
from ibloc_amd.parallel import fitness_rmse_from_d2
fit0, rmse0 = fitness_rmse_from_d2(eval_local_on(full["pts"])(torch.from_numpy(det), jb, je, G), (je - jb).tolist())
assert np.array_equal(fit, fit0) and np.allclose(rmse, rmse0, rtol=1e-12), (fit, fit0)
# parameters that every job of an executor shares travel with the job tables: ranks that disagree raise, all of them
try:
    routed_register(tr, M, torch.from_numpy(det), det_off, src, tgt, ids, store_for(lo, hi), sizes, compute, {}, params=(7 + rank, 0.05))
    raise SystemExit("ranks with different parameters did not raise")
except ValueError as e:
    assert "different registration parameters" in str(e)
got2, _ = routed_register(tr, M, torch.from_numpy(det), det_off, src, tgt, ids, store_for(lo, hi), sizes, compute, {}, params=(7, 0.05))
assert np.array_equal(got2["digest"], ref["digest"])
dist.barrier()
if rank == 0:
    print("ROUTING_OK", stats)
'''


@pytest.mark.parametrize("world,port", [(2, 29531), (3, 29532)])
def test_routed_registration_matches_unsharded(tmp_path, world, port):
    script = tmp_path / "routing_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, IBL_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ROUTING_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]

"""Registration with the memory CLOUDS sharded by instance range (BASELINE configs[4]: 100 k-point objects do not fit
replicated).  The loop being distributed is object_memory/object_memory.py:1020-1106: per assignment, concatenate its <= 3 detected
clouds and its <= 3 memory clouds, register them (:1023-1034), then score the transform against the WHOLE memory (:1087-1106).

Every rank localises its own frames ("home" of their jobs) and owns the clouds + cached registration features of the instances
[lo, hi) = parallel.shard_range(M, rank, world).  Per batch, all ranks together:

  1. all-gather the job tables (<= 3 detected segment ids, <= 3 global instance ids, RANSAC id per job; a few KB);
  2. every rank derives the same plan (`plan_routes`): a job whose targets all live on ONE rank runs there -- its detected segments
     (<= 3 x ~5 000 points x 16 B) travel to the owner, 50x less than the instance arrays (280 B per point with the cached features);
     a job whose targets span ranks runs at home, which fetches the missing instances' clouds + cached features from their owners;
  3. one point-to-point exchange moves both kinds of payload (each ordered pair of ranks: one message);
  4. each executor registers its job list (`ibl_register_batch_ids`: a job keeps its RANSAC id, and per-instance features do not depend on
     the batch they were computed in, so the result is bit-identical to the unsharded run);
  5. results (a few hundred bytes per job) return to the home ranks;
  6. whole-memory evaluation: the cleaned detected points and the global transforms are all-gathered, every rank measures every job
     against the points it owns, reduce-scatter(MIN) of the per-point squared distances (a rank only needs its own jobs' minima).

The transport is torch.distributed ("nccl" = RCCL over xGMI; "gloo" with host staging in the tests).  Everything here that is not a
collective is a pure function of the gathered tables, so the ranks cannot disagree on message sizes or orders.
"""
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.distributed as dist

from .parallel import shard_range


def owners_of(inst, M: int, world: int) -> np.ndarray:
    """Rank owning each global instance id (-1 stays -1)."""
    inst = np.asarray(inst, dtype=np.int64)
    bounds = np.array([shard_range(M, r, world)[0] for r in range(world)] + [M], dtype=np.int64)
    own = np.searchsorted(bounds, inst, side="right") - 1
    return np.where(inst < 0, -1, own).astype(np.int64)


@dataclass
class RoutePlan:
    executor: list                                   # per home rank: (J_h,) rank that runs each job
    det_send: dict = field(default_factory=dict)     # (home, executor) -> sorted unique segment ids of home's batch
    inst_send: dict = field(default_factory=dict)    # (owner, home) -> sorted unique global instance ids


def plan_routes(job_src, job_tgt, M: int, world: int) -> RoutePlan:
    """job_src[h] (J_h, 3) detected segment ids of rank h's batch, job_tgt[h] (J_h, 3) global instance ids, -1 padded."""
    plan = RoutePlan(executor=[])
    det_need, inst_need = {}, {}
    for h in range(world):
        src = np.asarray(job_src[h], dtype=np.int64).reshape(-1, 3)
        tgt = np.asarray(job_tgt[h], dtype=np.int64).reshape(-1, 3)
        own = owners_of(tgt, M, world)
        ex = np.full(len(tgt), h, dtype=np.int64)
        for j in range(len(tgt)):
            o = np.unique(own[j][own[j] >= 0])
            if len(o) == 1:
                ex[j] = o[0]                                           # all targets on one rank: the job goes there
            if ex[j] != h:
                det_need.setdefault((h, int(ex[j])), set()).update(int(s) for s in src[j] if s >= 0)
            else:
                for t, ow in zip(tgt[j], own[j]):
                    if t >= 0 and ow != h:
                        inst_need.setdefault((int(ow), h), set()).add(int(t))
        plan.executor.append(ex)
    plan.det_send = {k: np.array(sorted(v), dtype=np.int64) for k, v in det_need.items()}
    plan.inst_send = {k: np.array(sorted(v), dtype=np.int64) for k, v in inst_need.items()}
    return plan


class Transport:
    """torch.distributed group + where its tensors live ("cuda" for nccl, "cpu" for gloo)."""

    def __init__(self, group=None, comm_device=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if comm_device is None:
            comm_device = "cuda" if dist.is_initialized() and dist.get_backend(group) == "nccl" else "cpu"
        self.comm_device = torch.device(comm_device)

    def all_gather_object(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def exchange(self, send: dict, recv_bytes: dict) -> dict:
        """send[dst] = uint8 tensor; recv_bytes[src] = size of the message from src (0: none).  Returns {src: uint8 tensor}."""
        ops, bufs, keep = [], {}, []
        for src in sorted(recv_bytes):
            if recv_bytes[src] > 0 and src != self.rank:
                bufs[src] = torch.empty(int(recv_bytes[src]), dtype=torch.uint8, device=self.comm_device)
                ops.append(dist.P2POp(dist.irecv, bufs[src], src, group=self.group))
        for dst in sorted(send):
            if send[dst].numel() > 0 and dst != self.rank:
                t = send[dst].to(self.comm_device).contiguous()
                keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, dst, group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return bufs

    def all_gather_rows(self, rows: torch.Tensor, counts) -> list:
        """rows (n_me, C) of every rank -> list of (n_r, C) tensors on rows.device (padded all-gather)."""
        if self.world == 1:
            return [rows]
        nmax = max(int(c) for c in counts)
        pad = torch.zeros((nmax,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=self.comm_device)
        pad[:rows.shape[0]] = rows.to(self.comm_device)
        out = torch.empty((self.world * nmax,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=self.comm_device)
        dist.all_gather_into_tensor(out, pad, group=self.group)
        out = out.view((self.world, nmax) + tuple(rows.shape[1:]))
        return [out[r, :int(counts[r])].to(rows.device) for r in range(self.world)]

    def all_reduce_min(self, t: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return t
        c = t.to(self.comm_device)
        dist.all_reduce(c, op=dist.ReduceOp.MIN, group=self.group)
        return c.to(t.device)

    def reduce_scatter_min(self, t: torch.Tensor, sizes) -> torch.Tensor:
        """t = the sections of all ranks back to back (sizes[r] elements for rank r), the same layout on every rank -> the element-wise
        minimum over the ranks of THIS rank's section.  RCCL: one reduce-scatter on sections padded to the longest (each rank receives
        1 / W of what the all-reduce moved to everyone); gloo has no reduce-scatter: all-reduce, then slice."""
        start = int(sum(sizes[:self.rank]))
        n = int(sizes[self.rank])
        if self.world == 1:
            return t[start:start + n]
        if dist.get_backend(self.group) != "nccl":
            return self.all_reduce_min(t)[start:start + n]
        nmax = max(int(x) for x in sizes)
        pad = torch.full((self.world, max(nmax, 1)), float("inf"), dtype=t.dtype, device=self.comm_device)
        o = 0
        for r, k in enumerate(sizes):
            pad[r, :int(k)] = t[o:o + int(k)]
            o += int(k)
        out = torch.empty(max(nmax, 1), dtype=t.dtype, device=self.comm_device)
        dist.reduce_scatter_tensor(out, pad.view(-1), op=dist.ReduceOp.MIN, group=self.group)
        return out[:n].to(t.device)


def _bytes(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous().view(torch.uint8).reshape(-1)


def _row_index(off: np.ndarray, ids, device) -> torch.Tensor:
    """Row numbers of the segments `ids` of a packed array with offsets `off`, in the order given."""
    if len(ids) == 0:
        return torch.zeros(0, dtype=torch.int64, device=device)
    return torch.from_numpy(np.concatenate([np.arange(off[i], off[i + 1], dtype=np.int64) for i in ids])).to(device)


class InstanceStore:
    """What an owner keeps per instance: packed per-point arrays (all with the same row offsets) + per-instance rows (bbox)."""

    def __init__(self, lo: int, off: np.ndarray, arrays: dict, per_instance: dict = None):
        self.lo = int(lo)
        self.off = np.asarray(off, dtype=np.int64)                 # (n_local + 1,)
        self.arrays = arrays                                        # name -> (rows, ...) tensor
        self.per_instance = per_instance or {}                      # name -> (n_local, ...) numpy array
        self.names = sorted(arrays)
        self.pi_names = sorted(self.per_instance)

    @property
    def n_local(self):
        return len(self.off) - 1

    def row_bytes(self) -> int:
        return sum(int(np.prod(self.arrays[k].shape[1:], dtype=np.int64)) * self.arrays[k].element_size() for k in self.names)

    def inst_bytes(self) -> int:
        return sum(int(np.prod(self.per_instance[k].shape[1:], dtype=np.int64)) * self.per_instance[k].itemsize for k in self.pi_names)

    def layout(self):
        """What a peer needs to unpack a message of this store's instances: (name, row shape, dtype) lists."""
        return ([(k, tuple(self.arrays[k].shape[1:]), self.arrays[k].dtype) for k in self.names],
                [(k, tuple(self.per_instance[k].shape[1:]), self.per_instance[k].dtype) for k in self.pi_names])


def routed_register(tr: Transport, M: int, det_pts: torch.Tensor, det_off: np.ndarray, job_src, job_tgt, job_ids, store: InstanceStore,
                    inst_sizes_all: np.ndarray, compute, stats: dict = None, params=None):
    """Collective.  det_pts (N, 4) float32 + det_off (S + 1): this rank's cleaned detected segments; job_src (J, 3) segment ids, job_tgt
    (J, 3) global instance ids, job_ids (J,) uint32; store: the instances this rank owns; inst_sizes_all (M,): points of every instance.
    compute(det_pts, det_off, n_home_segs, arrays, per_instance, mem_off, js, jt, ids) -> dict of (J_exec, ...) numpy arrays runs the job
    list of this rank (pool-relative indices).  Returns the dict for THIS rank's J jobs, in their order.
    params: whatever `compute` applies to EVERY job it runs (seed, radii, iteration caps): a shipped job runs with the executor's
    values, so they travel with the job tables and must be equal on all ranks -- otherwise every rank raises (a job's result must not
    depend on where it ran)."""
    me, W = tr.rank, tr.world
    det_off = np.asarray(det_off, dtype=np.int64)
    job_src = np.asarray(job_src, dtype=np.int64).reshape(-1, 3)
    job_tgt = np.asarray(job_tgt, dtype=np.int64).reshape(-1, 3)
    job_ids = np.asarray(job_ids, dtype=np.uint32).reshape(-1)
    metas = tr.all_gather_object({"src": job_src, "tgt": job_tgt, "ids": job_ids, "seg": np.diff(det_off), "params": params})
    if any(m["params"] != metas[0]["params"] for m in metas):
        raise ValueError("routed registration: the ranks passed different registration parameters "
                         f"({[m['params'] for m in metas]}); a shipped job would run with its executor's values")
    plan = plan_routes([m["src"] for m in metas], [m["tgt"] for m in metas], M, W)
    dev = det_pts.device
    row_b, inst_b = store.row_bytes(), store.inst_bytes()
    lo_of = [shard_range(M, r, W)[0] for r in range(W)]

    # ---- one message per ordered pair: [detected segments for jobs that run there | instances fetched by jobs that run at their home]
    send, recv_bytes = {}, {}
    for x in range(W):
        if x == me:
            continue
        parts = []
        segs = plan.det_send.get((me, x), ())
        if len(segs):
            parts.append(_bytes(det_pts[_row_index(det_off, segs, dev)]))
        insts = plan.inst_send.get((me, x), ())
        if len(insts):
            rows = _row_index(store.off, [int(t) - store.lo for t in insts], dev)
            for k in store.names:
                parts.append(_bytes(store.arrays[k][rows]))
            for k in store.pi_names:
                parts.append(torch.from_numpy(np.ascontiguousarray(store.per_instance[k][[int(t) - store.lo for t in insts]]).view(np.uint8).reshape(-1)).to(dev))
        send[x] = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.uint8, device=dev)
        n_det = int(sum(metas[x]["seg"][s] for s in plan.det_send.get((x, me), ())))
        n_rows = int(sum(inst_sizes_all[t] for t in plan.inst_send.get((x, me), ())))
        recv_bytes[x] = 16 * n_det + row_b * n_rows + inst_b * len(plan.inst_send.get((x, me), ()))
    got = tr.exchange(send, recv_bytes)
    if stats is not None:
        stats["bytes_sent"] = stats.get("bytes_sent", 0) + int(sum(t.numel() for t in send.values()))
        stats["jobs_shipped"] = stats.get("jobs_shipped", 0) + int((plan.executor[me] != me).sum())
        stats["instances_fetched"] = stats.get("instances_fetched", 0) + int(sum(len(v) for (o, h), v in plan.inst_send.items() if h == me))

    # ---- pools of this executor
    det_parts, det_sizes = [det_pts], [np.diff(det_off)]
    det_index = {(me, s): s for s in range(len(det_off) - 1)}
    fetched = {k: [] for k in store.names}
    fetched_pi = {k: [] for k in store.pi_names}
    fetched_ids, fetched_sizes = [], []
    arr_layout, pi_layout = store.layout()
    for x in sorted(got):
        buf = got[x].to(dev)
        pos = 0
        segs = plan.det_send.get((x, me), ())
        if len(segs):
            sizes = np.array([metas[x]["seg"][s] for s in segs], dtype=np.int64)
            n = int(sizes.sum())
            base = sum(len(z) for z in det_sizes)
            det_parts.append(buf[pos:pos + 16 * n].view(torch.float32).reshape(n, 4))
            pos += 16 * n
            for k, s in enumerate(segs):
                det_index[(x, int(s))] = base + k
            det_sizes.append(sizes)
        insts = plan.inst_send.get((x, me), ())
        if len(insts):
            n = int(sum(inst_sizes_all[t] for t in insts))
            for name, shp, dt in arr_layout:
                nb = n * int(np.prod(shp, dtype=np.int64)) * torch.empty(0, dtype=dt).element_size()
                fetched[name].append(buf[pos:pos + nb].view(dt).reshape((n,) + shp))
                pos += nb
            for name, shp, dt in pi_layout:
                nb = len(insts) * int(np.prod(shp, dtype=np.int64)) * np.dtype(dt).itemsize
                fetched_pi[name].append(buf[pos:pos + nb].cpu().numpy().view(dt).reshape((len(insts),) + shp))
                pos += nb
            fetched_ids += [int(t) for t in insts]
            fetched_sizes += [int(inst_sizes_all[t]) for t in insts]
        assert pos == buf.numel(), "routing: message layout mismatch"
    pool_pts = torch.cat(det_parts) if len(det_parts) > 1 else det_pts
    pool_off = np.concatenate([[0], np.cumsum(np.concatenate(det_sizes))]).astype(np.int64)

    # jobs this rank runs, in (home, job) order
    ex_home, ex_job = [], []
    for h in range(W):
        for j in np.nonzero(plan.executor[h] == me)[0]:
            ex_home.append(h)
            ex_job.append(int(j))
    js = np.full((len(ex_job), 3), -1, dtype=np.int32)
    tg = np.full((len(ex_job), 3), -1, dtype=np.int64)
    ids = np.zeros(len(ex_job), dtype=np.uint32)
    for k, (h, j) in enumerate(zip(ex_home, ex_job)):
        for c in range(3):
            s = int(metas[h]["src"][j, c])
            if s >= 0:
                js[k, c] = det_index[(h, s)]
        tg[k] = metas[h]["tgt"][j]
        ids[k] = metas[h]["ids"][j]
    # memory pool: the store itself when nothing was fetched (instance t -> t - lo); otherwise the referenced own instances followed by
    # the fetched ones (either way an instance's rows are the owner's bits)
    if not fetched_ids:
        arrays, per_inst, mem_off = store.arrays, store.per_instance, store.off
        jt = np.where(tg >= 0, tg - store.lo, -1).astype(np.int32)
    else:
        own_ref = sorted({int(t) for t in tg.reshape(-1) if t >= 0 and lo_of[me] <= t < lo_of[me] + store.n_local})
        rows = _row_index(store.off, [t - store.lo for t in own_ref], dev)
        arrays = {k: torch.cat([store.arrays[k][rows]] + fetched[k]) for k in store.names}
        per_inst = {k: np.concatenate([store.per_instance[k][[t - store.lo for t in own_ref]]] + fetched_pi[k]) for k in store.pi_names}
        sizes = [int(store.off[t - store.lo + 1] - store.off[t - store.lo]) for t in own_ref] + fetched_sizes
        mem_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        where = {t: i for i, t in enumerate(own_ref + fetched_ids)}
        jt = np.array([[where[int(t)] if t >= 0 else -1 for t in row] for row in tg], dtype=np.int32).reshape(-1, 3)
    res = compute(pool_pts, pool_off, len(det_off) - 1, arrays, per_inst, mem_off, js, jt, ids) if len(ex_job) else {}

    # ---- results go home (small: all-gather of {home: (job numbers, rows)})
    out = {}
    for h in range(W):
        sel = [k for k, hh in enumerate(ex_home) if hh == h]
        if sel:
            out[h] = (np.array([ex_job[k] for k in sel], dtype=np.int64), {name: np.asarray(v)[sel] for name, v in res.items()})
    mine = {}
    J = len(job_src)
    for part in tr.all_gather_object(out):
        if me in part:
            jobs, rows = part[me]
            for name, v in rows.items():
                if name not in mine:
                    mine[name] = np.zeros((J,) + v.shape[1:], dtype=v.dtype)
                mine[name][jobs] = v
    return mine, plan


def routed_evaluate(tr: Transport, clean_pts: torch.Tensor, job_begin, job_end, G, evaluate_points_local):
    """Collective whole-memory evaluation with sharded clouds: every rank scores every rank's jobs against the points it owns.
    evaluate_points_local(pts, jb, je, G) -> per-point squared distances (+inf: none within the threshold) of the jobs back to back.
    Returns (fitness (J,), inlier rmse (J,)) of this rank's jobs."""
    from .parallel import fitness_rmse_from_d2
    jb = np.asarray(job_begin, dtype=np.int64)
    je = np.asarray(job_end, dtype=np.int64)
    metas = tr.all_gather_object({"jb": jb, "je": je, "G": np.asarray(G, dtype=np.float64), "n": int(clean_pts.shape[0])})
    pts_all = tr.all_gather_rows(clean_pts, [m["n"] for m in metas])
    d2_parts, sizes = [], []
    for r, m in enumerate(metas):
        n = int((m["je"] - m["jb"]).sum())
        sizes.append(n)
        if n:
            d2_parts.append(evaluate_points_local(pts_all[r], m["jb"], m["je"], m["G"]))
    if not d2_parts:
        return np.zeros(0), np.zeros(0)
    d2 = tr.reduce_scatter_min(torch.cat(d2_parts), sizes)          # each rank only needs the minima of its own jobs
    return fitness_rmse_from_d2(d2, (je - jb).tolist())

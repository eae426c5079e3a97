"""Multi-GPU layout of the path.  One process per GPU; `torch.distributed` is the transport (backend "nccl" IS RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

* Query frames are data-parallel: every rank localises its own frames -- no communication.
* The embedding memory can be sharded by contiguous instance range (`shard_range`).  Per step every rank then
    1. all-gathers the L2-normalised query embeddings of all ranks (R x D fp32 per rank; 688 KB for 224 rows of 768),
    2. matches ALL ranks' rows against ITS instance range and keeps, per row, the k_hi largest / k_lo smallest fp16 similarities
       with their GLOBAL instance indices (`ibl_match_topk`, csrc/topk.hip),
    3. exchanges those candidate lists ((6 S + 8) bytes per row and shard, S = k_hi + k_lo: 1.3 KB -- the north star's "RCCL
       all-gather of per-shard top-k matches over xGMI before registration"; since round 3 as an all-to-all: the owner of a row
       receives the W per-shard lists of that row, W * cap * 1.3 KB = 2.4 MB per rank and step at W = 8, cap = 224, where the
       all-gather delivered every rank's lists to everyone, W times as much),
   and the owner of a frame merges the per-shard lists and runs the assignment search on them (`ibl_assign_candidates`), which proves
   per frame that the result equals the search on the full rows.  A frame it cannot prove (ties at the candidate threshold) is redone
   on full rows: the ranks agree with one all-reduce(MAX) of a flag and all-gather their similarity blocks for that step only.
* Whole-memory evaluation with the clouds sharded: all-reduce(MIN) of per-point nearest distances (`evaluate_sharded`).
All collectives of a step are issued by ONE thread in a fixed order (LocaliseEngine's stage A), so they pair up across ranks.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, balanced range [lo, hi) of rank `rank`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def frames_for_rank(n_frames: int, rank: int, world: int):
    return range(*shard_range(n_frames, rank, world))


def pack_candidates(val: torch.Tensor, idx: torch.Tensor, cnt: torch.Tensor) -> torch.Tensor:
    """(R, S) fp16 values, (R, S) int32 global indices, (R, 2) int32 counts -> one (R, S + S/2 + 2) int32 payload."""
    assert val.dtype == torch.float16 and idx.dtype == torch.int32 and cnt.dtype == torch.int32 and val.shape[1] % 2 == 0
    return torch.cat([idx, val.contiguous().view(torch.int32), cnt], dim=1).contiguous()


def unpack_candidates(buf: torch.Tensor, S: int):
    idx = buf[..., :S]
    val = buf[..., S:S + S // 2].contiguous().view(torch.float16)
    cnt = buf[..., S + S // 2:]
    return val, idx, cnt


def merge_lists_host(val: np.ndarray, idx: np.ndarray, cnt: np.ndarray):
    """Per-shard candidate lists of the same rows, (W, R, S) / (W, R, S) / (W, R, 2) host arrays -> (R, W * S) lists with the valid
    entries first and (R,) counts: the layout `ibl_assign_candidates` takes (it selects the overall k_hi / k_lo itself)."""
    W, R, S = val.shape
    n = cnt.sum(axis=2)                                                  # (W, R) valid entries per shard
    valid = np.arange(S)[None, None, :] < n[:, :, None]                  # (W, R, S)
    v = np.transpose(val, (1, 0, 2)).reshape(R, W * S)
    j = np.transpose(idx, (1, 0, 2)).reshape(R, W * S)
    m = np.transpose(valid, (1, 0, 2)).reshape(R, W * S)
    order = np.argsort(~m, axis=1, kind="stable")                        # valid entries first, source order kept
    return np.take_along_axis(v, order, 1), np.take_along_axis(j, order, 1), n.sum(axis=0).astype(np.int32)


class RcclComm:
    """The library's own RCCL communicator (include/ibloc.h `ibl_comm_*`): the collectives of the sharded path without going through
    torch.distributed's process group.  The unique id is created on rank 0 and broadcast with torch.distributed (any host channel
    would do); every rank then calls ncclCommInitRank on its current device.  `ShardExchange(comm=RcclComm(...))` uses it."""

    def __init__(self, group=None):
        import ctypes as C
        from . import _lib
        self._lib, self._C = _lib, C
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        buf = (C.c_ubyte * 256)()
        n = 128
        if self.rank == 0:
            n = _lib.lib.ibl_comm_unique_id(buf, 256)
            if n < 0:
                _lib.check(n, "ibl_comm_unique_id")
        ids = [bytes(buf[:n])] if self.rank == 0 else [None]
        dist.broadcast_object_list(ids, src=0, group=group)
        raw = (C.c_ubyte * len(ids[0])).from_buffer_copy(ids[0])
        self._h = C.c_void_p()
        _lib.check(_lib.lib.ibl_comm_init(C.byref(self._h), self.rank, self.world, raw, len(ids[0])), "ibl_comm_init")

    @staticmethod
    def single():
        """a one-rank communicator (no process group needed): what a single-GPU process gets"""
        self = RcclComm.__new__(RcclComm)
        import ctypes as C
        from . import _lib
        self._lib, self._C, self.rank, self.world = _lib, C, 0, 1
        buf = (C.c_ubyte * 256)()
        n = _lib.lib.ibl_comm_unique_id(buf, 256)
        if n < 0:
            _lib.check(n, "ibl_comm_unique_id")
        self._h = C.c_void_p()
        _lib.check(_lib.lib.ibl_comm_init(C.byref(self._h), 0, 1, buf, n), "ibl_comm_init")
        return self

    def close(self):
        if self._h:
            self._lib.lib.ibl_comm_destroy(self._h)
            self._h = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def all_gather(self, out: torch.Tensor, mine: torch.Tensor):
        assert out.is_contiguous() and mine.is_contiguous() and out.numel() * out.element_size() == self.world * mine.numel() * mine.element_size()
        self._lib.check(self._lib.lib.ibl_allgather_topk(self._h, mine.data_ptr(), out.data_ptr(), mine.numel() * mine.element_size(),
                                                         torch.cuda.current_stream().cuda_stream), "ibl_allgather_topk")

    def all_to_all(self, out: torch.Tensor, mine: torch.Tensor):
        """equal blocks: block r of `mine` goes to rank r, block r of `out` comes from rank r"""
        assert out.is_contiguous() and mine.is_contiguous() and out.numel() == mine.numel() and out.dtype == mine.dtype
        nbytes = mine.numel() * mine.element_size()
        assert nbytes % self.world == 0
        self._lib.check(self._lib.lib.ibl_alltoall(self._h, mine.data_ptr(), out.data_ptr(), nbytes // self.world,
                                                   torch.cuda.current_stream().cuda_stream), "ibl_alltoall")

    def all_reduce_min(self, buf: torch.Tensor):
        assert buf.dtype == torch.float32 and buf.is_contiguous()
        self._lib.check(self._lib.lib.ibl_allreduce_min(self._h, buf.data_ptr(), buf.numel(), torch.cuda.current_stream().cuda_stream),
                        "ibl_allreduce_min")

    def all_reduce_max_i32(self, buf: torch.Tensor):
        assert buf.dtype == torch.int32 and buf.is_contiguous()
        self._lib.check(self._lib.lib.ibl_allreduce_max_i32(self._h, buf.data_ptr(), buf.numel(), torch.cuda.current_stream().cuda_stream),
                        "ibl_allreduce_max_i32")


class ShardExchange:
    """The collectives of one rank's stage A when the embedding memory is sharded over `group`.  rows_cap: fixed number of query
    rows every rank contributes per step (shorter batches are zero-padded), so that no size has to be negotiated.  comm: an
    `RcclComm` (the library's communicator) instead of torch.distributed's process group -- the same collectives either way."""

    def __init__(self, group=None, rows_cap: int = 224, comm: "RcclComm" = None):
        self.group = group
        self.comm = comm
        self.world = comm.world if comm is not None else dist.get_world_size(group)
        self.rank = comm.rank if comm is not None else dist.get_rank(group)
        self.cap = int(rows_cap)

    def _all_gather(self, out, mine):
        if self.comm is not None:
            self.comm.all_gather(out, mine)
        else:
            dist.all_gather_into_tensor(out, mine, group=self.group)

    def gather_queries(self, detn: torch.Tensor) -> torch.Tensor:
        """(R, D) normalised query rows of this rank -> (W * cap, D): rank r's rows at [r * cap, r * cap + R_r), zeros after."""
        R, D = detn.shape
        if R > self.cap:
            raise ValueError(f"{R} query rows in a step, rows_cap is {self.cap}")
        mine = torch.zeros((self.cap, D), dtype=detn.dtype, device=detn.device)
        mine[:R] = detn
        out = torch.empty((self.world * self.cap, D), dtype=detn.dtype, device=detn.device)
        self._all_gather(out, mine)
        return out

    def gather_candidates(self, val: torch.Tensor, idx: torch.Tensor, cnt: torch.Tensor):
        """This rank's lists for all W * cap rows -> the lists of every shard for THIS rank's rows: (W, cap, S) / (W, cap, S) / (W, cap, 2)."""
        S = val.shape[1]
        mine = pack_candidates(val, idx, cnt)                             # (W * cap, P): block r = my shard's lists for rank r's rows
        # all-to-all: rank r receives only the W x cap lists of ITS rows (W * cap * P ints per rank and step; an all-gather of `mine`
        # delivered W times that and every rank kept 1 / W of it)
        own = torch.empty_like(mine)
        if self.comm is not None:
            self.comm.all_to_all(own, mine)
        elif self.world == 1:
            own.copy_(mine)
        else:
            dist.all_to_all_single(own, mine, group=self.group)
        return unpack_candidates(own.view(self.world, self.cap, mine.shape[1]), S)

    def agree(self, error: bool, done: bool, device):
        """One all-reduce(MAX) of (error, done, active) flags at the start of a step: a rank that found an error in its inputs, or ran
        out of batches, must not leave the others waiting in the step's collectives -- every rank learns it and raises.  Returns
        (any rank has an error, any rank is done, any rank is active)."""
        t = torch.tensor([1 if error else 0, 1 if done else 0, 0 if done else 1], dtype=torch.int32, device=device)
        if self.comm is not None:
            self.comm.all_reduce_max_i32(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        e, d, a = (bool(v) for v in t.tolist())
        return e, d, a

    def any_flag(self, flag: bool, device) -> bool:
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
        if self.comm is not None:
            self.comm.all_reduce_max_i32(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(t.item())

    def gather_blocks(self, aug_local: torch.Tensor, n_inst: int) -> torch.Tensor:
        """aug_local (W * cap, M_s + 1) fp16 rows of this rank's instance range (last column = the constant 1) -> the full (cap, M + 1)
        rows of this rank's queries.  Ranges differ by at most one instance: blocks are padded to the widest for the collective."""
        widths = [shard_range(n_inst, r, self.world)[1] - shard_range(n_inst, r, self.world)[0] for r in range(self.world)]
        wmax = max(widths)
        mine = torch.zeros((aug_local.shape[0], wmax), dtype=aug_local.dtype, device=aug_local.device)
        mine[:, :aug_local.shape[1] - 1] = aug_local[:, :-1]
        out = torch.empty((self.world * mine.shape[0], wmax), dtype=mine.dtype, device=mine.device)
        self._all_gather(out, mine)
        own = out.view(self.world, mine.shape[0], wmax)[:, self.rank * self.cap:(self.rank + 1) * self.cap]
        full = torch.cat([own[r, :, :w] for r, w in enumerate(widths)] +
                         [torch.ones((self.cap, 1), dtype=aug_local.dtype, device=aug_local.device)], dim=1)
        return full


def sharded_candidates(ex: ShardExchange, detn: torch.Tensor, match_fn):
    """Steps 1-3 of the sharded match for one batch of this rank's R normalised query rows.  match_fn(all_rows (W * cap, D)) ->
    (val, idx, cnt, aug_local) is the rank-local match of ALL ranks' rows against this rank's instance range with GLOBAL indices
    (`match.match_topk(..., index_base=lo)` on the GPU).  Returns the merged host lists of this rank's rows -- (R, W * S) values,
    (R, W * S) indices, (R,) counts, ready for `assign.assign_candidates` -- and aug_local for the full-row fall-back."""
    R = detn.shape[0]
    allq = ex.gather_queries(detn)
    val, idx, cnt, aug_local = match_fn(allq)
    gv, gi, gc = ex.gather_candidates(val, idx, cnt)
    val_h, idx_h, cnt_h = merge_lists_host(gv[:, :R].cpu().numpy(), gi[:, :R].cpu().numpy(), gc[:, :R].cpu().numpy())
    return val_h, idx_h, cnt_h, aug_local


def allgather_similarity_blocks(local_block: torch.Tensor, n_inst: int, group=None) -> torch.Tensor:
    """local_block: (Nq, hi - lo) similarities of this rank's instance range for the SAME query rows on every rank -> (Nq, n_inst) on
    every rank.  Shards may differ by one column, so blocks are padded to the widest shard for the collective."""
    world = dist.get_world_size(group)
    widths = [shard_range(n_inst, r, world)[1] - shard_range(n_inst, r, world)[0] for r in range(world)]
    wmax = max(widths)
    nq = local_block.shape[0]
    padded = torch.zeros((nq, wmax), dtype=local_block.dtype, device=local_block.device)
    padded[:, :local_block.shape[1]] = local_block
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return torch.cat([o[:, :w] for o, w in zip(out, widths)], dim=1)


def augment_half(sims: torch.Tensor) -> np.ndarray:
    """[sims | 1] as float16 (utils/similarity_volume.py:13-18) on the host."""
    s = sims.detach().cpu().numpy()
    aug = np.ones((s.shape[0], s.shape[1] + 1), dtype=np.float16)
    aug[:, :-1] = s
    return aug


def fitness_rmse_from_d2(d2: torch.Tensor, job_sizes):
    """evaluate_registration's (fitness, inlier rmse) per job from per-point squared distances (+inf = no correspondence)."""
    fit, rmse = [], []
    off = 0
    for n in job_sizes:
        d = d2[off:off + n]
        off += n
        ok = torch.isfinite(d)
        c = int(ok.sum().item())
        fit.append(c / n if n else 0.0)
        rmse.append(float(torch.sqrt(d[ok].double().sum() / c).item()) if c else 0.0)
    return np.array(fit), np.array(rmse)


def evaluate_sharded(local_d2: torch.Tensor, job_sizes, group=None, comm: "RcclComm" = None):
    """Whole-memory evaluation with the memory clouds sharded by instance range: `local_d2` = this rank's
    `registration.evaluate_points` output against the points it owns; the nearest memory point overall is the minimum over the ranks
    (all-reduce MIN over RCCL / gloo: 4 bytes per transformed detected point and candidate), then fitness / rmse as on one GPU."""
    d2 = local_d2.clone()
    if comm is not None:
        comm.all_reduce_min(d2)
    elif dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(d2, op=dist.ReduceOp.MIN, group=group)
    return fitness_rmse_from_d2(d2, job_sizes)

"""Multi-GPU layout of the path: query frames are data-parallel (no communication); the memory
embeddings can additionally be sharded by instance range, in which case every rank computes its block of the
closest-similarity matrix and the blocks are all-gathered (RCCL over xGMI; backend "nccl" on ROCm, "gloo" in the
CPU tests) before the exact assignment search."""
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


def allgather_similarity_blocks(local_block: torch.Tensor, n_inst: int, group=None) -> torch.Tensor:
    """local_block: (Nq, hi - lo) similarities of this rank's instance range -> (Nq, n_inst) on every rank.
    Shards may differ by one column, so blocks are padded to the widest shard for the collective."""
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


def evaluate_sharded(local_d2: torch.Tensor, job_sizes, group=None):
    """Whole-memory evaluation with the memory clouds sharded by instance range: `local_d2` = this rank's
    `registration.evaluate_points` output against the points it owns; the nearest memory point overall is the minimum over the ranks
    (all-reduce MIN over RCCL / gloo: 4 bytes per transformed detected point and candidate), then fitness / rmse as on one GPU."""
    d2 = local_d2.clone()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(d2, op=dist.ReduceOp.MIN, group=group)
    return fitness_rmse_from_d2(d2, job_sizes)

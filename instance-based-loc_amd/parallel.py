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

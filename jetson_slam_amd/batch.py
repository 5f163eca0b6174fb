"""Multi-GPU batch mode plumbing: independent stereo pairs are sharded over the ranks of one node (one process per GPU),
every rank uploads / processes only its own pairs, and the ONLY collective of the path is an all-gather of the per-pair
(N_left, N_right, N_matched) counts.  torch.distributed is used for the collective ("nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(n_pairs, rank, world):
    """contiguous block of pair indices owned by `rank` (sizes differ by at most one)"""
    base, rem = divmod(n_pairs, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def max_shard(n_pairs, world):
    return (n_pairs + world - 1) // world


def all_gather_counts(local_counts, n_pairs, group=None):
    """local_counts: int32 tensor [n_local, 3] of this rank's pairs (device tensor for nccl, CPU tensor for gloo).
    Returns the [n_pairs, 3] table in global pair order on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_counts.clone()
    rank = dist.get_rank(group)
    m = max_shard(n_pairs, world)
    pad = torch.full((m, 3), -1, dtype=torch.int32, device=local_counts.device)
    pad[: local_counts.shape[0]] = local_counts
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    rows = []
    for r in range(world):
        a, b = shard_range(n_pairs, r, world)
        rows.append(parts[r][: b - a])
    del rank
    return torch.cat(rows, 0)

"""Multi-GPU plumbing: members are independent, so the ensemble shards as
contiguous blocks with NO data-path collective; the only exchange is one
all-reduce of per-year sufficient statistics {count, sum, sumsq, min, max}
(SURVEY.md 8e).  Backend "nccl" is RCCL on ROCm; "gloo" is used by CPU tests."""
import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous block partition: -> (offset, count) of this rank."""
    base, rem = divmod(n_total, world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def allreduce_stats(stats, dist=None, force=False):
    """stats: torch tensor [..., 5] = count,sum,sumsq,min,max per year.
    In-place reduction across ranks (SUM for the first three, MIN, MAX) with ONE
    collective: every rank writes its block into its own slot of a zero-filled
    [world, ...] buffer and a single SUM all-reduce hands every rank all the blocks
    (x + 0 is exact); the five statistics are then combined locally in rank order, so the
    result is bit-identical on every rank.  world x 44 KB per variable, latency-bound."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return stats  # (force: run the collective even in a world of one)
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    buf = torch.zeros((world,) + tuple(stats.shape), dtype=stats.dtype, device=stats.device)
    buf[rank] = stats
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    # ranks are combined in rank order on every rank: the result is bit-identical everywhere
    acc = buf[0].clone()
    for r in range(1, world):
        acc[..., 0:3] += buf[r][..., 0:3]
        acc[..., 3] = torch.minimum(acc[..., 3], buf[r][..., 3])
        acc[..., 4] = torch.maximum(acc[..., 4], buf[r][..., 4])
    stats.copy_(acc)
    return stats


def stats_numpy(x):
    """x: [n_years, n_members] -> [n_years, 5]"""
    return np.stack([np.full(x.shape[0], x.shape[1], dtype=np.float64), x.sum(1),
                     (x * x).sum(1), x.min(1), x.max(1)], axis=1)


def finalize(stats):
    """-> mean, std, min, max per year from reduced sufficient statistics."""
    n, s, s2 = stats[..., 0], stats[..., 1], stats[..., 2]
    mean = s / n
    var = np.maximum(s2 / n - mean * mean, 0.0)
    return mean, np.sqrt(var), stats[..., 3], stats[..., 4]

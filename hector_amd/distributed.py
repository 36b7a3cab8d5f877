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


def allreduce_stats(stats, dist=None):
    """stats: torch tensor [..., 5] = count,sum,sumsq,min,max per year.
    In-place all-reduce across ranks (SUM for the first three, MIN, MAX)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return stats
    s3 = stats[..., 0:3].contiguous()
    mn = stats[..., 3].contiguous()
    mx = stats[..., 4].contiguous()
    dist.all_reduce(s3, op=dist.ReduceOp.SUM)
    dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    stats[..., 0:3] = s3
    stats[..., 3] = mn
    stats[..., 4] = mx
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

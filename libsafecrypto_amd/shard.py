"""Multi-GPU sharding of independent units (polynomials / signatures / samples): contiguous
slices of the batch, one per rank, no data-path collective."""


def shard_range(total, rank, world):
    """[lo, hi) of the units rank `rank` of `world` owns; sizes differ by at most one and the
    slices tile [0, total) exactly once, in rank order (so the host can concatenate results)."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError('bad shard arguments')
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value, dist=None):
    """the bench's timing reduction: MAX of a python float over ranks (identity without a group)"""
    if dist is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    import torch
    backend = dist.get_backend()
    dev = 'cuda' if backend == 'nccl' else 'cpu'
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

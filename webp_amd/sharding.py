"""Multi-GPU policy for this path: independent images / animation frames are split into contiguous
index ranges, one per rank (one process per GPU).  There is no exchange step, hence no collective on
the data path; ranks only meet for timing."""


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous [lo, hi) of `n_items` owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)

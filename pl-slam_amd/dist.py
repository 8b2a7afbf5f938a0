"""Multi-GPU plumbing: frames are independent, so a batch is sharded in contiguous chunks, one per
rank (one process per GPU), and the fixed-stride result records are collected with ONE all_gather per tensor
(RCCL over xGMI on GPUs; the same code runs on gloo for the CPU tests).  No collective is used inside the front end."""


def shard_range(total, rank, world):
    """Contiguous chunk of `total` frames owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_records(local, world, dist):
    """local: dict name -> tensor [b_local, ...] with the SAME b_local on every rank (pad the last shard).
    Returns dict name -> tensor [world * b_local, ...] in rank order."""
    out = {}
    for name, t in local.items():
        t = t.contiguous()
        g = t.new_empty((world * t.shape[0],) + tuple(t.shape[1:]))
        dist.all_gather_into_tensor(g, t)
        out[name] = g
    return out

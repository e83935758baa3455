"""Batch sharding across ranks (one process per GPU).  The hot path partitions by item: every encrypt /
keygen / decrypt call is independent, so a batch is cut into contiguous blocks, each rank
runs its block on its own GPU, and fixed-size result records are gathered -- there is no data-path
collective.  Works with any torch.distributed backend (nccl = RCCL on the GPUs, gloo in the CPU tests)."""


def shard_range(n_items, rank, world):
    """Contiguous block [lo, hi) of rank `rank`: sizes differ by at most one, blocks are in rank order."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def gather_records(local_records, dst=0):
    """Gathers per-rank lists of result records (bytes) to `dst` in item order; other ranks get None."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local_records)
    world, rank = dist.get_world_size(), dist.get_rank()
    bucket = [None] * world if rank == dst else None
    dist.gather_object(list(local_records), bucket, dst=dst)
    if rank != dst:
        return None
    out = []
    for part in bucket:
        out.extend(part)
    return out


def max_over_ranks(value):
    """The bench's max-over-ranks of an elapsed time (the only collective on the timed path)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():        # a one-rank group still runs the collective (RABE_FORCE_DIST)
        return float(value)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

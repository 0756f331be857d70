"""Multi-GPU sharding of a batch: one process per GPU, contiguous item ranges, no data-path
collective; the only exchange is the final gather of the result bytes (RCCL over xGMI when the
process group's backend is "nccl", gloo in the CPU tests).  SURVEY 8(e)."""


def shard_bounds(n, rank, world):
    """Contiguous, balanced partition of n items: rank r owns [lo, hi)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_bytes(local, n_total, group=None):
    """All-gather the per-rank result rows (uint8 tensor, first dimension = items of this rank's
    shard) into the full batch order on every rank.  Shards may differ by one item, so shards are
    padded to the largest one for the collective and trimmed after it."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1:
        return local
    rank = dist.get_rank(group)
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    longest = max(hi - lo for lo, hi in sizes)
    row = tuple(local.shape[1:])
    padded = torch.zeros((longest,) + row, dtype=local.dtype, device=local.device)
    lo, hi = sizes[rank]
    if local.shape[0] != hi - lo:
        raise ValueError("local shard has the wrong number of items")
    padded[: hi - lo] = local
    out = torch.empty((world * longest,) + row, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    out = out.view((world, longest) + row)
    return torch.cat([out[r, : sizes[r][1] - sizes[r][0]] for r in range(world)], dim=0)

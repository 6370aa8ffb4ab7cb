"""Multi-GPU sharding of a frame batch: frames are independent
once state is explicit, so rank g of W takes the contiguous range
[g*N/W, (g+1)*N/W) and no collective runs during compute.  The only exchange
step is the optional gather of PCM onto one rank (BASELINE config 5), done with
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests)."""


def shard_range(n, rank, world):
    """Contiguous, balanced split: the first n % world ranks get one extra frame."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n, world):
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def gather_pcm(pcm_local, n_total, dst=0):
    """Gather per-rank PCM shards ([n_local, ...] tensors) onto `dst` in frame order.
    Shards may differ by one frame, so every rank pads to the largest shard."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = shard_sizes(n_total, world)
    pad = max(sizes)
    buf = pcm_local
    if pcm_local.shape[0] < pad:
        buf = torch.cat([pcm_local, pcm_local.new_zeros((pad - pcm_local.shape[0],) + tuple(pcm_local.shape[1:]))])
    # ship raw bytes: gloo has no int16, and the payload is opaque to the collective
    raw = buf.contiguous().view(torch.uint8)
    out = [torch.empty_like(raw) for _ in range(world)] if rank == dst else None
    dist.gather(raw, out, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o.view(buf.dtype).view(buf.shape)[:s] for o, s in zip(out, sizes)])

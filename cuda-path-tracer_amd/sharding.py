"""Sample-index sharding for multi-GPU renders.

Every (pixel, sample index) pair owns an independent random stream (the sampler is seeded from the
pixel index and the 1-based sample index, /root/reference/src/kernels/integrator/path.cu:92), so a
render of S samples per pixel splits into disjoint sample-index sets, one per GPU, with the scene
and BVH replicated.  Each GPU accumulates plain sums; one sum-reduce of the W*H*4 float buffers
(RCCL over xGMI via torch.distributed) gives the image.  No other exchange is needed.
"""


def sample_block(rank, world_size, samples_per_rank, first=1):
    """Contiguous block of 1-based sample indices owned by `rank` when every rank renders
    `samples_per_rank` samples (weak scaling): returns (begin, count)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    if samples_per_rank < 0 or first < 1:
        raise ValueError("bad sample block")
    return first + rank * samples_per_rank, samples_per_rank


def split_samples(total_samples, world_size, first=1):
    """Partition of `total_samples` sample indices over `world_size` ranks (strong scaling):
    list of (begin, count), remainders going to the lowest ranks."""
    if world_size < 1 or total_samples < 0:
        raise ValueError("bad partition")
    q, r = divmod(total_samples, world_size)
    out, begin = [], first
    for rank in range(world_size):
        n = q + (1 if rank < r else 0)
        out.append((begin, n))
        begin += n
    return out


def reduce_sum_to_root(buffer, dst=0):
    """Sum-reduce a torch tensor (device or host) to rank `dst` over the default process group
    and return it; a no-op without an initialised group."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(buffer, dst=dst, op=dist.ReduceOp.SUM)
    return buffer


def finish_image(sum_buffer):
    """sum buffer (.., 4) with alpha = number of samples  ->  mean radiance, alpha 1."""
    count = sum_buffer[..., 3:4].clip(min=1.0) if hasattr(sum_buffer, "clip") else sum_buffer[..., 3:4]
    out = sum_buffer / count
    return out

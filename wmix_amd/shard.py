"""Multi-GPU plumbing of the hot path: streams shard by contiguous ranges, state never
leaves its GPU, and the only exchange is the shared AEC far-end packet, broadcast from the ingest rank."""


def stream_range(n_total, rank, world):
    """Contiguous [lo, hi) of the global stream ids owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_far(far, dist, src=0, async_op=False):
    """far: the int16 far-end packet tensor (same shape on every rank); rank `src` holds the data.
    async_op=True returns the collective's work handle (or None when there is nothing to exchange): the caller launches
    the stages that do not need the far-end (NS) and calls .wait() before the AEC, so the 320-byte broadcast hides
    behind the noise suppressor."""
    if dist is not None and dist.is_initialized():  # a group of one rank still makes the call (bench.py's WMIX_BENCH_FORCE_DIST)
        # neither RCCL nor gloo has an int16 type: broadcast the same bytes as uint8 (a view, no copy)
        import torch
        work = dist.broadcast(far.view(torch.uint8), src=src, async_op=async_op)
        return work if async_op else far
    return None if async_op else far

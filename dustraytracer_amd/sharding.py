"""Framebuffer sharding across the GPUs of one node (new: the reference is single-device).

A renderer with drt_renderer_set_shard(stripe_rows, rank, world) owns the rows y with
(y // stripe_rows) % world == rank and stores them compactly, stripe after stripe.  The helpers here
are host logic only: which rows a rank owns, and the gather of the rank-local framebuffers to rank 0
with torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
Re-assembling the image on rank 0 is a HIP kernel (drt_assemble_shards).
"""
import numpy as np


def shard_row_map(height, stripe_rows, rank, world):
    """Global row index of every local row of `rank`, in local (storage) order."""
    rows = [y for y in range(height) if (y // stripe_rows) % world == rank]
    return np.asarray(rows, dtype=np.int64)


def padded_rows(height, stripe_rows, world):
    """Rows every rank's send buffer must hold so that the gather is uniform."""
    return max(len(shard_row_map(height, stripe_rows, r, world)) for r in range(world))


def gather_shards(local, gathered, rank, dst=0):
    """dist.gather of the rank-local framebuffer [padded_rows, W, C] into gathered[world, padded_rows, W, C] on dst."""
    import torch.distributed as dist
    dist.gather(local, list(gathered.unbind(0)) if rank == dst else None, dst=dst)

"""Test-only transport for the sharding tests: the row gather over torch.distributed (gloo on CPU).  The product's
transport is the RCCL leg of the C ABI (wfa_rccl_allgather_counts + wfa_rccl_gather_rows); this restates its two steps."""

import numpy as np


def gather_rows_torch(rows: np.ndarray, root: int = 0):
    """Gather structured rows with torch.distributed (gloo on CPU / nccl = RCCL on GPU).
    Returns the list of per-rank arrays on `root`, None elsewhere.  Same two steps as the RCCL leg
    of the C ABI: a count all-gather, then padded byte payloads."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([len(rows)], dtype=torch.int64))
    counts = [int(c.item()) for c in counts]
    item = rows.dtype.itemsize
    width = max(counts) * item
    payload = torch.zeros(max(width, 1), dtype=torch.uint8)
    if len(rows):
        payload[: len(rows) * item] = torch.from_numpy(np.frombuffer(rows.tobytes(), dtype=np.uint8).copy())
    out = [torch.zeros_like(payload) for _ in range(world)] if rank == root else None
    dist.gather(payload, out, dst=root)
    if rank != root:
        return None
    return [np.frombuffer(out[r].numpy().tobytes()[: counts[r] * item], dtype=rows.dtype).copy() for r in range(world)]

"""Channel sharding for one-process-per-GPU runs.

Channels are independent (one reference handle = one channel), so the only exchange is the coefficient tables at
setup: rank 0 designs them (host C code) and broadcasts them -- over RCCL/xGMI when the process group is `nccl`, over
gloo in the CPU tests.  There is no steady-state collective.
"""
import numpy as np


def channel_range(total_channels, rank, world_size):
    """Contiguous slice [lo, hi) of the channel axis owned by `rank` (remainder spread over the first ranks)."""
    if not (0 <= rank < world_size) or total_channels < 0:
        raise ValueError("bad rank / world_size / channel count")
    base, rem = divmod(total_channels, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_table(table, src=0, device=None):
    """Broadcast a float64 coefficient table (taps, biquad rows, L x Q resample matrix) from `src` to every rank.

    table: numpy array on `src` (ignored elsewhere, but its SHAPE must be known: pass an array of the right
    shape on every rank, e.g. zeros).  Returns the table as numpy float64 on every rank.
    Single-process runs (no initialised process group) return the input unchanged.
    """
    import torch
    import torch.distributed as dist

    arr = np.ascontiguousarray(table, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return arr
    t = torch.from_numpy(arr.copy())
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def broadcast_shape(shape, src=0, device=None, ndim=2):
    """Agree on a table's shape first (e.g. Q depends on the window and ratio designed on rank 0)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tuple(int(v) for v in shape)
    vals = list(shape) + [0] * (ndim - len(shape))
    t = torch.tensor(vals, dtype=torch.int64)
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=src)
    return tuple(int(v) for v in t.cpu().tolist())


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (the bench's step time)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

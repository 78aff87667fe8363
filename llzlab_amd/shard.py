"""Channel sharding for one-process-per-GPU runs.

Channels are independent (one reference handle = one channel), so the only exchange is the coefficient tables at
setup: rank 0 designs them (host C code) and broadcasts them -- over RCCL/xGMI when the process group is `nccl`, over
gloo in the CPU tests.  There is no steady-state collective.
"""
import numpy as np

# how the setup collectives travelled in this process: "rccl" (device tensors over the nccl backend), "gloo" (host tensors), and
# the reason if the device path was given up (one failure on ANY rank switches every later call of EVERY rank to the host group)
_state = {"cpu_group": None, "device_ok": True, "used": [], "error": None}


def use_cpu_group(group):
    """A gloo process group for the timing barrier / max-reduce and as the fallback of the table broadcasts."""
    _state["cpu_group"] = group


def transport():
    """e.g. "rccl", "gloo", or "gloo (rccl failed: ...)" -- for the bench line"""
    used = "+".join(dict.fromkeys(_state["used"])) or "none"
    return used + (f" (rccl failed: {_state['error']})" if _state["error"] else "")


def _device_collective(t, device, fn):
    """run fn(tensor) on a device copy of t over the default (nccl) group.  Whether the device path worked is decided by ALL
    ranks together (a MIN all-reduce of an ok flag over the host group): if any rank failed -- communicator creation or the
    collective itself -- every rank repeats the operation on the host group and stays there; a rank never falls back alone
    (it would enter a collective no peer joins).  Without a host group to agree on, a failure is raised."""
    import torch
    import torch.distributed as dist

    if device is None:                                   # the default group carries host tensors (gloo runs, CPU tests)
        h = t.clone()
        fn(h, None)
        _state["used"].append("gloo")
        return h
    cpu_group = _state["cpu_group"]

    def agreed(err):
        """True when NO rank reported an error (MIN all-reduce over the host group); without a host group: this rank's own"""
        if cpu_group is None:
            return err is None
        ok = torch.tensor([1 if err is None else 0], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=cpu_group)
        return int(ok.item()) == 1

    if _state["device_ok"]:
        d, err = None, None
        try:                                             # phase 1, local: the tensor reaches this rank's device
            d = t.to(device)
        except Exception as e:  # noqa: BLE001
            err = str(e).splitlines()[0][:160]
        if agreed(err):
            try:                                         # phase 2: the collective itself (communicator creation included)
                fn(d, None)
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                err = str(e).splitlines()[0][:160]
            if agreed(err):
                _state["used"].append("rccl")
                return d.cpu()
        if cpu_group is None:
            raise RuntimeError("device collective failed and there is no host group to fall back to: " + str(err))
        _state["device_ok"] = False
        _state["error"] = err or "a peer rank's device collective failed"
    if cpu_group is None:
        raise RuntimeError("device collectives were given up and there is no host group to fall back to")
    h = t.clone()
    fn(h, cpu_group)
    _state["used"].append("gloo")
    return h


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
    return _device_collective(t, device, lambda x, g: dist.broadcast(x, src=src, group=g)).numpy()


def broadcast_shape(shape, src=0, device=None, ndim=2):
    """Agree on a table's shape first (e.g. Q depends on the window and ratio designed on rank 0)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tuple(int(v) for v in shape)
    vals = list(shape) + [0] * (ndim - len(shape))
    t = torch.tensor(vals, dtype=torch.int64)
    t = _device_collective(t, device, lambda x, g: dist.broadcast(x, src=src, group=g))
    return tuple(int(v) for v in t.tolist())


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (the bench's step time)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    if _state["cpu_group"] is not None:                      # timing bookkeeping stays on the host group
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_state["cpu_group"])
        return float(t.item())
    t = _device_collective(t, device, lambda x, g: dist.all_reduce(x, op=dist.ReduceOp.MAX, group=g))
    return float(t.item())

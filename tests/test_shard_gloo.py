"""Multi-rank host logic on CPU: world_size 2 over gloo.  Channel ranges partition the channel axis, the tap-table
broadcast delivers rank 0's table everywhere, the step time is the max over ranks."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_channel_range_partitions():
    from llzlab_amd.shard import channel_range
    for total, world in ((8192, 8), (4096, 1), (1000, 3), (5, 8), (0, 2)):
        spans = [channel_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        for a, b in zip(spans, spans[1:]):
            assert a[1] == b[0]
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        channel_range(10, 3, 3)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from llzlab_amd import filters, shard
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        # rank 0 designs (host C code); the others only know the length
        taps = filters.fir_design("lpf", 257, 0.1, 0.0, filters.KAISER) if rank == 0 else np.zeros(257)
        got = shard.broadcast_table(taps)
        shape = shard.broadcast_shape((3, 134) if rank == 0 else (0, 0))
        mat = np.arange(3 * 134, dtype=np.float64).reshape(3, 134) if rank == 0 else np.zeros(shape)
        gotm = shard.broadcast_table(mat)
        t = shard.max_over_ranks(1.0 + rank)
        lo, hi = shard.channel_range(8192, rank, world)
        # the bench's layout: a host-side gloo group for the bookkeeping, device tensors for the tables; here the device path
        # must fail on every rank (no GPU in this process) and the table still arrives, over the host group, with the failure
        # reported
        if not __import__("torch").cuda.is_available():
            shard.use_cpu_group(dist.new_group(backend="gloo"))
            # ONE rank fails (rank 1 has no such device; rank 0's "device" is the CPU, which works): the fallback must be a
            # decision of all ranks -- rank 0 may not stay on the device path while rank 1 waits in the host group
            again = shard.broadcast_table(taps, device="cpu" if rank == 0 else "cuda:0")
            assert again.tobytes() == got.tobytes() and "rccl failed" in shard.transport(), shard.transport()
            assert shard._state["device_ok"] is False                  # on BOTH ranks, also the one whose own step worked
            assert shard.max_over_ranks(5.0 - rank, device="cuda:0") == 5.0
            shape2 = shard.broadcast_shape((7, 9) if rank == 0 else (0, 0), device="cuda:0")
            assert shape2 == (7, 9)
        q.put((rank, got.tobytes(), shape, float(gotm.sum()), t, lo, hi))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_max_over_two_ranks():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from llzlab_amd import filters
    ref = filters.fir_design("lpf", 257, 0.1, 0.0, filters.KAISER).tobytes()
    for rank, taps, shape, msum, t, lo, hi in res:
        assert taps == ref                      # bit-identical table on every rank
        assert shape == (3, 134) and msum == float(np.arange(3 * 134).sum())
        assert t == 2.0                         # max over ranks
    assert (res[0][5], res[0][6], res[1][5], res[1][6]) == (0, 4096, 4096, 8192)


def test_single_process_helpers_are_identity():
    from llzlab_amd import shard
    a = np.arange(5.0)
    assert np.array_equal(shard.broadcast_table(a), a)
    assert shard.broadcast_shape((2, 3)) == (2, 3)
    assert shard.max_over_ranks(3.5) == 3.5

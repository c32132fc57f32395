"""Voice sharding across the GPUs of one node -- the host side of the C ABI's "sharded render" (include/graphaudio_hip.h).

Every voice chain (source -> resample -> biquads -> gain -> convolver) is independent until the destination's input mix
(AudioNodeInput.cs:121-132), so the voices are split into contiguous ranges, one per rank (one context per GPU, ranks =
processes or threads), and the only exchange is ONE sum of the destination bus per render call, done by the library with
RCCL over xGMI (ga_render_reduce).  What the host has to do is (1) build the same graph with its own voices on every rank
and (2) carry the 128-byte communicator id from rank 0 to the others; `exchange_comm_id` does (2) over a torch.distributed
process group of ANY backend (gloo in the CPU tests and in bench.py: no collective library besides RCCL inside the product).
"""
from __future__ import annotations

import ctypes as C

from ._capi import product_api


def shard_range(total: int, world: int, rank: int, _api=None):
    """Contiguous [begin, end) of the `total` voices owned by `rank` (ga_shard_range: sizes differ by at most one)."""
    api = _api or product_api()
    first, count = C.c_int64(), C.c_int64()
    code = api.shard_range(int(total), int(world), int(rank), C.byref(first), C.byref(count))
    if code != 0:
        raise ValueError("bad world/rank")
    return first.value, first.value + count.value


def exchange_comm_id(ctx, rank: int, world: int):
    """Rank 0 creates the communicator id, everybody receives it (torch.distributed must be initialised when world > 1)."""
    if world == 1:
        return None
    import torch.distributed as dist
    box = [ctx.CommUniqueId() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def init_sharded(ctx, rank: int, world: int):
    """ga_comm_init on this rank's context (collective)."""
    ctx.CommInit(exchange_comm_id(ctx, rank, world), world, rank)

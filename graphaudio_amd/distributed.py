"""Voice sharding across the GPUs of one node.

Every voice chain (source -> resample -> biquads -> gain -> convolver) is independent until the destination's input
mix (AudioNodeInput.cs:121-132), so the voices are split into contiguous ranges, one per rank (one process per GPU),
and the only exchange is one sum-reduce of the destination bus per render call: float32 [channels][frames], RCCL over
xGMI on the GPU box (torch.distributed backend "nccl"), gloo in the CPU tests.
"""
from __future__ import annotations


def shard_range(total: int, world: int, rank: int):
    """Contiguous [begin, end) of the `total` voices owned by `rank` (last rank takes the remainder)."""
    if world <= 0 or rank < 0 or rank >= world:
        raise ValueError("bad world/rank")
    per = total // world
    begin = rank * per
    end = total if rank == world - 1 else begin + per
    return begin, end


def reduce_bus(bus, dst: int = 0):
    """Sum the per-rank destination buses onto `dst` (in place).  `bus` is a torch tensor [channels, frames]."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(bus, dst=dst, op=dist.ReduceOp.SUM)
    return bus

"""Sample-per-pixel sharding across GPUs: one process per GPU, disjoint stratum ranges, one reduce.

Passes of ``Camera::render`` (``src/camera.rs:185-198``) are independent once the generator is
keyed by (seed, pixel, stratum); the only state they share is the additive accumulator
(``:172-173``, ``:216``) and the pass count (``:199``).  So the s_u-major stratum list is cut into
contiguous ranges, each rank renders its range into its own accumulator, and the accumulators are
summed onto rank 0 with a single reduce (RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import List, Tuple


def stratum_ranges(total: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous split of ``range(total)`` over ``world`` ranks; sizes differ by at most one."""
    if world <= 0:
        raise ValueError("world size must be positive")
    q, r = divmod(total, world)
    out, b = [], 0
    for k in range(world):
        e = b + q + (1 if k < r else 0)
        out.append((b, e))
        b = e
    return out


def reduce_accumulator(accum, dst: int = 0):
    """Sums the (W*H*3,) float32 accumulators of all ranks onto ``dst`` (in place); returns ``accum``.
    A no-op when torch.distributed is not initialised."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum

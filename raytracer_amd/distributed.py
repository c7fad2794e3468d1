"""Splitting the sample budget over GPUs and combining the radiance accumulators.

Samples are independent and `aggregate_image += sample` (src/renderer.rs:135-137) is a commutative sum, so
the path shards by SAMPLE INDEX: every rank renders a contiguous block of samples of every pixel into its own
full-frame accumulator, and one all-reduce (RCCL over xGMI on GPUs, gloo on CPU) adds the accumulators.  The
random stream is keyed on (pixel, global sample index), so the set of samples is the same for any number of
ranks; only the order of the float additions differs.
"""
from __future__ import annotations

from typing import Callable, Tuple


def sample_range(total_samples: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(begin, count) of rank's contiguous block; the first `total % world` ranks get one extra sample."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world size")
    base, extra = divmod(total_samples, world_size)
    begin = rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)


def render_sharded(render_range: Callable[[int, int, "object"], None], total_samples: int, accum,
                   world_size: int = 1, rank: int = 0, group=None, first_sample: int = 0):
    """Renders this rank's share of samples [first_sample, first_sample + total_samples) into `accum` (a torch
    tensor of un-normalised sums, zeroed by the caller) via `render_range(begin, count, accum)` and sums the
    accumulators of all ranks in place.  Every rank returns the full sum."""
    begin, count = sample_range(total_samples, world_size, rank)
    if count:
        render_range(first_sample + begin, count, accum)
    if world_size > 1:
        import torch.distributed as dist

        dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
    return accum

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


def allreduce_sum(contexts, device_pointers, count: int) -> None:
    """`rt_allreduce_sum`: one host process driving several GPUs (one `Context` each) sums their accumulators
    (`device_pointers[i]` = address of `count` floats on `contexts[i]`'s device) in place with RCCL.  The
    reduction is enqueued on every context's stream; synchronise the contexts before reading."""
    import ctypes as C

    from . import _ffi as F

    lib = F.load_library()
    n = len(contexts)
    if n != len(device_pointers):
        raise ValueError("one buffer per context")
    ctxs = (C.c_void_p * n)(*[c.handle for c in contexts])
    bufs = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in device_pointers])
    F.check(lib, lib.rt_allreduce_sum(ctxs, n, bufs, count))


def render_node(contexts, scenes, camera, settings):
    """`rt_render_node`: one pass over the GPUs of a node from this process; `scenes[i]` is the scene created on
    `contexts[i]`.  Returns the un-normalised sums, shape (H, W, 3)."""
    import ctypes as C

    import numpy as np

    from . import _ffi as F

    lib = F.load_library()
    n = len(contexts)
    if n != len(scenes):
        raise ValueError("one scene per context")
    st = settings.to_c() if hasattr(settings, "to_c") else settings
    ctxs = (C.c_void_p * n)(*[c.handle for c in contexts])
    scs = (C.c_void_p * n)(*[s.handle for s in scenes])
    out = np.zeros((st.image_height, st.image_width, 3), dtype=np.float32)
    F.check(lib, lib.rt_render_node(ctxs, scs, n, C.byref(camera), C.byref(st), out.ctypes.data))
    return out

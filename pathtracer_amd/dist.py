"""Multi-GPU rendering: one process per GPU, tiles dealt across ranks, one integer reduce.

The reference runs its tiles on a thread pool and adds each finished tile into a shared
fixed-point accumulator under a mutex (pathtracer.cpp:568-584, 407-427); integer addition makes
the result independent of the order tiles finish in (pathtracer.h:80-83).  Across GPUs the same
property carries the design: every rank owns a private accumulator for the tiles it was dealt,
and ONE exchange at the end -- a sum all-reduce of int64[3*W*H] and of the per-pixel sample
counts over RCCL (torch.distributed backend "nccl"; "gloo" on CPU for tests) -- produces on every
rank exactly the frame a single process rendering all tiles would have.

Ray-query batches (ptx_trace_closest / ptx_trace_any) shard by ray range with no exchange at all.
"""
import numpy as np


def deal_tiles(tiles, rank, world):
    """Round-robin over the reference's centre-out tile order: neighbouring tiles cost about the same,
    so every rank gets an even mix of cheap border and expensive centre tiles."""
    return tiles[rank::world]


def ray_range(n, rank, world):
    """[begin, end) of the contiguous slice of an n-ray batch that `rank` traces."""
    per = (n + world - 1) // world
    return min(n, rank * per), min(n, (rank + 1) * per)


def reduce_frame(accum, samples, group=None):
    """In-place sum over ranks of the fixed-point accumulator and the sample counts.

    accum: torch int64 tensor [3*W*H]; samples: torch int32 tensor [W*H] holding the uint32 counts
    (counts stay far below 2^31).  Device tensors with the nccl backend, CPU tensors with gloo."""
    import torch
    import torch.distributed as dist
    assert accum.dtype == torch.int64 and samples.dtype == torch.int32
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if accum.is_cuda and dist.get_backend(group) == "gloo":
            # gloo is the CPU rehearsal backend: stage device tensors through host memory
            a, s = accum.cpu(), samples.cpu()
            dist.all_reduce(a, op=dist.ReduceOp.SUM, group=group)
            dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
            accum.copy_(a)
            samples.copy_(s)
        else:
            dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
            dist.all_reduce(samples, op=dist.ReduceOp.SUM, group=group)
    return accum, samples


def render_sharded(scene, camera, tiles, seed, rank, world, device=None, group=None, timings=None):
    """Renders this rank's share of `tiles` on its GPU and reduces: returns (accum, samples) torch tensors
    holding the complete frame on every rank.  `timings` (a dict, optional) receives this rank's `compute_s` (its tiles
    rendered, device idle again), `reduce_s` (the frame exchange) and `tiles` (how many it was dealt)."""
    import time
    import torch
    from . import abi
    npix = camera.film_width * camera.film_height
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    accum = torch.zeros(npix * 3, dtype=torch.int64, device=dev)
    samples = torch.zeros(npix, dtype=torch.int32, device=dev)
    mine = np.ascontiguousarray(deal_tiles(np.asarray(tiles), rank, world))
    t0 = time.perf_counter()
    if len(mine):
        scene.render_tiles(camera, mine, seed, accum, samples, flags=abi.PTX_ASYNC)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    out = reduce_frame(accum, samples, group)
    if timings is not None:
        torch.cuda.synchronize(dev)
        timings.update(compute_s=t1 - t0, reduce_s=time.perf_counter() - t1, tiles=int(len(mine)))
    return out

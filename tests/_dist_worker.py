"""One rank of the world_size-2 CPU rehearsal of the multi-GPU render path (gloo backend).

The rank renders the tiles pathtracer_amd.dist.deal_tiles gives it -- with the CPU checker standing
in for the GPU, since this runs where there is none -- reduces with pathtracer_amd.dist.reduce_frame
and rank 0 saves the frame.  Also shards a ray batch with dist.ray_range."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    out = sys.argv[1]
    import torch
    import torch.distributed as dist
    from pathtracer_amd import capi, dist as pdist, synth
    from _oracle import Oracle
    import scenes
    from test_dist_gloo import make_camera, SEED, FILM

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sb = scenes.instanced_scene()
    cam = make_camera()
    w, h, spp = FILM
    tiles = capi.make_tiles(w, h, spp, 99)
    sc = Oracle().scene(sb.desc())
    acc, smp = sc.render_tiles(cam, np.ascontiguousarray(pdist.deal_tiles(tiles, rank, world)), 1, SEED, nthreads=2)
    acc_t, smp_t = torch.from_numpy(acc), torch.from_numpy(smp.view(np.int32))
    pdist.reduce_frame(acc_t, smp_t)
    # ray batches shard by contiguous ranges; gather the pieces to check coverage
    rays = synth.rays(5001)
    b, e = pdist.ray_range(len(rays), rank, world)
    hits = sc.trace_closest(rays[b:e], nthreads=2)["hits"]
    pieces = [None] * world
    dist.all_gather_object(pieces, (b, e, hits.tobytes()))
    if rank == 0:
        np.savez(out, acc=acc_t.numpy(), smp=smp_t.numpy().view(np.uint32), ranges=np.array([(p[0], p[1]) for p in pieces]),
                 hits=np.frombuffer(b"".join(p[2] for p in pieces), dtype=np.uint8))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

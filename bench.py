#!/usr/bin/env python3
"""Benchmark of the ray-query hot path on MI355X.

Workload (the synthetic ray-batch configuration, the one the >= 1 Gray/s closest-hit target is quoted on):
16,777,216 incoherent rays against a 1,000,000-triangle procedural soup whose BVH is built
by the host builder (max_leaf_size 4).  One "step" is one closest-hit pass over the whole ray
batch, inputs and outputs resident in HBM.  With --gpus N every rank traces its own batch of
the same size against its own replica of the scene (weak scaling, no data-path collective).

Prints ONE JSON line on rank 0.  `value` is the whole-job closest-hit rate in Mrays/s.

Secondary legs in the same line (never part of `value`): the occlusion-query rate, the rate of the
reference-order traversal, and renders of the reference's cbox.js3d (256x256, 16 spp, depth 4) and
refract_cow_cbox.js3d (1024x1024, 256 spp, depth 8) from the committed scene fixtures; with --gpus N
the render tiles are dealt across ranks and the frame is summed with one RCCL all-reduce.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak


def cpu_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def cpu_baseline_and_counts(verts, idx, nodes, order, rays, target_seconds=12.0):
    """Times the CPU checker on a bounded prefix of the same ray batch and counts the nodes and
    triangles a reference-semantics traversal touches (for the algorithmic-bytes figure)."""
    from _oracle import Oracle, Ref
    from pathtracer_amd import abi
    threads = cpu_threads()
    out = {}
    oracle = Oracle()
    sb = abi.SceneBuilder()
    scene = oracle.mesh_scene(sb.mesh_desc(verts, idx[order], nodes=nodes))
    # visit counts: reference semantics, 16k-ray prefix
    probe = rays[:16384]
    res = scene.trace_closest(probe, nthreads=threads, counters=True)
    nodes_per_ray = float(res["counters"][0]) / len(probe)
    tris_per_ray = float(res["counters"][1]) / len(probe)
    kind = "port"
    run = lambda r: scene.trace_closest(r, nthreads=threads)
    if Ref.available():
        ref = Ref()
        rmesh = ref.mesh(verts, idx[order], id_coded=True, nodes=nodes)   # the builder's tree injected into the reference BVH
        run = lambda r: rmesh.hit(r, nthreads=threads)
        kind = "reference"
    t0 = time.perf_counter()
    run(rays[:32768])
    dt = time.perf_counter() - t0
    n = int(min(len(rays), max(32768, 32768 * target_seconds / max(dt, 1e-3))))
    n = (n // 4096) * 4096
    t0 = time.perf_counter()
    got = run(rays[:n])
    dt = time.perf_counter() - t0
    out["sample_hits"] = got["hits"]   # (distance, primitive position in tree order) of the timed sample, for the caller's parity check
    out["cpu_baseline"] = {"value": round(n / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": kind,
                           "sample": f"first {n} rays of the same 16M-ray batch, closest-hit, BVH<Triangle>::hit on the same 1M-triangle tree"}
    out["nodes_per_ray"] = nodes_per_ray
    out["tris_per_ray"] = tris_per_ray
    return out


def instanced_cow_scene(n):
    """Config 5: n^3 instances of the reference's 5,856-triangle cow mesh (10.1 M instanced triangles at n = 12)."""
    import scenes
    return scenes.instanced_cows(n)


def cpu_render_baseline(sb, cam, w, h, spp, target_seconds=4.0):
    """The CPU checker's Pathtracer::render on a bounded slice of the same render: the first tiles of the reference's
    centre-out order at a reduced sample count (the reference's cost is linear in samples and pixels), per-tile
    mt19937 streams as in the reference, one tile per host thread."""
    from _oracle import Oracle
    from pathtracer_amd import capi
    threads = cpu_threads()
    scene = Oracle().scene(sb.desc())
    slice_spp = min(spp, 8)
    cam.film_samples = slice_spp
    tiles = capi.make_tiles(w, h, slice_spp, 4660)
    first = tiles[:min(len(tiles), threads)]
    t0 = time.perf_counter()
    scene.render_tiles(cam, first, 0, 0, nthreads=threads)
    dt = time.perf_counter() - t0
    n_tiles = min(len(tiles), threads * int(max(1, min(16, round(target_seconds / max(dt, 1e-3))))))
    used = tiles[:n_tiles]
    t0 = time.perf_counter()
    scene.render_tiles(cam, used, 0, 0, nthreads=threads)
    dt = time.perf_counter() - t0
    cam.film_samples = spp
    paths = int(((used["x_end"].astype(np.int64) - used["x_begin"]) * (used["y_end"].astype(np.int64) - used["y_begin"]) * (used["s_end"].astype(np.int64) - used["s_begin"])).sum())
    scene.close()
    return {"value": round(paths / dt / 1e6, 4), "unit": "Mspp/s", "cores": threads, "kind": "port",
            "sample": f"first {n_tiles} of {len(tiles)} tiles in the reference's centre-out order at {slice_spp} spp ({paths} paths)"}


def render_leg(name, w, h, spp, depth, rank, world, dev, dist, cpu=False):
    """One render of a committed reference scene; returns (paths, wall seconds incl. the frame reduce, mean radiance,
    CPU baseline of the same render or None)."""
    import torch
    from pathtracer_amd import abi, capi, dist as pdist
    if name == "instanced_cows":
        sb, cam = instanced_cow_scene(12)
    else:
        g = np.load(os.path.join(ROOT, "tests", "golden", f"scene_{name}.npz"), allow_pickle=False)
        sb, cams = abi.builder_from_arrays(g)
        cam = cams["Camera_instance"]
    cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth = w, h, spp, depth
    cam.aspect_ratio = w / float(h)
    scene = capi.Scene(sb.desc(), device=dev.index, keep=sb)
    tiles = capi.make_tiles(w, h, spp, 4660)
    warm = capi.make_tiles(w, h, min(spp, 2), 1)
    pdist.render_sharded(scene, cam, warm, 1, rank, world, dev)      # allocations, code load
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc, smp = pdist.render_sharded(scene, cam, tiles, 4660, rank, world, dev)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        wt = torch.tensor([wall], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)
        wall = float(wt.item())
    assert int(smp.min().item()) == spp and int(smp.max().item()) == spp
    mean = float((acc.double().sum() / (1 << 24) / (3.0 * w * h * spp)).item())
    scene.close()
    return w * h * spp, wall, mean, (cpu_render_baseline(sb, cam, w, h, spp) if cpu else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rays", type=int, default=1 << 24)
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-render", action="store_true", help="skip the render legs")
    ap.add_argument("--no-4k", action="store_true", help="skip the 3840x2160, 1024 spp render leg (about 30 s on one GPU)")
    ap.add_argument("--rehearse", action="store_true",
                    help="run the multi-rank code path on fewer GPUs than ranks: gloo collectives, rank r on device r %% device_count (a functional check, not a measurement)")
    args = ap.parse_args()

    # Exactly one line goes to stdout, the result: native libraries print there too (RCCL announces its path
    # at start-up), so file descriptor 1 points at stderr until the result is written.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from pathtracer_amd import abi, capi, synth

    def note(msg):
        print(f"[bench r{os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.rehearse:
        local_rank = local_rank % torch.cuda.device_count()
    if args.gpus > 1 or world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank)

    note("process group ready" if dist is not None else "single process")
    # ---- workload: same scene on every rank, a different ray batch per rank
    verts, idx = synth.soup(args.tris)
    t0 = time.perf_counter()
    bvh = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4)
    build_s = time.perf_counter() - t0
    nodes, order = bvh.nodes(), bvh.order()
    scene = capi.Scene.from_mesh(verts, idx[order], nodes=nodes, device=local_rank)
    scene.set_stream(torch.cuda.current_stream().cuda_stream)
    rays_h = synth.rays(args.rays, key=synth.RAY_KEY + 7919 * rank)
    shadow_h = synth.shadow_rays(args.rays, key=synth.SHADOW_KEY + 7919 * rank)
    rays = torch.from_numpy(rays_h.view(np.float32).reshape(-1, 8)).to(dev)
    shadow = torch.from_numpy(shadow_h.view(np.float32).reshape(-1, 8)).to(dev)
    hits_ref = torch.empty((args.rays, 2), dtype=torch.int32, device=dev)
    hits = torch.empty((args.rays, 2), dtype=torch.int32, device=dev)
    occ_ref = torch.empty(args.rays, dtype=torch.uint8, device=dev)
    occ = torch.empty(args.rays, dtype=torch.uint8, device=dev)
    n = args.rays

    note(f"scene resident: {len(nodes)} nodes, build {build_s:.2f} s")

    def closest(mode, out):
        scene.trace_closest_device(rays, out, flags=mode | abi.PTX_ASYNC, n=n)

    def anyhit(mode, out):
        scene.trace_any_device(shadow, out, flags=mode | abi.PTX_ASYNC, n=n)

    # ---- correctness of the fast modes on the whole batch, outside the timed region
    closest(abi.PTX_TRAVERSE_REFERENCE, hits_ref)
    closest(abi.PTX_TRAVERSE_CULLED, hits)
    anyhit(abi.PTX_TRAVERSE_REFERENCE, occ_ref)
    anyhit(abi.PTX_TRAVERSE_CULLED, occ)
    torch.cuda.synchronize()
    mismatch_closest = int((hits != hits_ref).any(dim=1).sum().item())
    mismatch_any = int((occ != occ_ref).sum().item())
    hit_fraction = float((hits_ref[:, 1] != -1).float().mean().item())
    headline_mode = abi.PTX_TRAVERSE_CULLED if mismatch_closest == 0 else abi.PTX_TRAVERSE_REFERENCE
    if mismatch_closest and rank == 0:   # leave evidence for whoever debugs it
        bad = torch.nonzero((hits != hits_ref).any(dim=1)).flatten()[:16].cpu().numpy()
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "closest_mismatch.json"), "w") as f:
            json.dump({"rays": [int(b) for b in bad], "culled": hits[bad].cpu().numpy().tolist(), "reference": hits_ref[bad].cpu().numpy().tolist(),
                       "ray_data": rays_h[bad].view(np.float32).reshape(-1, 8).tolist()}, f)

    note(f"parity of the culled order on {n} rays: {mismatch_closest} closest-hit and {mismatch_any} occlusion mismatches")

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
            w = torch.tensor([wall], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            wall = float(w.item())
        return wall, e0.elapsed_time(e1) / steps   # whole-job wall seconds, per-launch device ms (HIP events)

    wall, step_ms = timed(lambda: closest(headline_mode, hits), args.steps, args.warmup)
    ms_per_step = wall * 1e3 / args.steps
    value = n * world * args.steps / wall / 1e6
    # device time of the dominant kernel alone (k_mesh_persistent), one HIP event pair per launch
    # recorded by the library on the stream it launches on, over the launches of the timed region
    kt = scene.kernel_times_ms(min(64, args.steps))
    kernel_ms = float(kt.mean()) if len(kt) else step_ms

    note(f"timed region done: {value:.1f} Mrays/s")
    # secondary figures (not part of `value`)
    _, ref_ms = timed(lambda: closest(abi.PTX_TRAVERSE_REFERENCE, hits_ref), max(1, args.steps // 3), 1)
    _, cul_ms = timed(lambda: closest(abi.PTX_TRAVERSE_CULLED, hits), max(1, args.steps // 3), 1)
    _, any_ms = timed(lambda: anyhit(abi.PTX_TRAVERSE_CULLED if mismatch_any == 0 else abi.PTX_TRAVERSE_REFERENCE, occ), args.steps, 1)

    result = {
        "metric": "Mrays/sec closest-hit", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (REHEARSAL: ranks share GPUs, gloo collectives -- not a measurement)" if args.rehearse else ""),
        "config": {"workload": f"config 3: {n} incoherent rays per GPU vs {args.tris}-triangle soup BVH (max_leaf 4), closest-hit (t, primitive id)",
                   "traversal": "culled" if headline_mode == abi.PTX_TRAVERSE_CULLED else "reference", "rays_per_gpu": n,
                   "triangles": args.tris, "bvh_nodes": int(len(nodes)), "bvh_build_s": round(build_s, 3)},
        "parity": {"culled_vs_reference_mode_mismatches": mismatch_closest, "anyhit_culled_vs_reference_mismatches": mismatch_any,
                   "rays_checked": n, "hit_fraction": round(hit_fraction, 4)},
        "step_device_ms": round(step_ms, 4), "dominant_kernel": "k_mesh_persistent<culled, closest>", "dominant_kernel_ms": round(kernel_ms, 4),
        "reference_mode_mrays_per_s": round(n / ref_ms / 1e3, 2),
        "culled_mode_mrays_per_s": round(n / cul_ms / 1e3, 2),
        "anyhit_mrays_per_s": round(n / any_ms / 1e3, 2),
    }

    # ---- culled-traversal work per ray (device counters, 64k-ray prefix): what the kernel really fetches
    work = None
    if rank == 0:
        st = scene.trace_closest_stats(rays_h[:65536], flags=abi.PTX_TRAVERSE_CULLED).astype(np.float64).mean(axis=0)
        work_bytes = 32 + 8 + 4 + 64 * st[0] + 48 * st[2]      # ray, hit, order index, 64-B two-child nodes, 48-B triangles
        work = {"interior_nodes_per_ray": round(st[0], 2), "leaves_per_ray": round(st[1], 2), "tris_per_ray": round(st[2], 2),
                "bytes_per_ray": round(work_bytes, 1)}

    renders = []
    if not args.no_render:
        legs = [("cbox", 256, 256, 16, 4, "config 1: cbox.js3d 256x256, 16 spp, depth 4"),
                ("refract_cow_cbox", 1024, 1024, 256, 8, "config 2: refract_cow_cbox.js3d 1024x1024, 256 spp, depth 8")]
        legs.append(("instanced_cows", 1024, 1024, 64, 8, "config 5: 1728 instances of the 5856-triangle cow (10.1 M instanced triangles), Lambertian, 1024x1024, 64 spp, depth 8"))
        if not args.no_4k and not args.rehearse:
            legs.append(("refract_cow_cbox", 3840, 2160, 1024, 8, "config 4: refract_cow_cbox.js3d 3840x2160, 1024 spp, depth 8, tiles dealt across the GPUs"))
        for name, w, h, spp, depth, label in legs:
            note(f"render leg {name} {w}x{h} {spp} spp")
            want_cpu = rank == 0 and world == 1 and not args.no_cpu and not label.startswith("config 4")   # config 4 is config 2's scene
            paths, rwall, mean, cpu = render_leg(name, w, h, spp, depth, rank, world, dev, dist, cpu=want_cpu)
            renders.append({"workload": label, "paths": paths, "wall_s": round(rwall, 4), "Mspp_per_s": round(paths / rwall / 1e6, 2),
                            "mean_radiance": round(mean, 5), "frame_reduce": (("rccl" if dist.get_backend() == "nccl" else dist.get_backend()) + " all_reduce int64+int32") if world > 1 else "none"})
            if cpu:
                renders[-1]["cpu_baseline"] = cpu
    result["renders"] = renders
    if len(renders) >= 2:
        result["render_Mspp_per_s"] = renders[1]["Mspp_per_s"]     # config 2, the one-GPU render configuration
    if len(renders) >= 4:
        result["render_4k_Mspp_per_s"] = renders[3]["Mspp_per_s"]  # config 4, strong scaling across the GPUs

    if rank == 0:
        extra = {}
        if world == 1 and not args.no_cpu:
            extra = cpu_baseline_and_counts(verts, idx, nodes, order, rays_h)
            result["cpu_baseline"] = extra["cpu_baseline"]
            # the CPU side's answers on its sample against the GPU's for the same rays: distance bits and primitive
            cpu_hits = extra["sample_hits"]
            gpu_hits = hits[:len(cpu_hits)].cpu().numpy()
            cpu_prim = cpu_hits["prim"].astype(np.uint32).view(np.int32)
            bad = (cpu_hits["distance"].view(np.int32) != gpu_hits[:, 0]) | (cpu_prim != gpu_hits[:, 1])
            result["parity"]["cpu_" + extra["cpu_baseline"]["kind"] + "_vs_gpu_mismatches"] = int(bad.sum())
            result["parity"]["cpu_rays_checked"] = int(len(cpu_hits))
        if extra:
            # algorithmic bytes per ray: ray in + hit out + 32 B per node box tested + 36 B per triangle
            # tested by a reference-semantics traversal of the canonical binary tree
            b_ray = 32 + 8 + 32 * extra["nodes_per_ray"] + 36 * extra["tris_per_ray"]
            achieved = b_ray * n / (kernel_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                traffic = json.load(open(tpath)).get("closest_culled_hbm_bytes_per_launch")
            work_gbs = work["bytes_per_ray"] * n / (kernel_ms * 1e-3) / 1e9
            result["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                                  "kernel": "k_mesh_persistent<culled, closest>", "kernel_ms": round(kernel_ms, 4), "bytes_per_ray": round(b_ray, 1),
                                  "nodes_per_ray": round(extra["nodes_per_ray"], 2), "tris_per_ray": round(extra["tris_per_ray"], 2),
                                  "culled_traversal": dict(work, GBps=round(work_gbs, 1), frac=round(work_gbs / HBM_PEAK_GBS, 4)),
                                  "note": "achieved = algorithmic bytes of a reference-semantics traversal (32 ray + 8 hit + 32/node box test + 36/triangle, "
                                          "visit counts from the CPU checker) per launch / dominant-kernel time; the culled traversal that produces the same "
                                          "results fetches far less (culled_traversal), and the 91 MB working set stays in L2 / Infinity Cache, so neither "
                                          "figure is DRAM traffic -- see traffic"}
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    scene.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Benchmark of the ray-query hot path on MI355X.

Workload (the synthetic ray-batch configuration, the one the >= 1 Gray/s closest-hit target is quoted on):
16,777,216 incoherent rays against a 1,000,000-triangle procedural soup whose BVH is built
by the host builder (max_leaf_size 4).  One "step" is one closest-hit pass over the whole ray
batch, inputs and outputs resident in HBM.  With --gpus N every rank traces its own batch of
the same size against its own replica of the scene (weak scaling, no data-path collective).

Prints ONE JSON line on rank 0.  `value` is the whole-job closest-hit rate in Mrays/s.

Secondary legs in the same line (never part of `value`): the occlusion-query rate and the rate of the reference-order
traversal; the same queries against two 10 M-triangle meshes whose trees live in HBM rather than in the caches
(`closest_10m`: the 1,728 cows of the instanced grid baked into one mesh, and the soup generator at 10 M triangles);
ray queries against the instanced scene in both orders; and renders of the reference's cbox.js3d (256x256, 16 spp,
depth 4), refract_cow_cbox.js3d (1024x1024, 256 spp, depth 8), the cow grid as 1,728 instances and as one flattened
mesh (1024x1024, 64 spp, depth 8), and refract_cow_cbox at 3840x2160 x 1024 spp; with --gpus N the render tiles are
dealt across ranks and the frame is summed with one RCCL all-reduce.  Every leg carries a `roofline` block
(pathtracer_amd/roofline.py) and, at N = 1, a `cpu_baseline`.
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

def online_cores():
    """Host cores this process may run on."""
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return max(1, os.cpu_count() or 1)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_thread_counts():
    """Thread counts the CPU baselines are timed with: every core the process may run on, and -- when that is more
    than 16 -- also 16, the CPU share a one-GPU job is given on a shared host.  The faster one is reported as the
    baseline, the other beside it."""
    n = online_cores()
    return [n] if n <= 16 else [n, 16]


def cpu_baseline_and_counts(verts, idx, nodes, order, rays, target_seconds=12.0, what="the same 16M-ray batch, closest-hit, BVH<Triangle>::hit on the same 1M-triangle tree"):
    """Times the CPU side on a bounded prefix of the same ray batch (the reference's own BVH<Triangle>::hit with the
    tree injected where oracle/_ref/libptref.so is present, else the CPU checker) and counts the nodes and triangles a
    reference-semantics traversal touches (for the algorithmic-bytes figure)."""
    from _oracle import Oracle, Ref
    from pathtracer_amd import abi
    counts = cpu_thread_counts()
    threads = counts[0]
    out = {}
    oracle = Oracle()
    sb = abi.SceneBuilder()
    ordered = np.ascontiguousarray(idx[order])
    scene = oracle.mesh_scene(sb.mesh_desc(verts, ordered, nodes=nodes))
    # visit counts: reference semantics, 16k-ray prefix
    probe = rays[:16384]
    res = scene.trace_closest(probe, nthreads=threads, counters=True)
    nodes_per_ray = float(res["counters"][0]) / len(probe)
    tris_per_ray = float(res["counters"][1]) / len(probe)
    kind = "port"
    run = lambda r, t: scene.trace_closest(r, nthreads=t)
    rmesh = None
    if Ref.available():
        ref = Ref()
        rmesh = ref.mesh(verts, ordered, id_coded=True, nodes=nodes)   # the builder's tree injected into the reference BVH
        run = lambda r, t: rmesh.hit(r, nthreads=t)
        kind = "reference"
    measured = []
    for t in counts:
        t0 = time.perf_counter()
        run(rays[:65536], t)
        dt = time.perf_counter() - t0
        n = int(min(len(rays), max(65536, 65536 * (target_seconds / len(counts)) / max(dt, 1e-3))))
        n = (n // 4096) * 4096
        t0 = time.perf_counter()
        got = run(rays[:n], t)
        dt = time.perf_counter() - t0
        measured.append((n / dt / 1e6, t, n, got["hits"]))
    if rmesh is not None:
        rmesh.close()
    scene.close()
    rate, threads, n, hits_sample = max(measured, key=lambda m: m[0])
    out["sample_hits"] = hits_sample   # (distance, primitive position in tree order) of the timed sample, for the caller's parity check
    out["cpu_baseline"] = {"value": round(rate, 4), "unit": "Mrays/s", "cores": threads, "kind": kind,
                           "sample": f"first {n} rays of {what}",
                           "cpu_model": cpu_model(), "online_cores": online_cores(),
                           "other_thread_counts": [{"cores": t, "value": round(r, 4)} for r, t, _, _ in measured if t != threads]}
    out["nodes_per_ray"] = nodes_per_ray
    out["tris_per_ray"] = tris_per_ray
    return out


def cpu_render_baseline(sb, cam, w, h, spp, target_seconds=4.0, slice_div=None):
    """The CPU checker's Pathtracer::render on a bounded slice of the same render: the first tiles of the reference's
    centre-out order at a reduced sample count (the reference's cost is linear in samples and pixels), per-tile
    mt19937 streams as in the reference, one tile per host thread.  Also counts what the reference's recursion does per
    path on that slice: rays traced, nodes and triangles visited (the algorithmic-bytes figure of the render legs)."""
    from _oracle import Oracle
    from pathtracer_amd import capi
    scene = Oracle().scene(sb.desc())
    slice_spp = min(spp, 8)
    full = (w, h)
    if slice_div:   # a film 1/slice_div[0] the size in each direction at 1/slice_div[1] of the samples, same camera
        w, h, slice_spp = w // slice_div[0], h // slice_div[0], max(1, spp // slice_div[1])
        cam.film_width, cam.film_height = w, h
    cam.film_samples = slice_spp
    tiles = capi.make_tiles(w, h, slice_spp, 4660)
    measured = []
    counts = cpu_thread_counts()

    def tile_paths(ts):
        return int(((ts["x_end"].astype(np.int64) - ts["x_begin"]) * (ts["y_end"].astype(np.int64) - ts["y_begin"]) * (ts["s_end"].astype(np.int64) - ts["s_begin"])).sum())

    for threads in counts:
        first = tiles[:min(len(tiles), threads)]
        t0 = time.perf_counter()
        scene.render_tiles(cam, first, 0, 0, nthreads=threads)
        dt = time.perf_counter() - t0
        n_tiles = min(len(tiles), threads * int(max(1, min(16, round(target_seconds / len(counts) / max(dt, 1e-3))))))
        used = tiles[:n_tiles]
        t0 = time.perf_counter()
        _, _, st = scene.render_tiles(cam, used, 0, 0, nthreads=threads, stats=True)
        dt = time.perf_counter() - t0
        paths = tile_paths(used)
        measured.append((paths / dt / 1e6, threads, n_tiles, paths, st))
    cam.film_samples = spp
    cam.film_width, cam.film_height = full
    scene.close()
    rate, threads, n_tiles, paths, st = max(measured, key=lambda m: m[0])
    all_rays = float(st[0] + st[1])
    reference = {"rays_per_path": all_rays / paths, "nodes_per_ray": float(st[2]) / max(all_rays, 1.0), "tris_per_ray": float(st[3]) / max(all_rays, 1.0)}
    what = f"the same camera on a {w}x{h} film (1/{slice_div[0] ** 2} of the area)" if slice_div else "the same film"
    return {"value": round(rate, 4), "unit": "Mspp/s", "cores": min(threads, n_tiles), "kind": "port",
            "sample": f"{what}: first {n_tiles} of {len(tiles)} tiles in the reference's centre-out order at {slice_spp} spp ({paths} paths); the reference's cost is linear in pixels and samples (one tile per host thread, per-tile mt19937 streams)",
            "cpu_model": cpu_model(), "online_cores": online_cores(),
            "other_thread_counts": [{"cores": t, "value": round(r, 4)} for r, t, _, _, _ in measured if t != threads]}, reference


RENDER_SCENES = {}   # leg name -> (SceneBuilder, camera): built once per process (the flattened grid takes seconds)


def render_scene(name):
    """(SceneBuilder, camera) of a render leg: the committed descriptors of the reference's scene files, the cow grid as
    1,728 instances of the reference's cow mesh, or the same cows baked into one 10.1 M-triangle mesh."""
    from pathtracer_amd import abi
    if name not in RENDER_SCENES:
        import scenes
        if name == "instanced_cows":
            RENDER_SCENES[name] = scenes.instanced_cows(12)
        elif name == "flat_cows":
            RENDER_SCENES[name] = scenes.flattened_cow_scene(12)
        else:
            g = np.load(os.path.join(ROOT, "tests", "golden", f"scene_{name}.npz"), allow_pickle=False)
            sb, cams = abi.builder_from_arrays(g)
            RENDER_SCENES[name] = (sb, cams["Camera_instance"])
    return RENDER_SCENES[name]


def render_leg(name, w, h, spp, depth, rank, world, dev, dist, cpu=False):
    """One render; returns a dict: paths, wall seconds (max over ranks, frame reduce included), the compute / reduce
    split per rank, rays traced (device counters, all ranks), mean radiance, and at N = 1 the CPU baseline of the
    same render with the reference recursion's per-path work."""
    import torch
    from pathtracer_amd import capi, dist as pdist
    sb, cam = render_scene(name)
    cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth = w, h, spp, depth
    cam.aspect_ratio = w / float(h)
    t0 = time.perf_counter()
    scene = capi.Scene(sb.desc(), device=dev.index, keep=sb)
    create_s = time.perf_counter() - t0
    tiles = capi.make_tiles(w, h, spp, 4660)
    warm = capi.make_tiles(w, h, min(spp, 2), 1)
    pdist.render_sharded(scene, cam, warm, 1, rank, world, dev)      # allocations, code load
    torch.cuda.synchronize()
    c0 = scene.counters()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    timings = {}
    t0 = time.perf_counter()
    acc, smp = pdist.render_sharded(scene, cam, tiles, 4660, rank, world, dev, timings=timings)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    c1 = scene.counters()
    rays = int((c1.render_main_rays - c0.render_main_rays) + (c1.render_direct_rays - c0.render_direct_rays))
    per_rank = [dict(rank=rank, wall_s=round(wall, 4), compute_s=round(timings["compute_s"], 4), reduce_ms=round(timings["reduce_s"] * 1e3, 3), tiles=timings["tiles"], rays=rays)]
    if dist is not None:
        dist.barrier()
        wt = torch.tensor([wall], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)
        wall = float(wt.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
    assert int(smp.min().item()) == spp and int(smp.max().item()) == spp
    mean = float((acc.double().sum() / (1 << 24) / (3.0 * w * h * spp)).item())
    scene.close()
    out = {"paths": w * h * spp, "wall_s": wall, "mean_radiance": mean, "scene_create_s": create_s,
           "traced_rays": sum(r["rays"] for r in per_rank), "compute_s": max(r["compute_s"] for r in per_rank), "reduce_ms": max(r["reduce_ms"] for r in per_rank),
           "per_rank": per_rank, "cpu_baseline": None, "reference": None}
    if cpu:
        out["cpu_baseline"], out["reference"] = cpu_render_baseline(sb, cam, w, h, spp, slice_div=(8, 16) if w * h > (1 << 22) else None)
    return out


def ray_query_leg(label, verts, idx, n, extent, key_offset, rank, world, dev, dist, steps, warmup, timed, cpu_seconds=0.0, gpu_build_check=False, note=print):
    """Closest-hit and occlusion queries of `n` rays per GPU against one mesh: parity of the culled order against the
    reference order on the whole batch, timed rates, the dominant kernel's device time, the culled walk's visit counts,
    and (cpu_seconds > 0) the CPU baseline on a bounded prefix with its answers compared to the GPU's."""
    import torch
    from pathtracer_amd import abi, capi, synth
    out = {"triangles": int(len(idx))}
    if gpu_build_check:
        t0 = time.perf_counter()
        bvh = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4)
        out["bvh_build_s"] = round(time.perf_counter() - t0, 3)
        nodes, order = bvh.nodes(), bvh.order()
        bvh.close()
        # the same tree from the GPU builder (what scene creation uses for meshes of this size): time, and identity
        capi.Bvh(verts=verts[:64], indices=idx[:8] % 64, max_leaf_size=4, gpu_device=dev.index).close()   # code load
        t0 = time.perf_counter()
        gbvh = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4, gpu_device=dev.index)
        out["bvh_build_gpu_s"] = round(time.perf_counter() - t0, 4)
        out["bvh_gpu_build_identical_to_host"] = bool(gbvh.built_on_device and gbvh.nodes().tobytes() == nodes.tobytes() and np.array_equal(gbvh.order(), order))
        gbvh.close()
        t0 = time.perf_counter()
        scene = capi.Scene.from_mesh(verts, idx[order], nodes=nodes, device=dev.index)
    else:
        t0 = time.perf_counter()
        scene = capi.Scene.from_mesh(verts, idx, device=dev.index)      # the library builds the tree (GPU builder at this size)
        nodes, order = scene.mesh_nodes(), scene.mesh_order(len(idx))
        out["tree_built_on_device"] = scene.mesh_built_on_device()
    out["scene_create_s"] = round(time.perf_counter() - t0, 3)
    out["bvh_nodes"] = int(len(nodes))
    scene.set_stream(torch.cuda.current_stream().cuda_stream)
    rays_h = synth.rays(n, key=synth.RAY_KEY + key_offset + 7919 * rank, extent=extent)
    shadow_h = synth.shadow_rays(n, key=synth.SHADOW_KEY + key_offset + 7919 * rank, extent=extent)
    rays = torch.from_numpy(rays_h.view(np.float32).reshape(-1, 8)).to(dev)
    shadow = torch.from_numpy(shadow_h.view(np.float32).reshape(-1, 8)).to(dev)
    del shadow_h
    hits_ref = torch.empty((n, 2), dtype=torch.int32, device=dev)
    hits = torch.empty((n, 2), dtype=torch.int32, device=dev)
    occ_ref = torch.empty(n, dtype=torch.uint8, device=dev)
    occ = torch.empty(n, dtype=torch.uint8, device=dev)
    note(f"{label}: scene resident, {len(nodes)} nodes, {out['scene_create_s']:.2f} s")

    def closest(mode, dst):
        scene.trace_closest_device(rays, dst, flags=mode | abi.PTX_ASYNC, n=n)

    def anyhit(mode, dst):
        scene.trace_any_device(shadow, dst, flags=mode | abi.PTX_ASYNC, n=n)

    # ---- correctness of the fast modes on the whole batch, outside the timed region
    closest(abi.PTX_TRAVERSE_REFERENCE, hits_ref)
    closest(abi.PTX_TRAVERSE_CULLED, hits)
    anyhit(abi.PTX_TRAVERSE_REFERENCE, occ_ref)
    anyhit(abi.PTX_TRAVERSE_CULLED, occ)
    torch.cuda.synchronize()
    mismatch_closest = int((hits != hits_ref).any(dim=1).sum().item())
    mismatch_any = int((occ != occ_ref).sum().item())
    out["parity"] = {"culled_vs_reference_mode_mismatches": mismatch_closest, "anyhit_culled_vs_reference_mismatches": mismatch_any,
                     "rays_checked": n, "hit_fraction": round(float((hits_ref[:, 1] != -1).float().mean().item()), 4), "occluded_fraction": round(float(occ_ref.float().mean().item()), 4)}
    headline_mode = abi.PTX_TRAVERSE_CULLED if mismatch_closest == 0 else abi.PTX_TRAVERSE_REFERENCE
    out["traversal"] = "culled" if headline_mode == abi.PTX_TRAVERSE_CULLED else "reference"
    if mismatch_closest and rank == 0:   # leave evidence for whoever debugs it
        bad = torch.nonzero((hits != hits_ref).any(dim=1)).flatten()[:16].cpu().numpy()
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"closest_mismatch_{label.split(':')[0].replace(' ', '_')}.json"), "w") as f:
            json.dump({"rays": [int(b) for b in bad], "culled": hits[bad].cpu().numpy().tolist(), "reference": hits_ref[bad].cpu().numpy().tolist(),
                       "ray_data": rays_h[bad].view(np.float32).reshape(-1, 8).tolist()}, f)
    note(f"{label}: parity of the culled order on {n} rays: {mismatch_closest} closest-hit and {mismatch_any} occlusion mismatches")

    wall, step_ms = timed(lambda: closest(headline_mode, hits), steps, warmup)
    out["wall_s"], out["step_device_ms"] = wall, step_ms
    out["mrays_per_s"] = n * world * steps / wall / 1e6
    # device time of the dominant kernel alone (k_mesh_persistent), one HIP event pair per launch
    # recorded by the library on the stream it launches on, over the launches of the timed region
    kt = scene.kernel_times_ms(min(64, steps))
    out["kernel_ms"] = float(kt.mean()) if len(kt) else step_ms
    note(f"{label}: timed region done: {out['mrays_per_s']:.1f} Mrays/s")
    # secondary figures
    _, ref_ms = timed(lambda: closest(abi.PTX_TRAVERSE_REFERENCE, hits_ref), max(1, steps // 3), 1)
    _, cul_ms = timed(lambda: closest(abi.PTX_TRAVERSE_CULLED, hits), max(1, steps // 3), 1)
    _, any_ms = timed(lambda: anyhit(abi.PTX_TRAVERSE_CULLED if mismatch_any == 0 else abi.PTX_TRAVERSE_REFERENCE, occ), steps, 1)
    kt = scene.kernel_times_ms(min(64, steps))
    out["anyhit_kernel_ms"] = float(kt.mean()) if len(kt) else any_ms
    out["reference_mode_mrays_per_s"], out["culled_mode_mrays_per_s"], out["anyhit_mrays_per_s"] = n / ref_ms / 1e3, n / cul_ms / 1e3, n / any_ms / 1e3

    # ---- culled-traversal work per ray (device counters, 64k-ray prefix): what the kernel really fetches
    out["work"] = out["cpu"] = None
    if rank == 0:
        st = scene.trace_closest_stats(rays_h[:65536], flags=abi.PTX_TRAVERSE_CULLED).astype(np.float64).mean(axis=0)
        work_bytes = 32 + 8 + 4 + 64 * st[0] + 48 * st[2]      # ray, hit, order index, 64-B two-child nodes, 48-B triangles
        out["work"] = {"interior_nodes_per_ray": round(st[0], 2), "leaves_per_ray": round(st[1], 2), "tris_per_ray": round(st[2], 2), "bytes_per_ray": round(work_bytes, 1)}
        if cpu_seconds > 0:
            extra = cpu_baseline_and_counts(verts, idx, nodes, order, rays_h, target_seconds=cpu_seconds, what=f"the same {n}-ray batch, closest-hit, BVH<Triangle>::hit on the same {len(idx)}-triangle tree")
            # the CPU side's answers on its sample against the GPU's for the same rays: distance bits and primitive
            cpu_hits = extra.pop("sample_hits")
            gpu_hits = hits[:len(cpu_hits)].cpu().numpy()
            cpu_prim = cpu_hits["prim"].astype(np.uint32).view(np.int32)
            bad = (cpu_hits["distance"].view(np.int32) != gpu_hits[:, 0]) | (cpu_prim != gpu_hits[:, 1])
            out["parity"]["cpu_" + extra["cpu_baseline"]["kind"] + "_vs_gpu_mismatches"] = int(bad.sum())
            out["parity"]["cpu_rays_checked"] = int(len(cpu_hits))
            out["cpu"] = extra
    scene.close()
    del rays, shadow, hits, hits_ref, occ, occ_ref
    torch.cuda.empty_cache()
    return out


def scene_query_leg(name, n, extent, dev, timed, steps):
    """Closest-hit queries against a two-level scene (Aggregate::hit over instances) in both orders: the culled order
    (the replenishing two-level kernel) and the reference order, the API's default (one ray per lane)."""
    import torch
    from pathtracer_amd import abi, capi, synth
    sb, _ = render_scene(name)
    scene = capi.Scene(sb.desc(), device=dev.index, keep=sb)
    scene.set_stream(torch.cuda.current_stream().cuda_stream)
    rays = torch.from_numpy(synth.rays(n, key=synth.RAY_KEY + 20, extent=extent).view(np.float32).reshape(-1, 8)).to(dev)
    shadow = torch.from_numpy(synth.shadow_rays(n, key=synth.SHADOW_KEY + 20, extent=extent).view(np.float32).reshape(-1, 8)).to(dev)
    a, b = torch.empty((n, 2), dtype=torch.int32, device=dev), torch.empty((n, 2), dtype=torch.int32, device=dev)
    ia, ib = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
    occ_a, occ_b = torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
    scene.trace_closest_device(rays, a, inst=ia, flags=abi.PTX_TRAVERSE_REFERENCE, n=n)
    scene.trace_closest_device(rays, b, inst=ib, flags=abi.PTX_TRAVERSE_CULLED, n=n)
    scene.trace_any_device(shadow, occ_a, flags=abi.PTX_TRAVERSE_REFERENCE, n=n)
    scene.trace_any_device(shadow, occ_b, flags=abi.PTX_TRAVERSE_CULLED, n=n)
    torch.cuda.synchronize()
    bad = int(((a != b).any(dim=1) | (ia != ib)).sum().item())
    _, ref_ms = timed(lambda: scene.trace_closest_device(rays, a, flags=abi.PTX_TRAVERSE_REFERENCE | abi.PTX_ASYNC, n=n), max(1, steps // 3), 1)
    _, cul_ms = timed(lambda: scene.trace_closest_device(rays, b, flags=abi.PTX_TRAVERSE_CULLED | abi.PTX_ASYNC, n=n), steps, 1)
    _, any_ms = timed(lambda: scene.trace_any_device(shadow, occ_b, flags=abi.PTX_TRAVERSE_CULLED | abi.PTX_ASYNC, n=n), steps, 1)
    _, any_ref_ms = timed(lambda: scene.trace_any_device(shadow, occ_a, flags=abi.PTX_TRAVERSE_REFERENCE | abi.PTX_ASYNC, n=n), max(1, steps // 3), 1)
    out = {"workload": f"{n} incoherent rays vs the instanced cow grid (1,728 instances of the 5,856-triangle cow + 2 quads), Aggregate::hit", "rays": n,
           "culled_mode_mrays_per_s": round(n / cul_ms / 1e3, 2), "reference_mode_mrays_per_s": round(n / ref_ms / 1e3, 2),
           "anyhit_culled_mrays_per_s": round(n / any_ms / 1e3, 2), "anyhit_reference_mode_mrays_per_s": round(n / any_ref_ms / 1e3, 2),
           "culled_vs_reference_mode_mismatches": bad, "anyhit_culled_vs_reference_mismatches": int((occ_a != occ_b).sum().item()),
           "hit_fraction": round(float((a[:, 1] != -1).float().mean().item()), 4),
           "kernels": {"culled": "k_scene_persistent (replenishing, phase-parked)", "reference": "k_scene_persistent<reference order> (replenishing; every node and triangle the reference visits, in its order: exact by construction, the API default)"}}
    scene.close()
    return out


def refresh_roofline_blocks(line):
    """Every roofline block of a benchmark line again, from the newest counter summaries under profiles/."""
    from pathtracer_amd import roofline
    return roofline.refresh(line)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n):
    """`python bench.py --gpus N` from a plain shell: starts N fresh processes of this script, one rank per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before anything in this process has touched the GPU; relays
    rank 0's single result line and exits non-zero if any rank failed.  (Under torch.distributed.run the ranks
    already exist and this is not used.)"""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    # A rank that dies leaves the others waiting in a collective: once one has failed, the rest get 30 s, then are
    # killed (exactly the children this launcher started).
    deadline = None
    while any(pr.poll() is None for pr in procs):
        if deadline is None and any(pr.poll() not in (None, 0) for pr in procs):
            deadline = time.monotonic() + 30.0
        if deadline is not None and time.monotonic() > deadline:
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
        time.sleep(0.05)
    codes = [pr.wait() for pr in procs]
    failed = any(c != 0 for c in codes)
    reader.join(timeout=5.0)
    line = b"".join(chunks).decode()
    if failed:
        print(f"[bench] rank exit codes {codes}: failing", file=sys.stderr)
        sys.exit(1)
    results = [l for l in line.splitlines() if l.startswith("{")]
    if len(results) != 1:
        print(f"[bench] expected one result line from rank 0, got {len(results)}", file=sys.stderr)
        sys.exit(1)
    sys.stdout.write(results[0] + "\n")
    sys.stdout.flush()
    sys.exit(0)


def launch_check(args):
    """The launcher's plumbing without a GPU (what the CPU test runs): every rank joins a gloo group, the ranks are
    summed, rank 0 prints the line; --launch-check-fail-rank makes one rank exit non-zero."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([rank + 1], dtype=torch.int64)
    dist.all_reduce(t)
    ranks = [None] * world
    dist.all_gather_object(ranks, {"rank": rank, "local_rank": int(os.environ["LOCAL_RANK"]), "pid": os.getpid()})
    dist.barrier()
    dist.destroy_process_group()
    if rank == args.launch_check_fail_rank:
        sys.exit(3)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_joined": world, "rank_sum": int(t.item()), "ranks": ranks}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rays", type=int, default=1 << 24)
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-render", action="store_true", help="skip the render legs")
    ap.add_argument("--no-4k", action="store_true", help="skip the 3840x2160, 1024 spp render leg (about 7 s on one GPU)")
    ap.add_argument("--no-10m", action="store_true", help="skip the legs on the 10 M-triangle meshes (ray queries and the flattened-grid render)")
    ap.add_argument("--rehearse", action="store_true",
                    help="run the multi-rank code path on fewer GPUs than ranks: gloo collectives, rank r on device r %% device_count (a functional check, not a measurement)")
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--launch-check-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)      # does not return
    if args.launch_check:
        return launch_check(args)

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
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
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

    rank_devices = [{"rank": 0, "device": 0}]
    if dist is not None:
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        rank_devices = [None] * world
        props = torch.cuda.get_device_properties(local_rank)
        dist.all_gather_object(rank_devices, {"rank": rank, "device": local_rank, "name": props.name, "pci_bus_id": getattr(props, "pci_bus_id", None)})
    note("process group ready" if dist is not None else "single process")

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

    from pathtracer_amd import roofline
    want_cpu = rank == 0 and world == 1 and not args.no_cpu
    n = args.rays

    # ---- the headline workload: same scene on every rank, a different ray batch per rank
    verts, idx = synth.soup(args.tris)
    leg = ray_query_leg("config 3", verts, idx, n, 10.0, 0, rank, world, dev, dist, args.steps, args.warmup, timed, cpu_seconds=12.0 if want_cpu else 0.0, gpu_build_check=True, note=note)
    del verts, idx
    value, wall = leg["mrays_per_s"], leg["wall_s"]
    stress, stress_src = roofline.latest_profile("order_stress")
    result = {
        "metric": "Mrays/sec closest-hit", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(wall * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (REHEARSAL: ranks share GPUs, gloo collectives -- not a measurement)" if args.rehearse else ""),
        "rccl_ranks": dist.get_world_size() if dist is not None else 1, "collective_backend": (dist.get_backend() if dist is not None else "none"),
        "rank_devices": rank_devices,
        "config": {"workload": f"config 3: {n} incoherent rays per GPU vs {args.tris}-triangle soup BVH (max_leaf 4), closest-hit (t, primitive id)",
                   "traversal": leg["traversal"], "rays_per_gpu": n, "triangles": args.tris, "bvh_nodes": leg["bvh_nodes"], "bvh_build_s": leg["bvh_build_s"],
                   "bvh_build_gpu_s": leg["bvh_build_gpu_s"], "bvh_gpu_build_identical_to_host": leg["bvh_gpu_build_identical_to_host"]},
        "parity": leg["parity"],
        "step_device_ms": round(leg["step_device_ms"], 4), "dominant_kernel": "k_mesh_persistent<culled, closest>", "dominant_kernel_ms": round(leg["kernel_ms"], 4),
        "reference_mode_mrays_per_s": round(leg["reference_mode_mrays_per_s"], 2),
        "culled_mode_mrays_per_s": round(leg["culled_mode_mrays_per_s"], 2),
        "anyhit_mrays_per_s": round(leg["anyhit_mrays_per_s"], 2),
    }
    result["parity"]["culled_order_note"] = (
        "The culled order (the timed mode) is verified against the reference order, not exact by construction: 0 differences on every ray of this run; on rays "
        "constructed to lie IN the plane of a triangle (to about 1e-6 rad), where every determinant of Triangle::hit is rounding noise, it differs from the reference on "
        + (f"0 to {max(c['mismatches'] for c in stress['cases'])} of {stress['rays_per_case']} rays per adversarial set ({stress['total_mismatches']} of {stress['rays_per_case'] * len(stress['cases'])} in all)"
           if stress and stress.get("cases") else "up to a few thousand of 4.2 M rays per adversarial set")
        + f" ({stress_src or 'profiles/rNN_order_stress.json'}, pinned by tests/test_gpu_order_stress.py).  PTX_TRAVERSE_REFERENCE, the API's default, walks every triangle the reference walks and is "
        f"exact by construction: {round(leg['reference_mode_mrays_per_s'], 1)} Mrays/s on this workload.")
    if rank == 0 and leg["cpu"]:
        result["cpu_baseline"] = leg["cpu"]["cpu_baseline"]
        result["roofline"] = roofline.soup_1m_block(leg["kernel_ms"], n, leg["work"], leg["cpu"]["nodes_per_ray"], leg["cpu"]["tris_per_ray"])
    elif rank == 0:
        result["roofline"] = roofline.soup_1m_block(leg["kernel_ms"], n, leg["work"], None, None)

    # ---- the same queries where the tree lives in HBM: two meshes of 10 M triangles (not part of `value`)
    if not args.no_10m:
        import scenes
        result["closest_10m"] = {}
        for mesh, label, make, extent, key_offset in (
                ("cows10m", "config 5 (flattened): the 1,728 cows of the instanced grid baked into ONE 10,119,168-triangle mesh (tests/scenes.py flattened_cows(12))", lambda: scenes.flattened_cows(12), 14.4, 10),
                ("soup10m", "config 3 at ten times the triangles: the soup generator at 10,000,000 triangles (synth.soup)", lambda: synth.soup(10_000_000), 10.0, 10)):
            verts, idx = make()
            cpu_s = 6.0 if (want_cpu and mesh == "cows10m") else 0.0
            l10 = ray_query_leg(mesh, verts, idx, n, extent, key_offset, rank, world, dev, dist, args.steps, 1, timed, cpu_seconds=cpu_s, note=note)
            del verts, idx
            entry = {"workload": f"{label}; {n} incoherent rays per GPU (origins uniform in [0,{extent})^3, directions uniform), closest-hit and occlusion",
                     "triangles": l10["triangles"], "bvh_nodes": l10["bvh_nodes"], "tree_built_on_device": l10["tree_built_on_device"], "scene_create_s": l10["scene_create_s"],
                     "device_bytes": {"nodes": l10["bvh_nodes"] // 2 * 64, "triangle_geometry": l10["triangles"] * 48, "triangle_attributes": l10["triangles"] * 60},
                     "closest_mrays_per_s": round(l10["mrays_per_s"], 2), "closest_kernel_ms": round(l10["kernel_ms"], 4), "ms_per_step": round(l10["wall_s"] * 1e3 / args.steps, 4),
                     "anyhit_mrays_per_s": round(l10["anyhit_mrays_per_s"], 2), "anyhit_kernel_ms": round(l10["anyhit_kernel_ms"], 4),
                     "reference_mode_mrays_per_s": round(l10["reference_mode_mrays_per_s"], 2), "traversal": l10["traversal"], "parity": l10["parity"]}
            if rank == 0:
                cpu = l10["cpu"] or {}
                if cpu:
                    entry["cpu_baseline"] = cpu["cpu_baseline"]
                entry["roofline"] = roofline.mesh_10m_block(mesh, "closest", l10["kernel_ms"], n, l10["work"], cpu.get("nodes_per_ray"), cpu.get("tris_per_ray"))
                entry["roofline_occlusion"] = roofline.mesh_10m_block(mesh, "occlusion", l10["anyhit_kernel_ms"], n, None, None, None)
            result["closest_10m"][mesh] = entry
        result["closest_10m_mrays_per_s"] = result["closest_10m"]["cows10m"]["closest_mrays_per_s"]

    renders = []
    if not args.no_render:
        legs = [("cbox", 256, 256, 16, 4, "config 1: cbox.js3d 256x256, 16 spp, depth 4"),
                ("refract_cow_cbox", 1024, 1024, 256, 8, "config 2: refract_cow_cbox.js3d 1024x1024, 256 spp, depth 8"),
                ("instanced_cows", 1024, 1024, 64, 8, "config 5: 1728 instances of the 5856-triangle cow (10.1 M instanced triangles), Lambertian, 1024x1024, 64 spp, depth 8")]
        if not args.no_10m:
            legs.append(("flat_cows", 1024, 1024, 64, 8, "config 5 (flattened): the same cows baked into one 10,119,168-triangle mesh (BVH in HBM), Lambertian, 1024x1024, 64 spp, depth 8"))
        if not args.no_4k and not args.rehearse:
            legs.append(("refract_cow_cbox", 3840, 2160, 1024, 8, "config 4: refract_cow_cbox.js3d 3840x2160, 1024 spp, depth 8, tiles dealt across the GPUs"))
        for name, w, h, spp, depth, label in legs:
            note(f"render leg {name} {w}x{h} {spp} spp")
            r = render_leg(name, w, h, spp, depth, rank, world, dev, dist, cpu=want_cpu)
            entry = {"workload": label, "leg": name, "paths": r["paths"], "wall_s": round(r["wall_s"], 4), "Mspp_per_s": round(r["paths"] / r["wall_s"] / 1e6, 2),
                     "mean_radiance": round(r["mean_radiance"], 5), "traced_rays": r["traced_rays"], "scene_create_s": round(r["scene_create_s"], 3),
                     "frame_reduce": (("rccl" if dist.get_backend() == "nccl" else dist.get_backend()) + " all_reduce int64+int32") if world > 1 else "none"}
            if world > 1:
                entry.update(compute_s=r["compute_s"], reduce_ms=r["reduce_ms"], per_rank=r["per_rank"])
            if rank == 0:
                entry["roofline"] = roofline.render_block(name, r["paths"], r["wall_s"], r["traced_rays"], r["reference"])
            if r["cpu_baseline"]:
                entry["cpu_baseline"] = r["cpu_baseline"]
            renders.append(entry)
        if rank == 0 and world == 1 and not args.rehearse:
            note("ray queries against the instanced scene")
            result["scene_queries"] = scene_query_leg("instanced_cows", 1 << 22, 14.4, dev, timed, args.steps)
    result["renders"] = renders
    by_leg = {(r["leg"], r["paths"]): r for r in renders}
    if ("refract_cow_cbox", 1024 * 1024 * 256) in by_leg:
        result["render_Mspp_per_s"] = by_leg[("refract_cow_cbox", 1024 * 1024 * 256)]["Mspp_per_s"]     # config 2, the one-GPU render configuration
    if ("refract_cow_cbox", 3840 * 2160 * 1024) in by_leg:
        result["render_4k_Mspp_per_s"] = by_leg[("refract_cow_cbox", 3840 * 2160 * 1024)]["Mspp_per_s"]  # config 4, strong scaling across the GPUs

    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

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


def cpu_baseline_and_counts(verts, idx, nodes, order, rays, target_seconds=12.0):
    """Times the CPU checker on a bounded prefix of the same ray batch and counts the nodes and
    triangles a reference-semantics traversal touches (for the algorithmic-bytes figure)."""
    from _oracle import Oracle, Ref
    from pathtracer_amd import abi
    counts = cpu_thread_counts()
    threads = counts[0]
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
    run = lambda r, t: scene.trace_closest(r, nthreads=t)
    if Ref.available():
        ref = Ref()
        rmesh = ref.mesh(verts, idx[order], id_coded=True, nodes=nodes)   # the builder's tree injected into the reference BVH
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
    rate, threads, n, hits_sample = max(measured, key=lambda m: m[0])
    out["sample_hits"] = hits_sample   # (distance, primitive position in tree order) of the timed sample, for the caller's parity check
    out["cpu_baseline"] = {"value": round(rate, 4), "unit": "Mrays/s", "cores": threads, "kind": kind,
                           "sample": f"first {n} rays of the same 16M-ray batch, closest-hit, BVH<Triangle>::hit on the same 1M-triangle tree",
                           "cpu_model": cpu_model(), "online_cores": online_cores(),
                           "other_thread_counts": [{"cores": t, "value": round(r, 4)} for r, t, _, _ in measured if t != threads]}
    out["nodes_per_ray"] = nodes_per_ray
    out["tris_per_ray"] = tris_per_ray
    return out


def instanced_cow_scene(n):
    """Config 5: n^3 instances of the reference's 5,856-triangle cow mesh (10.1 M instanced triangles at n = 12)."""
    import scenes
    return scenes.instanced_cows(n)


def cpu_render_baseline(sb, cam, w, h, spp, target_seconds=4.0, slice_div=None):
    """The CPU checker's Pathtracer::render on a bounded slice of the same render: the first tiles of the reference's
    centre-out order at a reduced sample count (the reference's cost is linear in samples and pixels), per-tile
    mt19937 streams as in the reference, one tile per host thread."""
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
    for threads in counts:
        first = tiles[:min(len(tiles), threads)]
        t0 = time.perf_counter()
        scene.render_tiles(cam, first, 0, 0, nthreads=threads)
        dt = time.perf_counter() - t0
        n_tiles = min(len(tiles), threads * int(max(1, min(16, round(target_seconds / len(counts) / max(dt, 1e-3))))))
        used = tiles[:n_tiles]
        t0 = time.perf_counter()
        scene.render_tiles(cam, used, 0, 0, nthreads=threads)
        dt = time.perf_counter() - t0
        paths = int(((used["x_end"].astype(np.int64) - used["x_begin"]) * (used["y_end"].astype(np.int64) - used["y_begin"]) * (used["s_end"].astype(np.int64) - used["s_begin"])).sum())
        measured.append((paths / dt / 1e6, threads, n_tiles, paths))
    cam.film_samples = spp
    cam.film_width, cam.film_height = full
    scene.close()
    rate, threads, n_tiles, paths = max(measured, key=lambda m: m[0])
    what = f"the same camera on a {w}x{h} film (1/{slice_div[0] ** 2} of the area)" if slice_div else "the same film"
    return {"value": round(rate, 4), "unit": "Mspp/s", "cores": min(threads, n_tiles), "kind": "port",
            "sample": f"{what}: first {n_tiles} of {len(tiles)} tiles in the reference's centre-out order at {slice_spp} spp ({paths} paths); the reference's cost is linear in pixels and samples (one tile per host thread, per-tile mt19937 streams)",
            "cpu_model": cpu_model(), "online_cores": online_cores(),
            "other_thread_counts": [{"cores": t, "value": round(r, 4)} for r, t, _, _ in measured if t != threads]}


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
    return w * h * spp, wall, mean, (cpu_render_baseline(sb, cam, w, h, spp, slice_div=(8, 16) if w * h > (1 << 22) else None) if cpu else None)


L2_PEAK_GBS = 34500.0    # aggregate L2 bandwidth of the eight XCDs
CUS, LANES_PER_CU_CYCLE = 256, 64   # four 16-lane SIMDs per CU: 64 lane-operations per CU per cycle


def latest_profile(stem):
    """Newest profiles/rNN_<stem>.json (the per-round counter summaries written by tools/summarise_profiles.py)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{stem}.json")))
    return (json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)) if files else (None, None)


def roofline_block(kernel_ms, n, work, ref_nodes_per_ray, ref_tris_per_ray):
    """Roofs of the dominant kernel (k_mesh_persistent<culled, closest>).  Kernel time is this run's (HIP events around
    each launch on the launch stream); the counter figures are the committed per-launch sums of separate rocprofv3
    --pmc passes over the same kernel and workload, named in `sources`, so every fraction can be recomputed from
    profiles/ and this line.

    The kernel is not DRAM-bound: nodes + triangles (69 MB) stay in L2 / Infinity Cache.  Roofs, nearest first:
      valu_issue   cycles in which a SIMD issues a vector-ALU instruction / all SIMD cycles
      valu_lanes   lane-operations executed / (CUs x 64 lanes x cycles): issue rate x lanes active per instruction
      l2_requests  bytes the traversal requests (64 B per two-child node, 48 B per triangle, ray, hit, order index;
                   device-side visit counts of this run) / L2 aggregate peak
      l1_accesses  vector-L1 accesses (one per lane per 16-byte load of its own record) / (CUs x cycles): the L1
                   looks up about one line per clock
      hbm_fabric   L2<->fabric bytes (FETCH_SIZE + WRITE_SIZE) / HBM peak: an upper bound on DRAM traffic, since
                   Infinity Cache hits are counted
    `frac` of the block is hbm_fabric's (the memory roof the contract names); `nearest_roof` says which one binds."""
    sec = kernel_ms * 1e-3
    traffic, traffic_src = latest_profile("traffic")
    pmc, pmc_src = latest_profile("pmc_closest")
    roofs = {}
    fabric_bytes = traffic.get("closest_culled_hbm_bytes_per_launch") if traffic else None
    if fabric_bytes:
        gbs = fabric_bytes / sec / 1e9
        roofs["hbm_fabric"] = {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                               "bytes_per_launch": fabric_bytes, "l2_hit_rate": round(traffic.get("l2_hit_rate") or 0.0, 4)}
    req_gbs = work["bytes_per_ray"] * n / sec / 1e9
    roofs["l2_requests"] = {"achieved": round(req_gbs, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(req_gbs / L2_PEAK_GBS, 4),
                            "bytes_per_ray": work["bytes_per_ray"], "bytes_per_launch": round(work["bytes_per_ray"] * n)}
    if pmc:
        c = pmc["counters"]
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0                       # the counter sums the eight XCDs
        issue = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (cycles * CUS * 4)   # quad-cycles -> cycles, over 1024 SIMDs
        lanes = c["SQ_THREAD_CYCLES_VALU"] / (CUS * LANES_PER_CU_CYCLE * cycles)
        roofs["valu_issue"] = {"frac": round(issue, 4), "SQ_ACTIVE_INST_VALU": c["SQ_ACTIVE_INST_VALU"], "cycles_per_xcd": cycles}
        roofs["valu_lanes"] = {"frac": round(lanes, 4), "SQ_THREAD_CYCLES_VALU": c["SQ_THREAD_CYCLES_VALU"],
                               "lanes_active_per_instruction": round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"], 1),
                               "achieved": round(c["SQ_THREAD_CYCLES_VALU"] / sec / 1e12, 2), "peak": round(CUS * LANES_PER_CU_CYCLE * 2.4e9 / 1e12, 2), "unit": "T lane-ops/s at 2.4 GHz"}
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
            # every lane of a traversal step reads its own node or triangle record: four 16-byte accesses per node, three
            # per triangle, each a line lookup of its own in the CU's vector L1, which serves about one per clock
            acc = c["TCP_TOTAL_CACHE_ACCESSES_sum"]
            roofs["l1_accesses"] = {"frac": round(acc / (cycles * CUS), 4), "TCP_TOTAL_CACHE_ACCESSES_sum": acc, "accesses_per_ray": round(acc / n, 1),
                                    "peak": "one vector-L1 access per clock per CU"}
        if "TD_TD_BUSY_sum" in c:
            roofs["vector_memory_return_busy"] = {"busy_frac": round(c["TD_TD_BUSY_sum"] / (cycles * CUS), 4), "TD_TD_BUSY_sum": c["TD_TD_BUSY_sum"]}
    nearest = max((k for k in roofs if "frac" in roofs[k]), key=lambda k: roofs[k]["frac"])
    ref_bytes = 32 + 8 + 32 * ref_nodes_per_ray + 36 * ref_tris_per_ray
    ref_gbs = ref_bytes * n / sec / 1e9
    primary = roofs.get("hbm_fabric") or roofs["l2_requests"]
    return {"bound": "hbm", "achieved": primary["achieved"], "peak": primary["peak"], "unit": "GB/s", "frac": primary["frac"],
            "traffic": fabric_bytes, "kernel": "k_mesh_persistent<culled, closest>", "kernel_ms": round(kernel_ms, 4),
            "nearest_roof": nearest, "nearest_roof_frac": roofs[nearest]["frac"], "roofs": roofs,
            "sources": {"kernel_ms": "HIP events around each launch, this run", "traffic": traffic_src, "pmc": pmc_src,
                        "visit_counts": "device-side counters, 64k-ray prefix, this run"},
            "culled_traversal": work,
            "reference_semantics_equiv": {"bytes_per_ray": round(ref_bytes, 1), "nodes_per_ray": round(ref_nodes_per_ray, 2), "tris_per_ray": round(ref_tris_per_ray, 2),
                                          "GBps": round(ref_gbs, 1),
                                          "note": "bytes a traversal with the reference's visit semantics (no culling: every node whose ancestors pass the slab test, "
                                                  "32 B per box + 36 B per triangle, counts from the CPU checker) would move for the same batch, over this kernel's time: "
                                                  "an equivalent-work figure, not bytes this kernel moves -- the culled order skips about 3/4 of those visits"},
            "note": "achieved/peak/frac = L2<->fabric bytes per launch over live kernel time against HBM peak (Infinity Cache hits included, so an upper bound on DRAM "
                    "traffic); nearest_roof names the roof that binds: the vector L1's access rate (one record piece per lane per clock) and vector-ALU issue are both near saturation, with under half the lanes active per instruction; DRAM is not"}


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
    ap.add_argument("--no-4k", action="store_true", help="skip the 3840x2160, 1024 spp render leg (about 30 s on one GPU)")
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
    # ---- workload: same scene on every rank, a different ray batch per rank
    verts, idx = synth.soup(args.tris)
    t0 = time.perf_counter()
    bvh = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4)
    build_s = time.perf_counter() - t0
    nodes, order = bvh.nodes(), bvh.order()
    # the same tree from the GPU builder (what scene creation uses for meshes of this size): time, and identity
    capi.Bvh(verts=verts[:64], indices=idx[:8] % 64, max_leaf_size=4, gpu_device=local_rank).close()   # code load
    t0 = time.perf_counter()
    gbvh = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4, gpu_device=local_rank)
    gpu_build_s = time.perf_counter() - t0
    gpu_build_identical = bool(gbvh.built_on_device and gbvh.nodes().tobytes() == nodes.tobytes() and np.array_equal(gbvh.order(), order))
    gbvh.close()
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
        "rccl_ranks": dist.get_world_size() if dist is not None else 1, "collective_backend": (dist.get_backend() if dist is not None else "none"),
        "rank_devices": rank_devices,
        "config": {"workload": f"config 3: {n} incoherent rays per GPU vs {args.tris}-triangle soup BVH (max_leaf 4), closest-hit (t, primitive id)",
                   "traversal": "culled" if headline_mode == abi.PTX_TRAVERSE_CULLED else "reference", "rays_per_gpu": n,
                   "triangles": args.tris, "bvh_nodes": int(len(nodes)), "bvh_build_s": round(build_s, 3), "bvh_build_gpu_s": round(gpu_build_s, 4), "bvh_gpu_build_identical_to_host": gpu_build_identical},
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
            want_cpu = rank == 0 and world == 1 and not args.no_cpu
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
            result["roofline"] = roofline_block(kernel_ms, n, work, extra["nodes_per_ray"], extra["tris_per_ray"])
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    scene.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

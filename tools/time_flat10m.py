"""Timing of the ray queries on a >= 10 M-triangle mesh (a tree that lives in HBM):
    python tools/time_flat10m.py [cows|soup] [n_rays] [reps]
cows = the 1,728 cows of the instanced grid baked into one 10,119,168-triangle mesh (tests/scenes.py),
soup = the benchmark's triangle soup generator at 10,000,000 triangles.
Prints kernel and call times of the culled closest-hit, reference-order and occlusion queries, the per-ray visit
counts of the culled walk, and the mismatch count between the two orders."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from pathtracer_amd import abi, capi, synth

which = sys.argv[1] if len(sys.argv) > 1 else "cows"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 24
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
t0 = time.perf_counter()
if which == "cows":
    import scenes
    verts, idx = scenes.flattened_cows(12)
    extent = 14.4
else:
    verts, idx = synth.soup(10_000_000)
    extent = 10.0
t1 = time.perf_counter()
scene = capi.Scene.from_mesh(verts, idx, device=0)
t2 = time.perf_counter()
print(f"{which}: {len(idx)} triangles, generated in {t1 - t0:.2f} s, scene resident in {t2 - t1:.2f} s, {len(scene.mesh_nodes())} nodes", flush=True)
dev = torch.device("cuda", 0)
rays_h = synth.rays(n, key=synth.RAY_KEY + 10, extent=extent)
shadow_h = synth.shadow_rays(n, key=synth.SHADOW_KEY + 10, extent=extent)
rays = torch.from_numpy(rays_h.view(np.float32).reshape(-1, 8)).to(dev)
shadow = torch.from_numpy(shadow_h.view(np.float32).reshape(-1, 8)).to(dev)
ref = torch.empty((n, 2), dtype=torch.int32, device=dev)
hits = torch.empty((n, 2), dtype=torch.int32, device=dev)
occ_r = torch.empty(n, dtype=torch.uint8, device=dev)
occ = torch.empty(n, dtype=torch.uint8, device=dev)
scene.trace_closest_device(rays, ref, flags=abi.PTX_TRAVERSE_REFERENCE, n=n)
scene.trace_closest_device(rays, hits, flags=abi.PTX_TRAVERSE_CULLED, n=n)
scene.trace_any_device(shadow, occ_r, flags=abi.PTX_TRAVERSE_REFERENCE, n=n)
scene.trace_any_device(shadow, occ, flags=abi.PTX_TRAVERSE_CULLED, n=n)
torch.cuda.synchronize()
print(f"mismatches culled vs reference order: closest {int((ref != hits).any(dim=1).sum())}, occlusion {int((occ != occ_r).sum())}; "
      f"hit fraction {float((ref[:, 1] != -1).float().mean()):.3f}, occluded {float(occ.float().mean()):.3f}", flush=True)


def timed(fn, label):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    kt = scene.kernel_times_ms(reps)
    print(f"{label}: kernel {kt.mean():.3f} ms (min {kt.min():.3f}), call {wall:.3f} ms, {n / wall / 1e3:.1f} Mrays/s", flush=True)


timed(lambda: scene.trace_closest_device(rays, hits, flags=abi.PTX_TRAVERSE_CULLED | abi.PTX_ASYNC, n=n), "closest culled")
timed(lambda: scene.trace_any_device(shadow, occ, flags=abi.PTX_TRAVERSE_CULLED | abi.PTX_ASYNC, n=n), "occlusion culled")
timed(lambda: scene.trace_closest_device(rays, ref, flags=abi.PTX_TRAVERSE_REFERENCE | abi.PTX_ASYNC, n=n), "closest reference order")
st = scene.trace_closest_stats(rays_h[:65536], flags=abi.PTX_TRAVERSE_CULLED).astype(np.float64).mean(axis=0)
print(f"culled walk per ray: {st[0]:.1f} interior nodes, {st[1]:.1f} leaves, {st[2]:.1f} triangles, deepest stack {st[3]:.1f} -> {44 + 64 * st[0] + 48 * st[2]:.0f} B requested")

"""Profiling driver: runs the ray-query kernels a few times.
    python3 tools/prof_closest.py [n_rays] [reps] [modes: r c a] [mesh: soup1m | cows10m | soup10m]
soup1m  = the benchmark's 1,000,000-triangle soup (config 3), cows10m = the 1,728 cows baked into one
10,119,168-triangle mesh, soup10m = the soup generator at 10,000,000 triangles."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from pathtracer_amd import abi, capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
modes = sys.argv[3] if len(sys.argv) > 3 else "rca"
mesh = sys.argv[4] if len(sys.argv) > 4 else "soup1m"
if mesh == "cows10m":
    import scenes
    verts, idx = scenes.flattened_cows(12)
    rays_h, shadow_h = synth.rays(n, key=synth.RAY_KEY + 10, extent=14.4), synth.shadow_rays(n, key=synth.SHADOW_KEY + 10, extent=14.4)
elif mesh == "soup10m":
    verts, idx = synth.soup(10_000_000)
    rays_h, shadow_h = synth.rays(n, key=synth.RAY_KEY + 10), synth.shadow_rays(n, key=synth.SHADOW_KEY + 10)
else:
    verts, idx = synth.soup(1_000_000)
    rays_h, shadow_h = synth.rays(n), synth.shadow_rays(n)
scene = capi.Scene.from_mesh(verts, idx, device=0)
dev = torch.device("cuda", 0)
rays = torch.from_numpy(rays_h.view(np.float32).reshape(-1, 8)).to(dev)
shadow = torch.from_numpy(shadow_h.view(np.float32).reshape(-1, 8)).to(dev)
hits = torch.empty((n, 2), dtype=torch.int32, device=dev)
occ = torch.empty(n, dtype=torch.uint8, device=dev)
for _ in range(reps):
    if "r" in modes: scene.trace_closest_device(rays, hits, flags=abi.PTX_TRAVERSE_REFERENCE, n=n)
    if "c" in modes: scene.trace_closest_device(rays, hits, flags=abi.PTX_TRAVERSE_CULLED, n=n)
    if "a" in modes: scene.trace_any_device(shadow, occ, flags=abi.PTX_TRAVERSE_CULLED, n=n)
torch.cuda.synchronize()
print("done", mesh, n)

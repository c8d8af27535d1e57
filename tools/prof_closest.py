"""Profiling driver: runs the ray-query kernels a few times on the config-3 workload."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pathtracer_amd import abi, capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
modes = sys.argv[3] if len(sys.argv) > 3 else "rca"
verts, idx = synth.soup(1_000_000)
scene = capi.Scene.from_mesh(verts, idx, device=0)
dev = torch.device("cuda", 0)
rays = torch.from_numpy(synth.rays(n).view(np.float32).reshape(-1, 8)).to(dev)
shadow = torch.from_numpy(synth.shadow_rays(n).view(np.float32).reshape(-1, 8)).to(dev)
hits = torch.empty((n, 2), dtype=torch.int32, device=dev)
occ = torch.empty(n, dtype=torch.uint8, device=dev)
for _ in range(reps):
    if "r" in modes: scene.trace_closest_device(rays, hits, flags=abi.PTX_TRAVERSE_REFERENCE, n=n)
    if "c" in modes: scene.trace_closest_device(rays, hits, flags=abi.PTX_TRAVERSE_CULLED, n=n)
    if "a" in modes: scene.trace_any_device(shadow, occ, flags=abi.PTX_TRAVERSE_CULLED, n=n)
torch.cuda.synchronize()
print("done", n)

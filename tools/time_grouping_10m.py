"""What the grouping pass is worth where the tree lives in HBM: culled closest-hit on the 10 M-triangle meshes with the
library's grouping pass and without it (PTX_RAYS_COHERENT: the batch is traced in the order given, here random).
    python tools/time_grouping_10m.py [cows|soup] [n_rays]"""
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
if which == "cows":
    import scenes
    verts, idx = scenes.flattened_cows(12)
    extent = 14.4
else:
    verts, idx = synth.soup(10_000_000)
    extent = 10.0
scene = capi.Scene.from_mesh(verts, idx, device=0)
dev = torch.device("cuda", 0)
rays = torch.from_numpy(synth.rays(n, key=synth.RAY_KEY + 10, extent=extent).view(np.float32).reshape(-1, 8)).to(dev)
hits = torch.empty((n, 2), dtype=torch.int32, device=dev)
for label, flags in (("grouped", abi.PTX_TRAVERSE_CULLED), ("as given (random order)", abi.PTX_TRAVERSE_CULLED | abi.PTX_RAYS_COHERENT)):
    for _ in range(2):
        scene.trace_closest_device(rays, hits, flags=flags, n=n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        scene.trace_closest_device(rays, hits, flags=flags | abi.PTX_ASYNC, n=n)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    kt = scene.kernel_times_ms(5)
    print(f"{which} {label}: kernel {kt.mean():.3f} ms, call {wall:.3f} ms, {n / wall / 1e3:.1f} Mrays/s", flush=True)

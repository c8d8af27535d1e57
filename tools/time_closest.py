"""A/B timing of the config-3 closest-hit kernel:  PTX_AMD_LIB=<variant .so> python tools/time_closest.py [n_rays] [reps]
Prints the mean device time of the traversal kernel (HIP events around each launch), the whole-call time including
the grouping pass, and the number of rays whose culled result differs from the reference-order result."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pathtracer_amd import abi, capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
verts, idx = synth.soup(1_000_000)
scene = capi.Scene.from_mesh(verts, idx, device=0)
dev = torch.device("cuda", 0)
rays = torch.from_numpy(synth.rays(n).view(np.float32).reshape(-1, 8)).to(dev)
ref = torch.empty((n, 2), dtype=torch.int32, device=dev)
hits = torch.empty((n, 2), dtype=torch.int32, device=dev)
scene.trace_closest_device(rays, ref, flags=abi.PTX_TRAVERSE_REFERENCE, n=n)
scene.trace_closest_device(rays, hits, flags=abi.PTX_TRAVERSE_CULLED, n=n)
torch.cuda.synchronize()
bad = int((ref != hits).any(dim=1).sum().item())
for _ in range(2):
    scene.trace_closest_device(rays, hits, flags=abi.PTX_TRAVERSE_CULLED, n=n)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    scene.trace_closest_device(rays, hits, flags=abi.PTX_TRAVERSE_CULLED | abi.PTX_ASYNC, n=n)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps * 1e3
kt = scene.kernel_times_ms(reps)
print(f"{os.path.basename(capi.LIB_PATH)}: kernel {kt.mean():.3f} ms (min {kt.min():.3f}), call {wall:.3f} ms, {n / wall / 1e3:.1f} Mrays/s, mismatches vs reference order {bad}")

"""Device-resident closest-hit rate on the config-3 scene at several batch sizes: python3 tools/time_closest_sizes.py [sizes...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pathtracer_amd import abi, capi, synth

sizes = [int(a) for a in sys.argv[1:]] or [1 << 20, 1 << 22, 1 << 24]
verts, idx = synth.soup(1_000_000)
scene = capi.Scene.from_mesh(verts, idx, device=0)
dev = torch.device("cuda", 0)
all_rays = synth.rays(max(sizes))
for n in sizes:
    rays = torch.from_numpy(all_rays[:n].view(np.float32).reshape(-1, 8)).to(dev)
    ref = torch.empty((n, 2), dtype=torch.int32, device=dev)
    hits = torch.empty((n, 2), dtype=torch.int32, device=dev)
    scene.trace_closest_device(rays, ref, flags=abi.PTX_TRAVERSE_REFERENCE, n=n)
    for _ in range(3):
        scene.trace_closest_device(rays, hits, flags=abi.PTX_TRAVERSE_CULLED, n=n)
    torch.cuda.synchronize()
    bad = int((ref != hits).any(dim=1).sum().item())
    reps = max(5, (1 << 26) // n)
    t0 = time.perf_counter()
    for _ in range(reps):
        scene.trace_closest_device(rays, hits, flags=abi.PTX_TRAVERSE_CULLED | abi.PTX_ASYNC, n=n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    k = scene.kernel_times_ms(reps) if hasattr(scene, "kernel_times_ms") else []
    kern = f", traversal kernel {np.mean(k):.3f} ms" if len(k) else ""
    print(f"{n} rays: {dt * 1e3:.3f} ms per call, {n / dt / 1e6:.1f} Mrays/s{kern}, mismatches vs reference order {bad}", flush=True)

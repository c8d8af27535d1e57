"""Closest-hit rate in the reference order (the API's default) on the config-3 scene: python3 tools/time_reference_order.py [n_rays]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pathtracer_amd import abi, capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
verts, idx = synth.soup(1_000_000)
scene = capi.Scene.from_mesh(verts, idx, device=0)
dev = torch.device("cuda", 0)
rays = torch.from_numpy(synth.rays(n).view(np.float32).reshape(-1, 8)).to(dev)
hits = torch.empty((n, 2), dtype=torch.int32, device=dev)
scene.trace_closest_device(rays, hits, flags=abi.PTX_TRAVERSE_REFERENCE, n=n)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    scene.trace_closest_device(rays, hits, flags=abi.PTX_TRAVERSE_REFERENCE | abi.PTX_ASYNC, n=n)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"{n} rays, reference order: {dt * 1e3:.2f} ms per call, {n / dt / 1e6:.1f} Mrays/s, kernel {np.mean(scene.kernel_times_ms(3)):.2f} ms, checksum {int(hits.to(torch.int64).sum().item())}", flush=True)

"""Timing of the closest-hit kernel under the env knobs (one process per setting)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pathtracer_amd import abi, capi, synth
n = 1 << 24
verts, idx = synth.soup(1_000_000)
scene = capi.Scene.from_mesh(verts, idx, device=0)
dev = torch.device("cuda", 0)
rays = torch.from_numpy(synth.rays(n).view(np.float32).reshape(-1, 8)).to(dev)
hits = torch.empty((n, 2), dtype=torch.int32, device=dev)
flags = abi.PTX_TRAVERSE_CULLED | abi.PTX_ASYNC
for _ in range(2): scene.trace_closest_device(rays, hits, flags=flags, n=n)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): scene.trace_closest_device(rays, hits, flags=flags, n=n)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"tri_min={os.environ.get('PTX_TRI_MIN_LANES','1')} blocks/CU={os.environ.get('PTX_BLOCKS_PER_CU','8')} refill={os.environ.get('PTX_REFILL_MIN_IDLE','20')}: {ms:.2f} ms  {n/ms/1e3:.1f} Mrays/s", flush=True)
flags |= abi.PTX_RAYS_COHERENT
for _ in range(2): scene.trace_closest_device(rays, hits, flags=flags, n=n)
torch.cuda.synchronize()
e0.record()
for _ in range(5): scene.trace_closest_device(rays, hits, flags=flags, n=n)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"   without the grouping pass: {ms:.2f} ms  {n/ms/1e3:.1f} Mrays/s", flush=True)

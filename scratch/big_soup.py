"""10 M-triangle soup: build, upload, trace 4 M rays in both orders, compare."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pathtracer_amd import abi, capi, synth
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
t0 = time.perf_counter()
verts, idx = synth.soup(nt)
print(f"soup {nt}: {time.perf_counter()-t0:.1f} s", flush=True)
t0 = time.perf_counter()
scene = capi.Scene.from_mesh(verts, idx, device=0)
print(f"scene (build + upload): {time.perf_counter()-t0:.1f} s", flush=True)
n = 1 << 22
dev = torch.device("cuda", 0)
rays = torch.from_numpy(synth.rays(n).view(np.float32).reshape(-1, 8)).to(dev)
a = torch.empty((n, 2), dtype=torch.int32, device=dev); b = torch.empty_like(a)
for flags, out, name in ((abi.PTX_TRAVERSE_REFERENCE, a, "reference"), (abi.PTX_TRAVERSE_CULLED, b, "culled")):
    scene.trace_closest_device(rays, out, flags=flags, n=n); torch.cuda.synchronize()
    t0 = time.perf_counter()
    scene.trace_closest_device(rays, out, flags=flags, n=n); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {n/dt/1e6:.1f} Mrays/s", flush=True)
print("mismatches", int((a != b).any(dim=1).sum().item()), "hit fraction", float((a[:, 1] != -1).float().mean().item()))

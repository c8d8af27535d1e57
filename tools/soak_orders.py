"""Soak of the culled order against the reference order on bench-shaped workloads with fresh seeds (no fixture: the two
orders against each other on the GPU; closest hit compares distance bits and primitive, occlusion the flag):
python3 tools/soak_orders.py [rounds] [log2_rays] [triangle counts, comma-separated] > profiles/rNN_order_soak.json
Odd rounds use larger triangles (more overlap along a ray, more equal-distance candidates at shared positions)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pathtracer_amd import abi, capi, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
n = 1 << log2n
sizes = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (1_000_000, 300_000, 2_000_000)
dev = torch.device("cuda", 0)
out = {"what": "culled vs reference order on the GPU: fresh random soups and rays per round; closest hit (distance bits, primitive) and occlusion flags",
       "rays_per_round": n, "rounds": []}
t_start = time.time()
for k in range(rounds):
    tris = sizes[k % len(sizes)]
    half = 0.04 if k % 2 == 0 else 0.25
    verts, idx = synth.soup(tris, key=0xA000 + k, half=half)
    scene = capi.Scene.from_mesh(verts, idx, device=0)
    rays = torch.from_numpy(synth.rays(n, key=0xB000 + k).view(np.float32).reshape(-1, 8)).to(dev)
    shadow = torch.from_numpy(synth.shadow_rays(n, key=0xC000 + k).view(np.float32).reshape(-1, 8)).to(dev)
    a = torch.empty((n, 2), dtype=torch.int32, device=dev)
    b = torch.empty((n, 2), dtype=torch.int32, device=dev)
    oa = torch.empty(n, dtype=torch.uint8, device=dev)
    ob = torch.empty(n, dtype=torch.uint8, device=dev)
    scene.trace_closest_device(rays, a, flags=abi.PTX_TRAVERSE_REFERENCE, n=n)
    scene.trace_closest_device(rays, b, flags=abi.PTX_TRAVERSE_CULLED, n=n)
    scene.trace_any_device(shadow, oa, flags=abi.PTX_TRAVERSE_REFERENCE, n=n)
    scene.trace_any_device(shadow, ob, flags=abi.PTX_TRAVERSE_CULLED, n=n)
    torch.cuda.synchronize()
    rec = {"triangles": tris, "half_extent": half, "hit_fraction": round(float((a[:, 1] != -1).float().mean().item()), 4),
           "closest_mismatches": int((a != b).any(dim=1).sum().item()), "occluded_fraction": round(float(oa.float().mean().item()), 4),
           "occlusion_mismatches": int((oa != ob).sum().item())}
    out["rounds"].append(rec)
    print(f"[soak] round {k}: {rec} ({time.time() - t_start:.0f} s)", file=sys.stderr, flush=True)
    scene.close()
    del rays, shadow, a, b, oa, ob
out["total_rays"] = 2 * n * rounds
out["total_closest_mismatches"] = sum(r["closest_mismatches"] for r in out["rounds"])
out["total_occlusion_mismatches"] = sum(r["occlusion_mismatches"] for r in out["rounds"])
print(json.dumps(out, indent=1))

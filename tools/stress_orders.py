"""Stress of the culled traversal order against the reference order (no fixture: the two orders against each other on
the GPU) where the culled order's guard band matters: rays that skim thin triangles at shallow and very shallow
angles.  python3 tools/stress_orders.py [log2_rays_per_case] > profiles/rNN_order_stress.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import scenes
from pathtracer_amd import abi, capi, synth


from scenes import skimming_rays as grazing, slivers

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
from test_oracle_golden import sphere5_mesh
cases = {"cow": scenes.cow_mesh(), "sphere_level5": sphere5_mesh()[:2], "soup_1M": synth.soup(1_000_000), "slivers_200k": slivers(200_000, 5)}
out = {"what": "culled order vs reference order on the GPU, distance bits and primitive, rays skimming random triangles; lift = distance of the ray origin from the triangle's plane over the mesh's diagonal (smaller = shallower)",
       "rays_per_case": 1 << log2n, "cases": []}
for name, (verts, idx) in cases.items():
    sc = capi.Scene.from_mesh(verts, idx)
    for lift in (2e-3, 1e-4, 3e-6, 1e-7):
        rays = grazing(verts, idx, 1 << log2n, 1234, lift)
        a = sc.trace_closest(rays, flags=abi.PTX_TRAVERSE_REFERENCE)["hits"]
        b = sc.trace_closest(rays, flags=abi.PTX_TRAVERSE_CULLED)["hits"]
        bad = int(((a["distance"].view(np.uint32) != b["distance"].view(np.uint32)) | (a["prim"] != b["prim"])).sum())
        out["cases"].append({"mesh": name, "triangles": int(len(idx)), "lift": lift, "hit_fraction": round(float((a["prim"] != abi.PTX_NO_HIT).mean()), 4), "mismatches": bad})
        print(name, lift, bad, file=sys.stderr, flush=True)
    sc.close()
out["total_mismatches"] = sum(c["mismatches"] for c in out["cases"])
print(json.dumps(out, indent=1))

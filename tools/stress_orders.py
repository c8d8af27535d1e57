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


def grazing(verts, idx, n, seed, lift):
    """From just off a random triangle's plane (`lift` x the mesh's size) towards a point of that triangle, along the plane."""
    g = np.random.Generator(np.random.Philox(key=seed))
    tri = idx[g.integers(0, len(idx), n)]
    a, b, c = (verts[tri[:, k], :3].astype(np.float64) for k in range(3))
    nrm = np.cross(b - a, c - a)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    w = g.random((n, 3)) + 0.05
    w /= w.sum(axis=1, keepdims=True)
    target = a * w[:, :1] + b * w[:, 1:2] + c * w[:, 2:]
    along = (b - a) * (g.random((n, 1)) * 2 - 1) + (c - a) * (g.random((n, 1)) * 2 - 1)
    along /= np.maximum(np.linalg.norm(along, axis=1, keepdims=True), 1e-30)
    size = np.linalg.norm(verts[:, :3].max(axis=0) - verts[:, :3].min(axis=0))
    o = target - along * (0.05 + g.random((n, 1))) * size * 0.3 + nrm * (g.random((n, 1)) * 2 - 1) * size * lift
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros(n, dtype=abi.RAY_DTYPE)
    r["o"] = o.astype(np.float32)
    r["d"] = synth._unit_f32(d)
    r["tmin"], r["tmax"] = 0.0, np.inf
    return r


def slivers(n, key):
    """Long thin triangles (aspect 1:200) in random orientations inside a unit cube: many overlap along any line."""
    g = np.random.Generator(np.random.Philox(key=key))
    c = g.random((n, 3))
    u = g.standard_normal((n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = np.cross(u, g.standard_normal((n, 3)))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    p = np.stack([c - u * 0.1, c + u * 0.1, c + v * 0.001], axis=1).reshape(-1, 3)
    verts = np.zeros((3 * n, 8), dtype=np.float32)
    verts[:, :3] = p
    verts[:, 4] = 1.0
    return verts, np.arange(3 * n, dtype=np.uint32).reshape(n, 3)


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

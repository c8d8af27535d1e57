"""Soak against the reference's own code: random soups and rays, the GPU's answers in both traversal orders against
BVH<Triangle>::hit of oracle/_ref/libptref.so run on the host's cores (the builder's tree injected, primitives
id-coded): distance bits and primitive.  python3 tools/soak_reference.py [rounds] [log2_rays] > profiles/rNN_reference_soak.json
CPU-bound: about 1 Mray/s on 16 cores."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _oracle import Ref
from pathtracer_amd import abi, capi, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 22)
threads = min(16, len(os.sched_getaffinity(0)))
assert Ref.available(), "oracle/_ref/libptref.so is not built (make -C oracle ref)"
ref = Ref()
out = {"what": "GPU (reference order and culled order) against the reference's BVH<Triangle>::hit on the host: distance bits and primitive", "rays_per_round": n, "rounds": []}
t0 = time.time()
for k in range(rounds):
    tris = (1_000_000, 200_000, 50_000)[k % 3]
    half = (0.04, 0.25, 0.1, 0.5)[k % 4]
    verts, idx = synth.soup(tris, key=0xD000 + k, half=half)
    bvh = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4)
    nodes, order = bvh.nodes(), bvh.order()
    scene = capi.Scene.from_mesh(verts, idx[order], nodes=nodes)
    rmesh = ref.mesh(verts, idx[order], id_coded=True, nodes=nodes)
    rays = synth.rays(n, key=0xE000 + k)
    want = rmesh.hit(rays, nthreads=threads)["hits"]
    rec = {"triangles": tris, "half_extent": half, "hit_fraction": round(float((want["prim"] != abi.PTX_NO_HIT).mean()), 4)}
    for name, mode in (("reference_order", abi.PTX_TRAVERSE_REFERENCE), ("culled_order", abi.PTX_TRAVERSE_CULLED)):
        got = scene.trace_closest(rays, flags=mode, want_bary=False, want_inst=False)["hits"]
        rec[name + "_mismatches"] = int(((got["distance"].view(np.uint32) != want["distance"].view(np.uint32)) | (got["prim"] != want["prim"])).sum())
    out["rounds"].append(rec)
    print(f"[soak] round {k}: {rec} ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
    scene.close(), rmesh.close(), bvh.close()
out["total_rays"] = n * rounds
out["reference_order_mismatches"] = sum(r["reference_order_mismatches"] for r in out["rounds"])
out["culled_order_mismatches"] = sum(r["culled_order_mismatches"] for r in out["rounds"])
out["host_threads"] = threads
print(json.dumps(out, indent=1))

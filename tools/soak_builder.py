"""Soak of the GPU builder against the host builder (whose trees equal the reference builder's byte for byte on the
committed fixtures): random box sets of many shapes -- uniform, clustered, on a lattice (centres on bin boundaries),
with duplicates, flat, elongated, of sizes around powers of two -- node array, primitive order and statistics must be
identical.  python3 tools/soak_builder.py [rounds] > profiles/rNN_builder_soak.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pathtracer_amd import capi


def boxes_for(k):
    g = np.random.Generator(np.random.Philox(key=0x7000 + k))
    n = int(g.choice([1 << int(g.integers(4, 21)), (1 << int(g.integers(8, 20))) + int(g.integers(-3, 4)), int(g.integers(1, 400_000))]))
    n = max(1, n)
    kind = k % 6
    if kind == 0:      # uniform
        c = g.random((n, 3), dtype=np.float32) * np.float32(10)
    elif kind == 1:    # clustered
        centres = g.random((int(g.integers(1, 20)), 3)) * 10
        c = (centres[g.integers(0, len(centres), n)] + g.standard_normal((n, 3)) * 0.05).astype(np.float32)
    elif kind == 2:    # lattice: many equal centres per axis, centres on bin boundaries
        c = (g.integers(0, int(g.integers(2, 33)), (n, 3)).astype(np.float32) * np.float32(0.125))
    elif kind == 3:    # duplicates
        base = g.random((max(1, n // int(g.integers(2, 50))), 3), dtype=np.float32) * np.float32(4)
        c = base[g.integers(0, len(base), n)]
    elif kind == 4:    # flat: one axis without extent
        c = g.random((n, 3), dtype=np.float32) * np.float32(10)
        c[:, int(g.integers(0, 3))] = np.float32(g.random())
    else:              # elongated, far from the origin
        c = (g.random((n, 3)) * np.array([1000.0, 1.0, 0.01]) + np.array([1e4, -5e3, 7.0])).astype(np.float32)
    h = (g.random((n, 3), dtype=np.float32) * np.float32(g.random() * 0.2) + np.float32(1e-4)).astype(np.float32)
    if kind == 3:
        h[:] = h[0]
    b = np.concatenate([c - h, c + h], axis=1).astype(np.float32)
    b[b == 0] = 0.0    # no negative zeros (those inputs are built on the host by design)
    return b, int(g.choice([1, 2, 4, 8, 15]))


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
tot = {"rounds": 0, "boxes": 0, "identical": 0, "built_on_device": 0}
bad = []
t0 = time.time()
for k in range(rounds):
    b, leaf = boxes_for(k)
    host = capi.Bvh(boxes=b, max_leaf_size=leaf)
    gpu = capi.Bvh(boxes=b, max_leaf_size=leaf, gpu_device=0)
    same = host.nodes().tobytes() == gpu.nodes().tobytes() and np.array_equal(host.order(), gpu.order()) and np.array_equal(host.stats(), gpu.stats()) and host.max_depth == gpu.max_depth
    tot["rounds"] += 1
    tot["boxes"] += len(b)
    tot["identical"] += int(same)
    tot["built_on_device"] += int(gpu.built_on_device)
    if not same:
        bad.append({"round": k, "boxes": len(b), "max_leaf": leaf, "kind": k % 6})
    host.close(), gpu.close()
    if k % 100 == 99:
        print(f"[soak] {tot} ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
print(json.dumps({"what": "GPU builder against the host builder on random box sets (uniform, clustered, lattice, duplicates, flat, elongated; 1 .. 1M boxes; max leaf 1, 2, 4, 8, 15): node array, order, statistics, depth",
                  **tot, "rounds_with_differences": bad}, indent=1))

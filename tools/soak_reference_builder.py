"""Soak of the host builder against the reference's own BVH::build (oracle/_ref/libptref.so; needs no GPU): random
triangle sets of many shapes and sizes, leaf sizes 1..8 -- node arrays byte for byte, and the reordered primitives'
boxes.  python3 tools/soak_reference_builder.py [seconds] [first_input] > out.json   (profiles/rNN_reference_builder_soak.json sums several runs)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _oracle import Ref
from pathtracer_amd import capi


def triangles_for(k):
    g = np.random.Generator(np.random.Philox(key=0x9000 + k))
    n = int(g.choice([int(g.integers(1, 64)), int(g.integers(64, 1000)), int(g.integers(1000, 6000))]))
    kind = k % 5
    if kind == 0:
        c = g.random((n, 1, 3)) * 10
    elif kind == 1:
        centres = g.random((int(g.integers(1, 12)), 3)) * 10
        c = (centres[g.integers(0, len(centres), n)] + g.standard_normal((n, 3)) * 0.05)[:, None, :]
    elif kind == 2:
        c = (g.integers(0, int(g.integers(2, 20)), (n, 1, 3)) * 0.125).astype(np.float64)
    elif kind == 3:
        base = g.random((max(1, n // int(g.integers(2, 30))), 1, 3)) * 4
        c = base[g.integers(0, len(base), n)]
    else:
        c = g.random((n, 1, 3)) * np.array([100.0, 1.0, 0.0]) + np.array([50.0, -3.0, 2.0])   # flat in z: Triangle::bbox pads it
    half = float(g.random() * 0.3 + 0.001)
    off = g.random((n, 3, 3)) * 2 * half - half
    if kind == 3:
        off[:] = off[0]        # exact duplicates
    if kind == 4:
        off[:, :, 2] = 0.0
    verts = np.zeros((3 * n, 8), dtype=np.float32)
    verts[:, :3] = (c + off).reshape(-1, 3).astype(np.float32)
    verts[:, 4] = 1.0
    return verts, np.arange(3 * n, dtype=np.uint32).reshape(n, 3), int(g.integers(1, 9))


def compare(k):
    """(triangles, reference defined, nodes identical, order identical) for input k; run in a child process.
    The reference's builder indexes its bins with floor((centre - min) / binLength) (bvh.cpp:111-112).  A node's range
    is cut by a comparison (bvh.cpp:191-199) that can disagree with that index by rounding, so a primitive can sit in a
    child whose box does not hold its centre, and one level down its index falls outside the eight bins: an
    out-of-bounds write in the reference (it crashes on some such inputs and not on others).  The host builder clamps
    the index and counts it; inputs with a clamped index have no defined reference answer and are only counted."""
    verts, idx, leaf = triangles_for(k)
    b = capi.Bvh(verts=verts, indices=idx, max_leaf_size=leaf)
    if int(b.stats()[0]) != 0:
        return len(idx), False, False, False
    rm = Ref().mesh(verts, idx, max_leaf=leaf, id_coded=False)
    rn, hn = rm.nodes(), b.nodes()
    same_nodes = len(rn) == len(hn) and rn.tobytes() == hn.tobytes()
    # the reference's reordered primitives' boxes against the boxes of the host builder's order
    tri = verts[idx[b.order()], :3]
    lo, hi = tri.min(axis=1), tri.max(axis=1)
    flat = (hi - lo) == 0     # Triangle::bbox pads an axis without extent by 1e-4 either way (tri_mesh.cpp:28-39)
    lo, hi = np.where(flat, lo - np.float32(1e-4), lo), np.where(flat, hi + np.float32(1e-4), hi)
    same_order = bool(np.array_equal(rm.prim_boxes(), np.concatenate([lo, hi], axis=1).astype(np.float32)))
    return len(idx), True, bool(same_nodes), same_order


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    assert Ref.available()
    tot = {"meshes": 0, "triangles": 0, "reference_undefined_out_of_range_bin_index": 0, "compared": 0, "identical_node_arrays": 0, "identical_primitive_order": 0,
           "reference_builder_crashed": 0}
    bad, crashed = [], []
    t0 = time.time()
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 0     # first input: several runs side by side take disjoint ranges
    devnull = os.open(os.devnull, os.O_WRONLY)
    while time.time() - t0 < budget:
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.close(r)
            os.dup2(devnull, 1)     # the reference's builder reports unseparable boxes on stdout
            res = compare(k)
            os.write(w, json.dumps(res).encode())
            os._exit(0)
        os.close(w)
        data = b""
        while True:
            chunk = os.read(r, 4096)
            if not chunk:
                break
            data += chunk
        os.close(r)
        os.waitpid(pid, 0)
        tot["meshes"] += 1
        if not data:
            tot["reference_builder_crashed"] += 1
            crashed.append({"round": k, "kind": k % 5})
        else:
            n, defined, same_nodes, same_order = json.loads(data)
            tot["triangles"] += n
            if not defined:
                tot["reference_undefined_out_of_range_bin_index"] += 1
                k += 1
                continue
            tot["compared"] += 1
            tot["identical_node_arrays"] += int(same_nodes)
            tot["identical_primitive_order"] += int(same_order)
            if not (same_nodes and same_order):
                bad.append({"round": k, "triangles": n, "kind": k % 5, "nodes": same_nodes, "order": same_order})
        k += 1
        if k % 200 == 0:
            print(f"[soak] {tot} ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
    print(json.dumps({"what": "host builder against the reference's own BVH::build on random triangle sets (uniform, clustered, lattice, exact duplicates, flat; 1 .. 6k triangles; max leaf 1..8): node arrays byte for byte, and the reordered primitives' boxes",
                      **tot, "meshes_with_differences": bad[:20], "inputs_the_reference_builder_crashed_on": crashed[:20]}, indent=1))


if __name__ == "__main__":
    main()

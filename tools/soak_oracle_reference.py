"""Soak of the CPU checker (oracle/) against the reference's own Aggregate / Instance / Tri_Mesh / Sphere
(oracle/_ref/libptref.so; needs no GPU): the random two-level scenes and rays of tests/test_gpu_fuzz.py with fresh
seeds -- hit flag, distance bits, hit position bits, material; normals and uv to 1e-6 (analytic spheres pass through
atan2/acos).  Scenes in which a mesh's or the instance level's build leaves the reference's eight bins (see tools/soak_reference_builder.py)
are set aside.  python3 tools/soak_oracle_reference.py [seconds] [first_seed] > out.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _oracle import Oracle, Ref
from pathtracer_amd import abi
from test_gpu_fuzz import random_scene, random_rays


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def compare(seed, n_rays):
    # The reference's Tri_Mesh always builds with leaf size 4 (tri_mesh.cpp:173); the generator draws other sizes.  Trees
    # of different leaf sizes answer alike only while every primitive lies inside its leaf's box, which the reference's
    # builder does not guarantee (predicate and bin can disagree by rounding), so the comparison fixes the size.
    sb = random_scene(seed, max_leaf=4)
    desc = sb.desc()
    rays = random_rays(seed, n_rays)
    osc = Oracle().scene(desc)
    if int(osc.top_stats()[0]) != 0 or any(int(osc.mesh_stats(m)[0]) != 0 for m in range(len(sb.meshes))):
        return {"defined": False}
    want = Ref().scene(desc).hit(rays, nthreads=1)
    got = osc.trace_closest(rays, nthreads=1, full=True)
    hit = want["hit"] != 0
    ghit = got["hits"]["prim"] != abi.PTX_NO_HIT
    bad = ghit != hit
    bad |= hit & (bits(got["hits"]["distance"]) != bits(want["distance"]))
    bad |= hit & (bits(got["pos"]) != bits(want["pos"])).any(axis=1)
    bad |= hit & (got["material"] != want["material"])
    bad |= hit & ((np.abs(got["normal"] - want["normal"]) > 1e-6).any(axis=1) | (np.abs(got["uv"] - want["uv"]) > 1e-6).any(axis=1))
    return {"defined": True, "rays": int(n_rays), "hits": int(hit.sum()), "mismatches": int(bad.sum())}


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    assert Ref.available()
    tot = {"scenes": 0, "set_aside_reference_bins_out_of_range": 0, "compared": 0, "rays": 0, "hits": 0, "mismatches": 0, "reference_crashed": 0}
    bad = []
    t0 = time.time()
    devnull = os.open(os.devnull, os.O_WRONLY)
    while time.time() - t0 < budget:
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.close(r)
            os.dup2(devnull, 1)
            os.write(w, json.dumps(compare(seed, 1 << 14)).encode())
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
        tot["scenes"] += 1
        if not data:
            tot["reference_crashed"] += 1
            bad.append({"seed": seed, "crashed": True})
        else:
            res = json.loads(data)
            if not res["defined"]:
                tot["set_aside_reference_bins_out_of_range"] += 1
            else:
                tot["compared"] += 1
                tot["rays"] += res["rays"]
                tot["hits"] += res["hits"]
                tot["mismatches"] += res["mismatches"]
                if res["mismatches"]:
                    bad.append({"seed": seed, "mismatches": res["mismatches"]})
        seed += 1
        if tot["scenes"] % 200 == 0:
            print(f"[soak] {tot} ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
    print(json.dumps({"what": "CPU checker against the reference's own scene classes on random two-level scenes (tests/test_gpu_fuzz.py's generator): hit flag, distance bits, position bits, material; normal and uv to 1e-6",
                      **tot, "scenes_with_differences": bad[:20]}, indent=1))


if __name__ == "__main__":
    main()

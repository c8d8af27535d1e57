"""Soak of the two-level scene kernels: the random scenes and rays of tests/test_gpu_fuzz.py with fresh seeds and many
more rays, the culled order (replenishing kernel, and one ray per lane) against the reference order on the GPU:
hit flag, distance bits, primitive, instance, barycentrics; occlusion flags.  The first 65 536 rays of every scene are
also compared with the CPU checker (which tools/soak_oracle_reference.py compares with the reference's own classes).
The reference order itself runs twice: through the replenishing kernel's reference-order instance (what the comparisons above use)
and one ray per lane.  python3 tools/soak_scenes.py [scenes] [log2_rays] [first_seed] > profiles/rNN_scene_soak.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _oracle import Oracle
from pathtracer_amd import abi, capi
from test_gpu_fuzz import random_scene, random_rays


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
first_seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
oracle = Oracle()
N_CHECKER = 1 << 16
tot = {"checker_rays": 0, "closest_mismatches_vs_checker": 0, "occlusion_mismatches_vs_checker": 0, "scenes": 0, "closest_rays": 0, "occlusion_rays": 0, "closest_mismatches_persistent": 0, "closest_mismatches_plain": 0, "occlusion_mismatches_persistent": 0,
       "occlusion_mismatches_plain": 0, "closest_mismatches_reference_plain": 0, "occlusion_mismatches_reference_plain": 0, "hit_fraction_sum": 0.0}
bad_seeds = []
t0 = time.time()
for k in range(n_scenes):
    seed = first_seed + k
    sb = random_scene(seed)
    rays = random_rays(seed, n)
    shadow = rays.copy()
    shadow["tmin"], shadow["tmax"] = 1e-4, np.float32(2.5)
    sc = capi.Scene(sb.desc(), keep=sb)
    ref = sc.trace_closest(rays, flags=abi.PTX_TRAVERSE_REFERENCE)
    ref_any = sc.trace_any(shadow, flags=abi.PTX_TRAVERSE_REFERENCE)
    hit = ref["hits"]["prim"] != abi.PTX_NO_HIT
    m = min(n, N_CHECKER)
    osc = oracle.scene(sb.desc())
    want = osc.trace_closest(rays[:m], nthreads=16)
    want_any = osc.trace_any(shadow[:m], nthreads=16)
    osc.close()
    whit = want["hits"]["prim"] != abi.PTX_NO_HIT
    cbad = (bits(ref["hits"]["distance"][:m]) != bits(want["hits"]["distance"])) | (ref["hits"]["prim"][:m] != want["hits"]["prim"])
    cbad |= whit & ((ref["inst"][:m] != want["inst"]) | (bits(ref["bary"][:m]) != bits(want["bary"])).any(axis=1))
    tot["checker_rays"] += m
    tot["closest_mismatches_vs_checker"] += int(cbad.sum())
    tot["occlusion_mismatches_vs_checker"] += int((ref_any[:m] != want_any).sum())
    if cbad.any() or (ref_any[:m] != want_any).any():
        bad_seeds.append({"seed": seed, "kernel": "reference order vs checker", "closest": int(cbad.sum()), "occlusion": int((ref_any[:m] != want_any).sum())})
    for name, mode in (("persistent", abi.PTX_TRAVERSE_CULLED), ("plain", abi.PTX_TRAVERSE_CULLED | abi.PTX_NO_PERSISTENT),
                       ("reference_plain", abi.PTX_TRAVERSE_REFERENCE | abi.PTX_NO_PERSISTENT)):
        got = sc.trace_closest(rays, flags=mode)
        bad = (bits(got["hits"]["distance"]) != bits(ref["hits"]["distance"])) | (got["hits"]["prim"] != ref["hits"]["prim"])
        bad |= hit & ((got["inst"] != ref["inst"]) | (bits(got["bary"]) != bits(ref["bary"])).any(axis=1))
        m = int(bad.sum())
        tot["closest_mismatches_" + name] += m
        ma = int((sc.trace_any(shadow, flags=mode) != ref_any).sum())
        tot["occlusion_mismatches_" + name] += ma
        if m or ma:
            bad_seeds.append({"seed": seed, "kernel": name, "closest": m, "occlusion": ma, "first_rays": np.flatnonzero(bad)[:4].tolist()})
    tot["scenes"] += 1
    tot["closest_rays"] += n
    tot["occlusion_rays"] += n
    tot["hit_fraction_sum"] += float(hit.mean())
    sc.close()
    if k % 10 == 9:
        print(f"[soak] {k + 1} scenes, {tot} ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
tot["hit_fraction_mean"] = round(tot.pop("hit_fraction_sum") / max(1, tot["scenes"]), 4)
print(json.dumps({"what": f"random two-level scenes (tests/test_gpu_fuzz.py's generator, seeds {first_seed}..): the culled order (replenishing kernel; one ray per lane) and the one-ray-per-lane reference "
                          "order against the replenishing kernel's reference order on the GPU; the first 65 536 closest-hit and shadow rays of every scene also against the CPU checker",
                  "rays_per_scene": n, **tot, "scenes_with_differences": bad_seeds}, indent=1))

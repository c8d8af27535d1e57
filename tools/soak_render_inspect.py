"""One scene of tools/soak_render.py again, with the differing pixels listed: python3 tools/soak_render_inspect.py SEED"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import soak_render as sr
argv = sys.argv
from pathtracer_amd import abi, capi

k = int(argv[1])
sb, cam = sr.random_lit_scene(k)
desc = sb.desc()
tiles = capi.make_tiles(cam.film_width, cam.film_height, cam.film_samples, 31 + k)
osc, gsc = sr.oracle.scene(desc), capi.Scene(desc, keep=sb)
acc_o, smp_o = osc.render_tiles(cam, tiles, 1, sr.SEED + k, nthreads=16)
io = capi.accumulator_to_image(acc_o, smp_o)
print(f"seed {k}: {cam.film_width}x{cam.film_height}, depth {cam.film_max_ray_depth}, {len(sb.instances)} instances, use_bvh {bool(desc.use_bvh)}, lights {desc.n_delta_lights} delta {desc.n_env_lights} env")
for name, flags in (("default", 0), ("reference order", abi.PTX_RENDER_REFERENCE_ORDER), ("all rays traced", abi.PTX_RENDER_TRACE_UNUSED_RAYS)):
    acc_g, smp_g = gsc.render_tiles(cam, tiles, sr.SEED + k, flags=flags)
    ig = capi.accumulator_to_image(acc_g, smp_g)
    d = np.abs(io - ig)
    rel = (d / np.maximum(np.abs(io), 1e-3)).max(axis=1)
    bad = np.flatnonzero(rel > 1e-3)
    print(f"{name}: {len(bad)} pixels differ")
    for p in bad[:12]:
        print(f"   pixel ({p % cam.film_width}, {p // cam.film_width}): checker {io[p].round(5).tolist()} gpu {ig[p].round(5).tolist()}")

# which light (if any) the differences belong to: the scene again with one delta light at a time, and with none
lights = list(sb.delta_lights)
print("delta lights:", [(l.type, round(l.intensity, 3), l.inner_angle if hasattr(l, "inner_angle") else None) for l in lights])
for keep in [None] + list(range(len(lights))):
    sb.delta_lights = [] if keep is None else [lights[keep]]
    desc = sb.desc()
    osc, gsc = sr.oracle.scene(desc), capi.Scene(desc, keep=sb)
    acc_o, smp_o = osc.render_tiles(cam, tiles, 1, sr.SEED + k, nthreads=16)
    acc_g, smp_g = gsc.render_tiles(cam, tiles, sr.SEED + k)
    io, ig = capi.accumulator_to_image(acc_o, smp_o), capi.accumulator_to_image(acc_g, smp_g)
    d = np.abs(io - ig)
    rel = (d / np.maximum(np.abs(io), 1e-3)).max(axis=1)
    print(f"only light {keep}: {int((rel > 1e-3).sum())} pixels differ, largest absolute difference {d.max():.3e}, image mean {io.mean():.4f}")
    osc.close(), gsc.close()

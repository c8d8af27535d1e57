"""Profiling driver: one render of a benchmark leg.
    python3 tools/prof_render.py [spp] [leg: cbox | refract_cow_cbox | instanced_cows | flat_cows] [width]
cbox / refract_cow_cbox = the committed scene fixtures of the reference's files (configs 1, 2, 4), instanced_cows = 1,728
instances of the cow (config 5, instanced form), flat_cows = the same cows baked into one 10.1 M-triangle mesh."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pathtracer_amd import abi, capi
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
name = sys.argv[2] if len(sys.argv) > 2 else "refract_cow_cbox"
w = h = int(sys.argv[3]) if len(sys.argv) > 3 else (256 if name == "cbox" else 1024)
depth = 4 if name == "cbox" else 8
if name == "instanced_cows":
    import scenes
    sb, cam = scenes.instanced_cows(12)
elif name == "flat_cows":
    import scenes
    sb, cam = scenes.flattened_cow_scene(12)
else:
    g = np.load(os.path.join(ROOT, "tests", "golden", f"scene_{name}.npz"), allow_pickle=False)
    sb, cams = abi.builder_from_arrays(g)
    cam = cams["Camera_instance"]
cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth, cam.aspect_ratio = w, h, spp, depth, 1.0
scene = capi.Scene(sb.desc(), keep=sb)
tiles = capi.make_tiles(w, h, spp, 4660)
scene.render_tiles(cam, capi.make_tiles(w, h, 1, 1), 1)
c0 = scene.counters()
t0 = time.perf_counter()
acc, smp = scene.render_tiles(cam, tiles, 4660)
dt = time.perf_counter() - t0
c1 = scene.counters()
rays = (c1.render_main_rays - c0.render_main_rays) + (c1.render_direct_rays - c0.render_direct_rays)
print(f"{name} {w}x{h} {spp} spp depth {depth}: {dt:.3f} s, {w*h*spp/dt/1e6:.1f} Mspp/s (host accumulator, includes copies), mean {acc.sum()/(1<<24)/(3.0*w*h*spp):.5f}, "
      f"{rays / (w * h * spp):.3f} rays per path")

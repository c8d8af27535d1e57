"""Profiling driver: one render of the refract_cow_cbox fixture (1024x1024, N spp, depth 8)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pathtracer_amd import abi, capi
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
name = sys.argv[2] if len(sys.argv) > 2 else "refract_cow_cbox"
w = h = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
g = np.load(os.path.join(ROOT, "tests", "golden", f"scene_{name}.npz"), allow_pickle=False)
sb, cams = abi.builder_from_arrays(g)
cam = cams["Camera_instance"]
cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth, cam.aspect_ratio = w, h, spp, 8, 1.0
scene = capi.Scene(sb.desc(), keep=sb)
tiles = capi.make_tiles(w, h, spp, 4660)
scene.render_tiles(cam, capi.make_tiles(w, h, 1, 1), 1)
t0 = time.perf_counter()
acc, smp = scene.render_tiles(cam, tiles, 4660)
dt = time.perf_counter() - t0
print(f"{name} {w}x{h} {spp} spp: {dt:.3f} s, {w*h*spp/dt/1e6:.1f} Mspp/s (host accumulator, includes copies), mean {acc.sum()/(1<<24)/(3.0*w*h*spp):.5f}")

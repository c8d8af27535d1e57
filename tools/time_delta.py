"""Timing driver: a render of the test suite's instanced room with three delta lights (point, spot, directional).
   python3 tools/time_delta.py [spp] [width] [depth]
Prints seconds per render and the accumulator's checksum (equal across library variants: PTX_AMD_LIB=...)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes
from pathtracer_amd import abi, capi

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
w = h = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 4
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sb = scenes.instanced_scene()
sb.add_delta_light(abi.PTX_LIGHT_POINT, (1, 0.9, 0.8), 2.0, scenes.trs((0.5, 0.5, 0.5)))
sb.add_delta_light(abi.PTX_LIGHT_SPOT, (0.3, 0.9, 0.8), 3.0, scenes.trs((-0.5, 0.9, 0.5), axis=(1, 0, 0), deg=20), 40.0, 70.0)
sb.add_delta_light(abi.PTX_LIGHT_DIRECTIONAL, (0.5, 0.5, 1.0), 0.7, scenes.trs(axis=(0, 0, 1), deg=35))
cam = abi.PtxCamera()
cam.vertical_fov, cam.aspect_ratio = 60.0, 1.0
cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth = w, h, spp, depth
T = scenes.trs((0, 0, 3.4))
for i in range(16):
    cam.camera_to_world[i] = float(T[i])
scene = capi.Scene(sb.desc(), keep=sb)
tiles = capi.make_tiles(w, h, spp, 4660)
scene.render_tiles(cam, capi.make_tiles(w, h, 1, 1), 1, flags=flags)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    acc, smp = scene.render_tiles(cam, tiles, 4660, flags=flags)
    best = min(best, time.perf_counter() - t0)
print(f"delta-lit room {w}x{h} {spp} spp depth {depth} flags {flags}: {best:.3f} s, {w * h * spp / best / 1e6:.1f} Mspp/s, checksum {int(acc.sum())}")

"""Timing driver for the render legs (device-resident accumulator, best of N):
   python3 tools/time_render.py [config ...]      config: 2 (refract_cow_cbox 1024^2 x 256 spp), 5 (12^3 cows 1024^2 x 64 spp), 5f (the same cows as one flattened mesh),
                                                  1 (cbox 256^2 x 16 spp), d (delta-lit test room 1024^2 x 64 spp)
Prints seconds per render and the accumulator's checksum (equal across library variants: PTX_AMD_LIB=...)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import scenes
from pathtracer_amd import abi, capi


def fixture(name):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"scene_{name}.npz"), allow_pickle=False)
    sb, cams = abi.builder_from_arrays(g)
    return sb, cams["Camera_instance"]


def delta_room():
    sb = scenes.instanced_scene()
    sb.add_delta_light(abi.PTX_LIGHT_POINT, (1, 0.9, 0.8), 2.0, scenes.trs((0.5, 0.5, 0.5)))
    sb.add_delta_light(abi.PTX_LIGHT_SPOT, (0.3, 0.9, 0.8), 3.0, scenes.trs((-0.5, 0.9, 0.5), axis=(1, 0, 0), deg=20), 40.0, 70.0)
    sb.add_delta_light(abi.PTX_LIGHT_DIRECTIONAL, (0.5, 0.5, 1.0), 0.7, scenes.trs(axis=(0, 0, 1), deg=35))
    cam = abi.PtxCamera()
    cam.vertical_fov = 60.0
    T = scenes.trs((0, 0, 3.4))
    for i in range(16):
        cam.camera_to_world[i] = float(T[i])
    return sb, cam


CONFIGS = {
    "1": ("cbox 256x256 16 spp depth 4", lambda: fixture("cbox"), 256, 16, 4),
    "2": ("refract_cow_cbox 1024x1024 256 spp depth 8", lambda: fixture("refract_cow_cbox"), 1024, 256, 8),
    "2b": ("refract_cow_cbox 2048x2048 64 spp depth 8", lambda: fixture("refract_cow_cbox"), 2048, 64, 8),
    "2c": ("refract_cow_cbox 4096x4096 16 spp depth 8", lambda: fixture("refract_cow_cbox"), 4096, 16, 8),
    "2d": ("refract_cow_cbox 512x512 1024 spp depth 8", lambda: fixture("refract_cow_cbox"), 512, 1024, 8),
    "5": ("12^3 instanced cows 1024x1024 64 spp depth 8", lambda: scenes.instanced_cows(12), 1024, 64, 8),
    "5f": ("12^3 cows baked into one 10.1 M-triangle mesh 1024x1024 64 spp depth 8", lambda: scenes.flattened_cow_scene(12), 1024, 64, 8),
    "d": ("delta-lit room 1024x1024 64 spp depth 4", delta_room, 1024, 64, 4),
}
reps = int(os.environ.get("REPS", "3"))
for key in (sys.argv[1:] or ["2", "5"]):
    what, make, w, spp, depth = CONFIGS[key]
    sb, cam = make()
    cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth, cam.aspect_ratio = w, w, spp, depth, 1.0
    scene = capi.Scene(sb.desc(), keep=sb)
    tiles = capi.make_tiles(w, w, spp, 4660)
    acc = torch.zeros(w * w * 3, dtype=torch.int64, device="cuda")
    smp = torch.zeros(w * w, dtype=torch.int32, device="cuda")
    scene.render_tiles(cam, capi.make_tiles(w, w, min(spp, 2), 1), 1, acc, smp)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        acc.zero_(), smp.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        scene.render_tiles(cam, tiles, 4660, acc, smp)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"config {key}: {what}: {best:.4f} s, {w * w * spp / best / 1e6:.1f} Mspp/s, checksum {int(acc.sum().item())}", flush=True)
    scene.close()

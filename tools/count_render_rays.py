"""Rays a render traces per path, with the default (rays whose answer cannot reach the image are left out) and with
PTX_RENDER_TRACE_UNUSED_RAYS (every ray the reference traces):  python3 tools/count_render_rays.py
Read from the library's device-side totals (ptx_get_counters: render_main_rays, render_direct_rays)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes
from pathtracer_amd import abi, capi


def fixture(name):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"scene_{name}.npz"), allow_pickle=False)
    sb, cams = abi.builder_from_arrays(g)
    return sb, cams["Camera_instance"]


for what, (sb, cam), w, spp, depth in (("config 1: cbox 256x256 depth 4", fixture("cbox"), 256, 16, 4), ("config 2: refract_cow_cbox 1024x1024 depth 8", fixture("refract_cow_cbox"), 1024, 16, 8),
                                       ("config 5: instanced cows 1024x1024 depth 8", scenes.instanced_cows(12), 1024, 8, 8)):
    cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth, cam.aspect_ratio = w, w, spp, depth, 1.0
    sc = capi.Scene(sb.desc(), keep=sb)
    for flags, name in ((0, "default"), (abi.PTX_RENDER_TRACE_UNUSED_RAYS, "every ray")):
        c0 = sc.counters()
        sc.render_tiles(cam, capi.make_tiles(w, w, spp, 4660), 4660, flags=flags)
        c1 = sc.counters()
        main, direct, paths = c1.render_main_rays - c0.render_main_rays, c1.render_direct_rays - c0.render_direct_rays, c1.paths - c0.paths
        print(f"{what}, {name}: {paths} paths, {main} main rays, {direct} direct rays: {(main + direct) / paths:.2f} rays per path", flush=True)
    sc.close()

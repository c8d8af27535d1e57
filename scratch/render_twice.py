"""The same full render twice in one process: the first pays for growing the state buffers."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pathtracer_amd import abi, capi
g = np.load(os.path.join(ROOT, "tests", "golden", "scene_refract_cow_cbox.npz"), allow_pickle=False)
sb, cams = abi.builder_from_arrays(g)
cam = cams["Camera_instance"]
w = h = 1024; spp = 256
cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth, cam.aspect_ratio = w, h, spp, 8, 1.0
scene = capi.Scene(sb.desc(), keep=sb)
import torch
acc = torch.zeros(w * h * 3, dtype=torch.int64, device="cuda"); smp = torch.zeros(w * h, dtype=torch.int32, device="cuda")
scene.render_tiles(cam, capi.make_tiles(w, h, 2, 1), 1, acc, smp)
torch.cuda.synchronize()
tiles = capi.make_tiles(w, h, spp, 4660)
for rep in range(3):
    t0 = time.perf_counter()
    scene.render_tiles(cam, tiles, 4660, acc, smp)
    torch.cuda.synchronize()
    print(f"render {rep}: {time.perf_counter() - t0:.3f} s", flush=True)

"""Config 5 stand-in: 12^3 = 1728 instances of the 5856-triangle cow (10.1 M instanced triangles), all Lambertian,
one emissive quad above, 1024x1024, 64 spp, depth 8."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from pathtracer_amd import abi, capi
import scenes
g = np.load(os.path.join(ROOT, "tests", "golden", "scene_refract_cow_cbox.npz"), allow_pickle=False)
cow_i = max(range(int(g["n_meshes"])), key=lambda i: len(g[f"mesh{i}_indices"]))
sb = abi.SceneBuilder()
cow = sb.add_mesh(g[f"mesh{cow_i}_verts"], g[f"mesh{cow_i}_indices"])
quad = sb.add_mesh(*scenes.quad())
white = sb.add_material(abi.PTX_MAT_LAMBERTIAN, sb.add_constant_texture((0.7, 0.7, 0.7)))
red = sb.add_material(abi.PTX_MAT_LAMBERTIAN, sb.add_constant_texture((0.7, 0.3, 0.3)))
lamp = sb.add_material(abi.PTX_MAT_EMISSIVE, sb.add_constant_texture((1, 1, 1), 6.0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for ix in range(n):
    for iy in range(n):
        for iz in range(n):
            sb.add_instance(cow, white if (ix + iy + iz) % 2 else red, scenes.trs((ix * 1.2, iy * 1.2, iz * 1.2), axis=(0, 1, 0), deg=17.0 * (ix + 2 * iy + 3 * iz), scale=(0.9, 0.9, 0.9)))
L = n * 1.2
sb.add_instance(quad, white, scenes.trs((L / 2, -0.8, L / 2), scale=(L, 1, L)))
sb.add_instance(quad, lamp, scenes.trs((L / 2, L + 1.0, L / 2), axis=(1, 0, 0), deg=180, scale=(L / 2, 1, L / 2)))
t0 = time.perf_counter()
scene = capi.Scene(sb.desc(), keep=sb)
print(f"scene build {time.perf_counter() - t0:.2f} s: {n ** 3} instances x {len(g[f'mesh{cow_i}_indices'])} triangles = {n ** 3 * len(g[f'mesh{cow_i}_indices']) / 1e6:.1f} M instanced triangles")
w = h = 1024
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cam = abi.PtxCamera()
cam.vertical_fov, cam.aspect_ratio = 50.0, 1.0
cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth = w, h, spp, 8
T = scenes.trs((L / 2, L * 0.7, L * 2.4), axis=(1, 0, 0), deg=-12)
for i in range(16):
    cam.camera_to_world[i] = float(T[i])
scene.render_tiles(cam, capi.make_tiles(w, h, 1, 1), 1)
t0 = time.perf_counter()
acc, smp = scene.render_tiles(cam, capi.make_tiles(w, h, spp, 4660), 4660)
dt = time.perf_counter() - t0
print(f"render {w}x{h} {spp} spp depth 8: {dt:.3f} s, {w * h * spp / dt / 1e6:.1f} Mspp/s, mean radiance {acc.sum() / (1 << 24) / (3.0 * w * h * spp):.5f}")

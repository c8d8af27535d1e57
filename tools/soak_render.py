"""Soak of the render path: random lit scenes (every material kind, image and constant textures, point / spot /
directional lights, environment maps, spheres, mirrored and coincident instances), random cameras; the GPU render
against the CPU checker's render with the same counter-based random stream.  Per scene: the fraction of pixels whose
relative difference exceeds 1e-3 (device sin/cos/pow differ from libm in the last bit, which can send a sample down
another path) and the RMSE over the mean.  python3 tools/soak_render.py [scenes] > profiles/rNN_render_soak.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes
from _oracle import Oracle
from pathtracer_amd import abi, capi, synth

SEED = 0x5EED5EED


def random_lit_scene(seed):
    g = np.random.Generator(np.random.Philox(key=9000 + seed))
    sb = abi.SceneBuilder(use_bvh=seed % 6 != 0, fresnel_mode=abi.PTX_FRESNEL_GCC_INT_ABS if seed % 2 else abi.PTX_FRESNEL_FLOAT_ABS)
    tex = [sb.add_constant_texture(tuple(g.random(3) * 0.8 + 0.1)) for _ in range(3)]
    tex.append(sb.add_image_texture(g.random((8, 8, 3), dtype=np.float32), abi.PTX_SAMPLER_NEAREST if seed % 3 else abi.PTX_SAMPLER_BILINEAR))
    glow = sb.add_constant_texture(tuple(g.random(3) * 0.5 + 0.5), scale=float(g.random() * 6 + 1))
    mats = [sb.add_material(abi.PTX_MAT_LAMBERTIAN, tex[0]), sb.add_material(abi.PTX_MAT_LAMBERTIAN, tex[3]), sb.add_material(abi.PTX_MAT_MIRROR, tex[1]),
            sb.add_material(abi.PTX_MAT_GLASS, tex[1], tex[2], float(g.random() * 0.8 + 1.1)), sb.add_material(abi.PTX_MAT_REFRACT, -1, tex[2], float(g.random() * 0.8 + 1.1)),
            sb.add_material(abi.PTX_MAT_EMISSIVE, glow), sb.add_material(abi.PTX_MAT_LAMBERTIAN, tex[1])]
    mesh_bvh = seed % 7 != 0
    meshes = [sb.add_mesh(*scenes.quad(), use_bvh=mesh_bvh), sb.add_mesh(*scenes.icosphere(int(g.integers(0, 3))), use_bvh=mesh_bvh),
              sb.add_mesh(*synth.soup(int(g.integers(20, 300)), key=seed, extent=1.0, half=float(g.random() * 0.2 + 0.05)), use_bvh=mesh_bvh)]
    sb.add_instance(meshes[0], mats[0], scenes.trs((0, -1.5, 0), scale=(4, 1, 4)))   # a floor
    transforms = []
    for _ in range(int(g.integers(2, 16))):
        scale = g.random(3) * 1.2 + 0.2
        if g.random() < 0.2:
            scale[int(g.integers(0, 3))] *= -1.0
        T = scenes.trs(g.random(3) * 3 - 1.5, axis=g.standard_normal(3), deg=float(g.random() * 360), scale=scale)
        if transforms and g.random() < 0.1:
            T = transforms[int(g.integers(0, len(transforms)))]
        transforms.append(T)
        m = mats[int(g.integers(0, len(mats)))] if g.random() < 0.9 else -1
        if g.random() < 0.3:
            sb.add_instance(-1, m, T, sphere_radius=float(g.random() * 0.6 + 0.1))
        else:
            sb.add_instance(meshes[int(g.integers(0, len(meshes)))], m, T)
    for _ in range(int(g.integers(0, 3))):
        kind = (abi.PTX_LIGHT_POINT, abi.PTX_LIGHT_SPOT, abi.PTX_LIGHT_DIRECTIONAL)[int(g.integers(0, 3))]
        sb.add_delta_light(kind, tuple(g.random(3) * 0.7 + 0.3), float(g.random() * 3 + 0.5),
                           scenes.trs(g.random(3) * 2 - 1 + np.array([0, 1.0, 0]), axis=g.standard_normal(3), deg=float(g.random() * 90)), 30.0, 60.0)
    if g.random() < 0.5:
        if g.random() < 0.5:
            sb.add_env_light(abi.PTX_ENV_SPHERE, sb.add_image_texture(g.random((8, 16, 3), dtype=np.float32), abi.PTX_SAMPLER_BILINEAR))
        else:
            sb.add_env_light(abi.PTX_ENV_HEMISPHERE, sb.add_constant_texture(tuple(g.random(3) * 0.6 + 0.1)))
    cam = abi.PtxCamera()
    w, h = (96, 64) if seed % 2 else (64, 96)
    cam.vertical_fov, cam.aspect_ratio = float(g.random() * 50 + 35), w / h
    cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth = w, h, 4, int(g.integers(1, 7))
    T = scenes.trs((float(g.random() * 2 - 1), float(g.random() * 1.5), float(g.random() * 2 + 3.0)), axis=(0, 1, 0), deg=float(g.random() * 30 - 15))
    for i in range(16):
        cam.camera_to_world[i] = float(T[i])
    return sb, cam


oracle = Oracle()


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    fr, rm, worst = [], [], []
    t0 = time.time()
    for k in range(n_scenes):
        sb, cam = random_lit_scene(k)
        desc = sb.desc()
        tiles = capi.make_tiles(cam.film_width, cam.film_height, cam.film_samples, 31 + k)
        osc, gsc = oracle.scene(desc), capi.Scene(desc, keep=sb)
        acc_o, smp_o = osc.render_tiles(cam, tiles, 1, SEED + k, nthreads=16)
        acc_g, smp_g = gsc.render_tiles(cam, tiles, SEED + k)
        osc.close(), gsc.close()
        assert (smp_o == smp_g).all()
        io, ig = capi.accumulator_to_image(acc_o, smp_o), capi.accumulator_to_image(acc_g, smp_g)
        d = np.abs(io - ig)
        rel = (d / np.maximum(np.abs(io), 1e-3)).max(axis=1)
        f, r = float((rel > 1e-3).mean()), float(np.sqrt((d ** 2).mean()) / max(io.mean(), 1e-6))
        fr.append(f), rm.append(r)
        worst.append((f, k))
        if k % 50 == 49:
            print(f"[soak] {k + 1} scenes: pixels off mean {np.mean(fr):.5f} max {np.max(fr):.5f}, rmse/mean mean {np.mean(rm):.2e} max {np.max(rm):.2e} ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
    worst.sort(reverse=True)
    fr, rm = np.array(fr), np.array(rm)
    print(json.dumps({"what": "GPU render against the CPU checker's render, same counter-based random stream, random lit scenes and cameras (4 spp, depth 1-6, 96x64 or 64x96)",
                      "scenes": n_scenes, "pixels_off_by_more_than_1e-3_relative": {"mean": float(fr.mean()), "median": float(np.median(fr)), "p99": float(np.percentile(fr, 99)), "max": float(fr.max())},
                      "rmse_over_mean": {"mean": float(rm.mean()), "median": float(np.median(rm)), "p99": float(np.percentile(rm, 99)), "max": float(rm.max())},
                      "scenes_with_no_pixel_off": int((fr == 0).sum()), "scenes_within_the_test_suite_tolerance": int(((fr <= 0.005) & (rm <= 1e-3)).sum()),
                      "worst_scenes": [{"seed": k, "pixels_off": f} for f, k in worst[:5]]}, indent=1))


if __name__ == "__main__":
    main()

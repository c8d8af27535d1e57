"""Experiment: GPU render vs the CPU checker with the same counter-based random stream."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from pathtracer_amd import abi, capi
import scenes
from _oracle import Oracle

def camera(w, h, spp, depth, pos=(0, 0, 3.4)):
    cam = abi.PtxCamera()
    cam.vertical_fov, cam.aspect_ratio = 60.0, w / h
    cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth = w, h, spp, depth
    T = scenes.trs(pos)
    for i in range(16):
        cam.camera_to_world[i] = float(T[i])
    return cam

o = Oracle()
for fres in (abi.PTX_FRESNEL_GCC_INT_ABS, abi.PTX_FRESNEL_FLOAT_ABS):
    sb = scenes.instanced_scene(fresnel_mode=fres)
    sb.add_delta_light(abi.PTX_LIGHT_POINT, (1, 0.9, 0.8), 2.0, scenes.trs((0.5, 0.5, 0.5)))
    sb.add_delta_light(abi.PTX_LIGHT_SPOT, (0.3, 0.9, 0.8), 3.0, scenes.trs((-0.5, 0.9, 0.5), axis=(1, 0, 0), deg=20), 40.0, 70.0)
    desc = sb.desc()
    w, h, spp, depth = 96, 64, 8, 4
    cam = camera(w, h, spp, depth)
    tiles = capi.make_tiles(w, h, spp, 1234)
    osc = o.scene(desc)
    t0 = time.time()
    acc_o, smp_o = osc.render_tiles(cam, tiles, 1, 0xABCDEF12345)
    t1 = time.time()
    acc_m, smp_m = osc.render_tiles(cam, tiles, 0, 0)
    gsc = capi.Scene(desc, keep=sb)
    t2 = time.time()
    acc_g, smp_g = gsc.render_tiles(cam, tiles, 0xABCDEF12345)
    t3 = time.time()
    print("oracle s", t1 - t0, "gpu s", t3 - t2)
    assert (smp_o == smp_g).all()
    io, ig, im = (capi.accumulator_to_image(a, smp_o) for a in (acc_o, acc_g, acc_m))
    d = np.abs(io - ig)
    rel = d / np.maximum(np.abs(io), 1e-3)
    print("fresnel", fres, "mean img", io.mean(), ig.mean(), im.mean())
    print("  pixels rel>1e-4:", (rel.max(axis=1) > 1e-4).mean(), " rel>1e-2:", (rel.max(axis=1) > 1e-2).mean(), "max abs", d.max())
    print("  exact acc equal frac:", (acc_o == acc_g).mean())
    print("  rmse gpu-vs-philox", np.sqrt((d ** 2).mean()), " rmse mt-vs-philox", np.sqrt(((io - im) ** 2).mean()))
    bad = np.argsort(-rel.max(axis=1))[:5]
    for b in bad:
        print("   px", b % w, b // w, io[b], ig[b])

"""ptx_render_tiles (the Pathtracer::do_trace / trace / accumulate path, pathtracer.cpp:126-173, 407-477)
against the CPU checker.

Two comparisons:
  * same counter-based random stream on both sides: every path takes the same decisions, so the
    images agree to float rounding (device sin/cos/atan2/acos/pow may differ from libm in the last
    bit, and the device sums radiance front-to-back where the recursion sums back-to-front).
    Tolerance: per-pixel relative difference <= 1e-3 on >= 99.5 % of the pixels (a last-bit
    difference can flip a Fresnel coin or a hit and send one sample down another path) and
    RMSE <= 1e-3 * mean.
  * the checker with the reference's own mt19937-per-tile stream: a different sample set of the same
    estimator, so the images agree statistically; tolerance: RMSE <= 4x the RMSE between two
    independent checker renders.
"""
import numpy as np
import pytest

from pathtracer_amd import abi, capi
import scenes

pytestmark = pytest.mark.gpu

SEED = 0xABCDEF12345


def camera(w, h, spp, depth, pos=(0, 0, 3.4), axis=(0, 1, 0), deg=0.0, fov=60.0):
    cam = abi.PtxCamera()
    cam.vertical_fov, cam.aspect_ratio = fov, w / h
    cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth = w, h, spp, depth
    T = scenes.trs(pos, axis=axis, deg=deg)
    for i in range(16):
        cam.camera_to_world[i] = float(T[i])
    return cam


def lit_scene(kind, **kw):
    sb = scenes.instanced_scene(**kw)
    if kind in ("delta", "all"):
        sb.add_delta_light(abi.PTX_LIGHT_POINT, (1, 0.9, 0.8), 2.0, scenes.trs((0.5, 0.5, 0.5)))
        sb.add_delta_light(abi.PTX_LIGHT_SPOT, (0.3, 0.9, 0.8), 3.0, scenes.trs((-0.5, 0.9, 0.5), axis=(1, 0, 0), deg=20), 40.0, 70.0)
        sb.add_delta_light(abi.PTX_LIGHT_DIRECTIONAL, (0.5, 0.5, 1.0), 0.7, scenes.trs(axis=(0, 0, 1), deg=35))
    if kind in ("env", "all"):
        g = np.random.Generator(np.random.Philox(key=5))
        img = g.random((16, 32, 3), dtype=np.float32)
        sb.add_env_light(abi.PTX_ENV_SPHERE, sb.add_image_texture(img, abi.PTX_SAMPLER_BILINEAR))
        sb.add_env_light(abi.PTX_ENV_HEMISPHERE, sb.add_constant_texture((0.2, 0.3, 0.5)))
    return sb


def images(oracle, sb, cam, tiles):
    desc = sb.desc()
    osc = oracle.scene(desc)
    gsc = capi.Scene(desc, keep=sb)
    try:
        acc_o, smp_o = osc.render_tiles(cam, tiles, 1, SEED)
        acc_g, smp_g = gsc.render_tiles(cam, tiles, SEED)
    finally:
        osc.close()
        gsc.close()
    assert (smp_o == smp_g).all()
    return capi.accumulator_to_image(acc_o, smp_o), capi.accumulator_to_image(acc_g, smp_g)


def assert_same_stream(io, ig, rmse_rel=1e-3):
    d = np.abs(io - ig)
    rel = (d / np.maximum(np.abs(io), 1e-3)).max(axis=1)
    assert (rel <= 1e-3).mean() >= 0.995, f"{(rel > 1e-3).mean():.4f} of the pixels differ"
    assert np.sqrt((d ** 2).mean()) <= rmse_rel * max(io.mean(), 1e-6)


@pytest.mark.parametrize("kind,use_bvh,fresnel", [
    ("delta", True, abi.PTX_FRESNEL_GCC_INT_ABS),
    ("delta", True, abi.PTX_FRESNEL_FLOAT_ABS),
    ("env", True, abi.PTX_FRESNEL_GCC_INT_ABS),
    ("all", False, abi.PTX_FRESNEL_FLOAT_ABS),
    ("none", True, abi.PTX_FRESNEL_GCC_INT_ABS),
])
def test_render_same_stream(oracle, kind, use_bvh, fresnel):
    sb = lit_scene(kind, use_bvh=use_bvh, mesh_bvh=use_bvh, fresnel_mode=fresnel)
    w, h, spp = 120, 72, 6
    cam = camera(w, h, spp, 4)
    tiles = capi.make_tiles(w, h, spp, 77)
    io, ig = images(oracle, sb, cam, tiles)
    assert io.mean() > 0.01
    assert_same_stream(io, ig)


def test_render_open_scene_camera_inside_env(oracle):
    """No room: most camera rays leave the scene and pick up the environment."""
    sb = abi.SceneBuilder()
    g = np.random.Generator(np.random.Philox(key=6))
    env = sb.add_image_texture(g.random((8, 16, 3), dtype=np.float32), abi.PTX_SAMPLER_NEAREST)
    sb.add_env_light(abi.PTX_ENV_SPHERE, env)
    tex = sb.add_image_texture(g.random((32, 32, 3), dtype=np.float32), abi.PTX_SAMPLER_BILINEAR)
    ico = sb.add_mesh(*scenes.icosphere(3))
    sb.add_instance(ico, sb.add_material(abi.PTX_MAT_LAMBERTIAN, tex), scenes.trs((0, 0, 0), axis=(1, 1, 0), deg=40, scale=(1.0, 0.7, 1.2)))
    sb.add_instance(-1, sb.add_material(abi.PTX_MAT_GLASS, env, tex, 1.45), scenes.trs((1.6, 0.2, 0.4), scale=(0.6, 0.6, 0.6)), sphere_radius=1.0)
    w, h, spp = 96, 96, 4
    cam = camera(w, h, spp, 6, pos=(0.8, 0.6, 4.0), axis=(1, 0, 0), deg=-8, fov=45.0)
    io, ig = images(oracle, sb, cam, capi.make_tiles(w, h, spp, 3))
    assert_same_stream(io, ig)


def test_render_depth_zero_and_overlapping_tiles(oracle):
    """max depth 0 shows emission only; tiles that share pixels and split the sample range add up."""
    sb = lit_scene("delta")
    w, h = 64, 48
    cam = camera(w, h, 8, 0)
    tiles = np.zeros(4, dtype=abi.TILE_DTYPE)
    tiles[0] = (1, 0, 64, 0, 48, 0, 3)
    tiles[1] = (2, 10, 50, 5, 40, 3, 8)     # more samples for an inner window
    tiles[2] = (3, 0, 64, 0, 48, 3, 3)      # empty sample range: nothing
    tiles[3] = (4, 20, 20, 0, 48, 0, 8)     # empty pixel range: nothing
    io, ig = images(oracle, sb, cam, tiles)
    assert_same_stream(io, ig)
    cam4 = camera(w, h, 8, 3)
    io, ig = images(oracle, sb, cam4, tiles)
    assert_same_stream(io, ig)


def test_render_matches_reference_stream_statistically(oracle):
    """Against the checker driven by mt19937 per tile (what the reference does): same estimator, other samples."""
    sb = lit_scene("delta")
    desc = sb.desc()
    w, h, spp = 80, 60, 64
    cam = camera(w, h, spp, 4)
    tiles_a = capi.make_tiles(w, h, spp, 11)
    tiles_b = capi.make_tiles(w, h, spp, 12)
    osc = oracle.scene(desc)
    gsc = capi.Scene(desc, keep=sb)
    try:
        ia = capi.accumulator_to_image(*osc.render_tiles(cam, tiles_a, 0, 0))
        ib = capi.accumulator_to_image(*osc.render_tiles(cam, tiles_b, 0, 0))
        ig = capi.accumulator_to_image(*gsc.render_tiles(cam, tiles_a, SEED))
    finally:
        osc.close()
        gsc.close()
    noise = np.sqrt(((ia - ib) ** 2).mean())
    assert np.sqrt(((ig - ia) ** 2).mean()) <= 4 * noise
    assert abs(ig.mean() - ia.mean()) <= 0.05 * ia.mean()
    # tone-mapped 8-bit output of both, as the reference writes it
    ta, tg = capi.tonemap(ia, 1.0), capi.tonemap(ig, 1.0)
    assert abs(ta[:, :3].astype(np.float64).mean() - tg[:, :3].astype(np.float64).mean()) <= 3.0


def test_render_device_accumulator_and_errors():
    torch = pytest.importorskip("torch")
    sb = lit_scene("delta")
    w, h, spp = 64, 64, 4
    cam = camera(w, h, spp, 3)
    tiles = capi.make_tiles(w, h, spp, 5)
    gsc = capi.Scene(sb.desc(), keep=sb)
    acc_h, smp_h = gsc.render_tiles(cam, tiles, SEED)
    acc_d = torch.zeros(w * h * 3, dtype=torch.int64, device="cuda")
    smp_d = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    gsc.render_tiles(cam, tiles, SEED, acc_d, smp_d)
    torch.cuda.synchronize()
    assert (acc_d.cpu().numpy() == acc_h).all()          # same stream, same arithmetic: bit-identical
    assert (smp_d.cpu().numpy().astype(np.uint32) == smp_h).all()
    bad = tiles.copy()
    bad["x_end"][0] = w + 1
    with pytest.raises(capi.PtxError):
        gsc.render_tiles(cam, bad, SEED)
    gsc.close()


GOLDEN = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["cbox", "refract_cow_cbox"])
@pytest.mark.parametrize("mode,fresnel", [("gcc_int_abs", abi.PTX_FRESNEL_GCC_INT_ABS), ("float_abs", abi.PTX_FRESNEL_FLOAT_ABS)])
def test_image_gate_on_reference_scenes(name, mode, fresnel):
    """media/js3d/cbox.js3d and refract_cow_cbox.js3d (committed as descriptor fixtures) at 64x64, 1024 spp:
    RMSE against the checker's 4096 spp image (mt19937-per-tile stream, linear radiance) within 1.5x the RMSE
    an independent 1024 spp checker render has against that image.  Both Fresnel behaviours."""
    import ctypes as C
    g = np.load(f"{GOLDEN}/scene_{name}.npz", allow_pickle=False)
    img = np.load(f"{GOLDEN}/image_{name}_{mode}.npz", allow_pickle=False)
    sb, cams = abi.builder_from_arrays(g)
    sb.fresnel_mode = fresnel
    cam = cams["Camera_instance"]
    w = h = 64
    cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth = w, h, 1024, int(img["depth"])
    cam.aspect_ratio = 1.0
    sc = capi.Scene(sb.desc(), keep=sb)
    acc, smp = sc.render_tiles(cam, capi.make_tiles(w, h, 1024, 31), SEED)
    sc.close()
    assert (smp == 1024).all()
    got = capi.accumulator_to_image(acc, smp)
    ref_img = img["image"].reshape(-1, 3)
    rmse = float(np.sqrt(((got - ref_img) ** 2).mean()))
    assert rmse <= 1.5 * float(img["rmse_1024spp"]), (rmse, float(img["rmse_1024spp"]))
    np.testing.assert_allclose(got.mean(axis=0), img["mean"], rtol=0.03)


def test_large_batch_replenishing_extend_is_bit_identical(oracle):
    """Batches of >= 2^18 paths run the extend stage on the replenishing two-level kernel; smaller ones
    (and PTX_NO_PERSISTENT) keep one ray per lane.  Scheduling differs, arithmetic does not: the two
    accumulators are equal bit for bit, and both agree with the checker's same-stream render."""
    sb = lit_scene("delta")
    desc = sb.desc()
    w = h = 512
    spp = 2
    cam = camera(w, h, spp, 4)
    tiles = capi.make_tiles(w, h, spp, 21)
    gsc = capi.Scene(desc, keep=sb)
    acc_p, smp_p = gsc.render_tiles(cam, tiles, SEED)
    acc_n, smp_n = gsc.render_tiles(cam, tiles, SEED, flags=abi.PTX_NO_PERSISTENT)
    gsc.close()
    assert (smp_p == spp).all() and (smp_n == spp).all()
    assert (acc_p == acc_n).all()
    osc = oracle.scene(desc)
    acc_o, smp_o = osc.render_tiles(cam, tiles, 1, SEED)
    osc.close()
    # at 2 spp one sample that took another path (a last-bit difference in sin/cos/pow) moves its pixel by O(1): wider RMSE bound
    assert_same_stream(capi.accumulator_to_image(acc_o, smp_o), capi.accumulator_to_image(acc_p, smp_p), rmse_rel=1e-2)


@pytest.mark.parametrize("w,spp", [(96, 4), (512, 2)])
def test_render_reference_order_flag_is_bit_identical(w, spp):
    """PTX_RENDER_REFERENCE_ORDER walks every extend and shadow ray of the render in the reference's visit order
    (bvh.cpp:246-375: left subtree first, no distance culling).  The default culled order must give the very same
    accumulator, on the one-ray-per-lane kernels (small film) and on the replenishing ones (>= 2^18 paths),
    in a scene whose delta lights send shadow rays through the occlusion pass."""
    sb = lit_scene("delta")
    cam = camera(w, w, spp, 5)
    tiles = capi.make_tiles(w, w, spp, 9)
    gsc = capi.Scene(sb.desc(), keep=sb)
    acc_c, smp_c = gsc.render_tiles(cam, tiles, SEED)
    acc_r, smp_r = gsc.render_tiles(cam, tiles, SEED, flags=abi.PTX_RENDER_REFERENCE_ORDER)
    gsc.close()
    assert (smp_c == spp).all() and (smp_r == spp).all()
    assert acc_c.any()
    assert (acc_c == acc_r).all(), f"{int((acc_c != acc_r).sum())} accumulator words differ between the two orders"


def _emissive_mix_scene(n_lamps):
    """An open scene (no environment light): a floor, mirrors and glass, and `n_lamps` emitters of every kind the
    scene format has -- quads, an analytic sphere, a quad with an image texture as its emission."""
    sb = abi.SceneBuilder()
    g = np.random.Generator(np.random.Philox(key=17))
    white = sb.add_material(abi.PTX_MAT_LAMBERTIAN, sb.add_constant_texture((0.7, 0.7, 0.7)))
    mirror = sb.add_material(abi.PTX_MAT_MIRROR, sb.add_constant_texture((0.9, 0.9, 0.9)))
    glass = sb.add_material(abi.PTX_MAT_GLASS, sb.add_constant_texture((1, 1, 1)), sb.add_constant_texture((1, 1, 1)), 1.4)
    lamp_const = sb.add_material(abi.PTX_MAT_EMISSIVE, sb.add_constant_texture((1.0, 0.8, 0.6), 5.0))
    lamp_img = sb.add_material(abi.PTX_MAT_EMISSIVE, sb.add_image_texture(g.random((8, 8, 3), dtype=np.float32) * 4, abi.PTX_SAMPLER_BILINEAR))
    q = sb.add_mesh(*scenes.quad())
    ico = sb.add_mesh(*scenes.icosphere(2))
    sb.add_instance(q, white, scenes.trs((0, -1, 0), scale=(3, 1, 3)))
    sb.add_instance(ico, mirror, scenes.trs((-0.9, -0.4, 0.0), axis=(1, 2, 3), deg=20, scale=(0.6, 0.5, 0.6)))
    sb.add_instance(ico, glass, scenes.trs((0.9, -0.4, 0.2), scale=(0.55, 0.55, 0.55)))
    sb.add_instance(-1, white, scenes.trs((0.0, -0.6, -0.9), scale=(0.4, 0.4, 0.4)), sphere_radius=1.0)
    for k in range(n_lamps):
        a = 2 * np.pi * k / max(n_lamps, 1)
        pos = (1.6 * np.cos(a), 0.9 + 0.1 * (k % 3), 1.6 * np.sin(a))
        if k % 3 == 0:
            sb.add_instance(q, lamp_const, scenes.trs(pos, axis=(1, 0, 0), deg=180, scale=(0.4, 1, 0.4)))
        elif k % 3 == 1:
            sb.add_instance(-1, lamp_const, scenes.trs(pos, scale=(0.2, 0.2, 0.2)), sphere_radius=1.0)
        else:
            sb.add_instance(q, lamp_img, scenes.trs(pos, axis=(1, 0, 1), deg=170, scale=(0.5, 1, 0.3)))
    return sb


@pytest.mark.parametrize("what", ["room_small", "room_large", "lamps3", "lamps8", "lamps9_skip_off", "no_lamp", "env"])
@pytest.mark.parametrize("depth", [1, 5])
def test_rays_left_out_do_not_change_the_accumulator(what, depth):
    """By default the render leaves out rays whose answer cannot reach the image: the scattered ray of a path's last
    vertex (pathtracer.cpp:118,163: `.second` of a depth-0 trace() is {}) and direct-lighting rays that touch no
    primitive of an Emissive instance (pathtracer.cpp:64-76: only the emission of the first hit is used).
    PTX_RENDER_TRACE_UNUSED_RAYS traces them all, as the reference does; the accumulators must be equal bit for bit,
    on the one-ray-per-lane kernels and on the replenishing ones, with every kind of emitter, with more emitters
    than the shortcut takes (it is then off), without any, and with an environment light (off as well)."""
    w, spp = (512, 2) if what == "room_large" else (96, 6)
    if what.startswith("room"):
        sb = lit_scene("delta")
    elif what == "env":
        sb = lit_scene("env")
    elif what == "no_lamp":
        sb = _emissive_mix_scene(0)
        sb.add_delta_light(abi.PTX_LIGHT_POINT, (1, 1, 1), 3.0, scenes.trs((0.0, 1.5, 0.5)))
    else:
        sb = _emissive_mix_scene(int(what[5]))
    cam = camera(w, w, spp, depth, pos=(0, 0.3, 3.6))
    tiles = capi.make_tiles(w, w, spp, 13)
    gsc = capi.Scene(sb.desc(), keep=sb)
    acc_d, smp_d = gsc.render_tiles(cam, tiles, SEED)
    acc_a, smp_a = gsc.render_tiles(cam, tiles, SEED, flags=abi.PTX_RENDER_TRACE_UNUSED_RAYS)
    gsc.close()
    assert (smp_d == spp).all() and (smp_a == spp).all()
    assert acc_d.any()
    assert (acc_d == acc_a).all(), f"{int((acc_d != acc_a).sum())} accumulator words differ"


def test_emissive_scene_same_stream(oracle):
    """The emitter mix against the CPU checker driven by the same random stream (the checker traces every ray)."""
    sb = _emissive_mix_scene(6)
    w, h, spp = 96, 64, 6
    cam = camera(w, h, spp, 4, pos=(0, 0.3, 3.6))
    io, ig = images(oracle, sb, cam, capi.make_tiles(w, h, spp, 5))
    assert io.mean() > 0.01
    assert_same_stream(io, ig)


@pytest.mark.parametrize("extra", [4, 5])
def test_same_stream_around_the_shade_stage_instance_table_limit(oracle, extra):
    """The shade stage reads the instance records of a scene with at most 16 instances from a copy in LDS and those of a
    larger scene from memory: the test room with 16 and with 17 instances, each against the CPU checker driven by the
    same random stream (Instance::hit and Trace::transform use the same two matrices either way, instance.h:32-40)."""
    sb = scenes.instanced_scene()
    n_before = sb.desc().n_instances
    q = sb.add_mesh(*scenes.quad())
    tex = sb.add_constant_texture((0.3, 0.7, 0.4))
    mat = sb.add_material(abi.PTX_MAT_LAMBERTIAN, tex)
    for k in range(extra):
        sb.add_instance(q, mat, scenes.trs((-1.2 + 0.6 * k, -0.6 + 0.2 * k, 0.5), axis=(1, 0.3 * k, 0), deg=25 + 10 * k, scale=(0.25, 1, 0.2)))
    assert sb.desc().n_instances == n_before + extra == 12 + extra
    w, h, spp = 96, 64, 6
    cam = camera(w, h, spp, 4)
    io, ig = images(oracle, sb, cam, capi.make_tiles(w, h, spp, 9))
    assert io.mean() > 0.01
    assert_same_stream(io, ig)


@pytest.mark.parametrize("w,h,spp", [(30, 20, 5), (41, 25, 3), (33, 31, 7)])
def test_same_stream_on_films_smaller_than_a_path_block(oracle, w, h, spp):
    """Films of fewer pixels than the 1,024 slots whose samples are laid out together (600), of one block plus one
    pixel (1,025) and just short of one (1,023): every path must still be the sample of the pixel the checker renders."""
    sb = lit_scene("none")
    cam = camera(w, h, spp, 3)
    io, ig = images(oracle, sb, cam, capi.make_tiles(w, h, spp, 13))
    assert io.mean() > 0.01
    assert_same_stream(io, ig, rmse_rel=5e-3)

"""The CPU checker (oracle/) against vectors computed by the reference's own code.

tests/golden/*.npz were written by tests/golden/make_golden.py from oracle/_ref/libptref.so (the
reference's translation units compiled as they are).  Integer and float-bit fields must match
exactly; sin/cos/atan2-derived values (sphere uv, cosine sampler) are compared to 1e-6 because a
different libm build may round the last bit differently.
"""
import os

import numpy as np
import pytest

from pathtracer_amd import abi, capi, synth
import scenes
from conftest import bits

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def rays_of(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(abi.RAY_DTYPE).reshape(-1)


def positions_to_verts(pos):
    v = np.zeros((len(pos), 8), dtype=np.float32)
    v[:, :3] = pos
    v[:, 4] = 1.0
    return v


def check_mesh_hits(oracle, verts, idx, rays, g, use_bvh=True, check_attrs=True):
    sb = abi.SceneBuilder()
    sc = oracle.mesh_scene(sb.mesh_desc(verts, idx, use_bvh=use_bvh, max_leaf_size=4))
    r = sc.trace_closest(rays, full=True)
    order = sc.mesh_order(len(idx))      # hits name primitives by their position in the built tree's array
    sc.close()
    assert (bits(r["hits"]["distance"]) == g["distance_bits"]).all()
    prim = r["hits"]["prim"]
    hit = prim != abi.PTX_NO_HIT
    prim = np.where(hit, order[np.minimum(prim, len(order) - 1)], abi.PTX_NO_HIT)
    assert (prim == g["prim"]).all()
    assert hit.sum() > len(rays) // 10
    assert (bits(r["pos"][hit]) == bits(g["pos"][hit])).all()
    if check_attrs:
        assert (bits(r["normal"][hit]) == bits(g["normal"][hit])).all()
        assert (bits(r["uv"][hit]) == bits(g["uv"][hit])).all()


def test_fuzz_mesh_hits(oracle):
    """Reference fuzz generator (tests/a3/test.a3.task3.bvh.fuzz.cpp:13-44, RNG(462), first trial)."""
    g = load("hits_fuzz462.npz")
    verts = positions_to_verts(g["positions"])
    idx = np.arange(len(verts), dtype=np.uint32).reshape(-1, 3)
    # the fixture's normals carry triangle ids, so only distance / primitive / position are compared
    check_mesh_hits(oracle, verts, idx, rays_of(g["rays"]), g, check_attrs=False)


def test_sphere_mesh_hits(oracle):
    """Util::sphere_mesh(1, 3) through Tri_Mesh(BVH) and Tri_Mesh(List) (tri_mesh.cpp:160-196, list.h:25-32)."""
    g = load("hits_sphere3.npz")
    check_mesh_hits(oracle, g["verts"], g["indices"], rays_of(g["rays"]), g)
    gl = load("hits_sphere3_list.npz")
    sb = abi.SceneBuilder()
    sc = oracle.mesh_scene(sb.mesh_desc(g["verts"], g["indices"], use_bvh=False))
    r = sc.trace_closest(rays_of(g["rays"]), full=True)
    sc.close()
    assert (bits(r["hits"]["distance"]) == gl["distance_bits"]).all()
    assert (r["hits"]["prim"] == gl["prim"]).all()


def test_injected_soup_hits(oracle):
    """BVH::hit of the reference over this repo's tree (injected through bvh.h:53-55's public members)."""
    g = load("hits_soup20k.npz")
    verts, idx = synth.soup(20000)
    bvh = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4)
    nodes, order = bvh.nodes(), bvh.order()
    sb = abi.SceneBuilder()
    sc = oracle.mesh_scene(sb.mesh_desc(verts, idx[order], nodes=nodes))
    r = sc.trace_closest(synth.rays(len(g["prim"])), full=True)
    sc.close()
    prim = r["hits"]["prim"]
    hit = prim != abi.PTX_NO_HIT
    prim = np.where(hit, order[np.minimum(prim, len(order) - 1)], abi.PTX_NO_HIT)
    assert (bits(r["hits"]["distance"]) == g["distance_bits"]).all()
    assert (prim == g["prim"]).all()
    assert (bits(r["pos"][hit]) == bits(g["pos"][hit])).all()


@pytest.mark.parametrize("name,use_bvh", [("bvh", True), ("list", False)])
def test_instanced_scene_hits(oracle, name, use_bvh):
    """Aggregate over BVH<Instance> / List<Instance>, Instance::hit, Trace::transform (instance.h:15-40, trace.h:38-43)."""
    g = load(f"scene_hits_{name}.npz")
    sb = scenes.instanced_scene(use_bvh=use_bvh, mesh_bvh=use_bvh)
    sc = oracle.scene(sb.desc())
    r = sc.trace_closest(scenes.scene_rays(len(g["hit"])), full=True)
    sc.close()
    hit = r["hits"]["prim"] != abi.PTX_NO_HIT
    assert (hit == (g["hit"] != 0)).all()
    assert (bits(r["hits"]["distance"])[hit] == g["distance_bits"][hit]).all()
    assert (bits(r["pos"][hit]) == bits(g["pos"][hit])).all()
    assert (r["material"][hit] == g["material"][hit]).all()
    # normals and uv of analytic spheres pass through atan2/acos: value comparison
    np.testing.assert_allclose(r["normal"][hit], g["normal"][hit], rtol=0, atol=1e-6)
    np.testing.assert_allclose(r["uv"][hit], g["uv"][hit], rtol=0, atol=1e-6)


def test_builder_matches_reference_builder(oracle):
    """BVH::build (bvh.cpp:31-243): node arrays of the checker's builder == the reference's, byte for byte."""
    import hashlib
    g = load("builder.npz")
    f = load("hits_fuzz462.npz")
    s = load("hits_sphere3.npz")
    verts_f = positions_to_verts(f["positions"])
    cases = {
        "fuzz462": (verts_f, np.arange(len(verts_f), dtype=np.uint32).reshape(-1, 3)),
        "sphere3": (s["verts"], s["indices"]),
        "soup20k": synth.soup(20000),
    }
    for name, (verts, idx) in cases.items():
        sb = abi.SceneBuilder()
        sc = oracle.mesh_scene(sb.mesh_desc(verts, idx, use_bvh=True, max_leaf_size=4))
        nodes = sc.mesh_nodes()
        sc.close()
        assert len(nodes) == int(g[f"{name}_n_nodes"])
        assert (nodes[:64].tobytes() == g[f"{name}_nodes_head"].tobytes())
        assert hashlib.sha256(nodes.tobytes()).digest() == g[f"{name}_nodes_sha256"].tobytes()
        # ... and the product's host builder produces the same array
        b = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4)
        assert b.nodes().tobytes() == nodes.tobytes()
        b.close()


def test_probes(oracle):
    p = load("probes.npz")
    # BBox::hit (lib/bbox.h:81-154) incl. zero direction components and origins on faces
    rays = rays_of(p["bbox_rays"])
    for i in range(len(rays)):
        h, t = oracle.bbox_hit(p["bbox_boxes"][i], rays[i:i + 1])
        assert int(h) == p["bbox_hit"][i], i
        assert (bits(t) == p["bbox_times_bits"][i]).all(), i
    # Triangle::hit / bbox (tri_mesh.cpp:10-111)
    rays = rays_of(p["tri_rays"])
    n_hit = 0
    for i in range(len(rays)):
        h, o = oracle.triangle_hit(p["tri_verts"][i], rays[i:i + 1])
        assert int(h) == p["tri_hit"][i], i
        if h:
            n_hit += 1
            assert (bits(o[:9]) == p["tri_out_bits"][i][:9]).all(), i
        assert (bits(oracle.triangle_bbox(p["tri_verts"][i])) == p["tri_bbox_bits"][i]).all()
    assert n_hit > 20
    # Shapes::Sphere::hit (scene/shape.cpp:21-85)
    rays = rays_of(p["sphere_rays"])
    for i in range(len(rays)):
        h, o = oracle.sphere_hit(float(p["sphere_radius"][i]), rays[i:i + 1])
        assert int(h) == p["sphere_hit"][i], i
        if h:
            assert (bits(o[:7]) == bits(p["sphere_out"][i][:7])).all(), i
            np.testing.assert_allclose(o[7:], p["sphere_out"][i][7:], atol=1e-6)
    # Mat4::inverse, rotate_to (lib/mat4.h:325-393)
    for m, e in zip(p["mat4_in"], p["mat4_inverse_bits"]):
        assert (bits(oracle.mat4_inverse(m)) == e).all()
    for d, e in zip(p["rotate_to_in"], p["rotate_to_bits"]):
        assert (bits(oracle.rotate_to(d)) == e).all()
    # RNG::unit (util/rand.cpp:17-21)
    for s, e in zip(p["rng_seed"], p["rng_units_bits"]):
        assert (bits(oracle.rng_units(int(s), 64)) == e).all()
    # Materials::*::scatter with the g++ behaviour of schlick (material.cpp:58-71)
    ca, cb = (0.9, 0.6, 0.3), (0.25, 0.5, 0.75)
    for name, mt, ior in (("lambertian", abi.PTX_MAT_LAMBERTIAN, 1.5), ("mirror", abi.PTX_MAT_MIRROR, 1.5), ("refract", abi.PTX_MAT_REFRACT, 1.33),
                          ("glass", abi.PTX_MAT_GLASS, 1.5), ("glass_ior1", abi.PTX_MAT_GLASS, 1.0)):
        got = oracle.material_scatter(mt, ca, cb, ior, abi.PTX_FRESNEL_GCC_INT_ABS, 462, p["scatter_out_dirs"])
        np.testing.assert_allclose(got, p[f"scatter_{name}"], rtol=0, atol=2e-6, err_msg=name)
    # Environment_Lights::Sphere::evaluate over an image (env_light.cpp:25-31, texture.cpp:9-94)
    for sampler, key in ((abi.PTX_SAMPLER_BILINEAR, "env_bilinear"), (abi.PTX_SAMPLER_NEAREST, "env_nearest")):
        sb = abi.SceneBuilder()
        sb.add_env_light(abi.PTX_ENV_SPHERE, sb.add_image_texture(p["env_image"], sampler))
        sc = oracle.scene(sb.desc())
        got = np.stack([sc.env_evaluate(0, d) for d in p["env_dirs"]])
        sc.close()
        np.testing.assert_allclose(got, p[key], rtol=0, atol=2e-6)


def sphere5_mesh():
    g = load("hits_sphere5.npz")
    return positions_to_verts(g["positions"]), g["indices"].astype(np.uint32), g


def test_cow_and_sphere5_hits(oracle):
    """The cow of media/js3d/cow.js3d (shared vertices, thin triangles; half of the rays graze its surface) and
    Util::sphere_mesh(1, 5), 20,480 triangles: Tri_Mesh::hit of the reference through its own builder."""
    g = load("hits_cow.npz")
    verts, idx = scenes.cow_mesh()
    check_mesh_hits(oracle, verts, idx, rays_of(g["rays"]), g)
    verts, idx, g = sphere5_mesh()
    check_mesh_hits(oracle, verts, idx, rays_of(g["rays"]), g, check_attrs=False)


def test_builder_matches_reference_builder_cow_sphere5(oracle):
    import hashlib
    g = load("builder2.npz")
    cases = {"cow": scenes.cow_mesh(), "sphere5": sphere5_mesh()[:2]}
    for name, (verts, idx) in cases.items():
        sb = abi.SceneBuilder()
        sc = oracle.mesh_scene(sb.mesh_desc(verts, idx, use_bvh=True, max_leaf_size=4))
        nodes = sc.mesh_nodes()
        sc.close()
        assert len(nodes) == int(g[f"{name}_n_nodes"])
        assert nodes[:64].tobytes() == g[f"{name}_nodes_head"].tobytes()
        assert hashlib.sha256(nodes.tobytes()).digest() == g[f"{name}_nodes_sha256"].tobytes()
        b = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4)
        assert b.nodes().tobytes() == nodes.tobytes()
        b.close()


def test_probes2_tonemap(oracle):
    """HDR_Image::tonemap_to (util/hdr_image.cpp:201-225): the checker and the product's ptx_tonemap give the reference's bytes."""
    p = load("probes2.npz")
    for e, want in zip(p["tonemap_exposure"], p["tonemap_rgba"]):
        assert (oracle.tonemap(p["tonemap_rgb"], float(e)) == want).all()
        assert (capi.tonemap(p["tonemap_rgb"], float(e)) == want).all()


def test_probes2_delta_lights(oracle):
    """Delta_Lights::{Point,Directional,Spot}::incoming through PT::Light_Instance (delta_light.cpp:7-40, instance.h:74-91)."""
    p = load("probes2.npz")
    c = p["light_params"]
    for t, name in ((abi.PTX_LIGHT_POINT, "point"), (abi.PTX_LIGHT_DIRECTIONAL, "directional"), (abi.PTX_LIGHT_SPOT, "spot")):
        for T, key in ((None, "identity"), (p["light_T"], "transformed")):
            got = oracle.delta_incoming(t, c[:3], float(c[3]), float(c[4]), float(c[5]), T, p["light_points"])
            want = p[f"light_{name}_{key}_bits"]
            if t == abi.PTX_LIGHT_SPOT:   # atan2 in the falloff: value comparison for the radiance, bits for the geometry
                np.testing.assert_allclose(got[:, :3], want.view(np.float32)[:, :3], rtol=0, atol=2e-6)
                assert (bits(got[:, 3:]) == want[:, 3:]).all()
            else:
                assert (bits(got) == want).all(), (name, key)


def test_probes2_samplers(oracle):
    """Samplers::Rect (samplers.cpp:7-19, including which of its two draws scales x under g++) and
    Hemisphere::Cosine (samplers.cpp:67-83); Camera::sample_ray's pixel offset is that Rect sample."""
    p = load("probes2.npz")
    assert (bits(oracle.sampler_rect(4660, 1.0, 1.0, 256)) == p["rect_unit_bits"]).all()
    assert (bits(oracle.sampler_rect(7, 3.0, 0.5, 256)) == p["rect_3x05_bits"]).all()
    np.testing.assert_allclose(oracle.sampler_cosine(462, 1024), p["cosine_bits"].view(np.float32), rtol=0, atol=2e-6)
    # the checker's Camera::sample_ray in the reference's draw order == the ray through the reference's Rect sample
    cam = abi.PtxCamera()
    cam.vertical_fov, cam.aspect_ratio, cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth = 50.0, 1.5, 30, 20, 1, 4
    T = scenes.trs((1, 2, 3), axis=(0, 1, 0), deg=25)
    for i in range(16):
        cam.camera_to_world[i] = float(T[i])
    offs = p["rect_unit_bits"].view(np.float32)
    rays = oracle.camera_sample_rays(cam, 4660, 7, 11, 64)
    for k in range(64):
        want = oracle.camera_ray(cam, 7, 11, float(offs[k, 0]), float(offs[k, 1]))
        assert rays[k:k + 1].tobytes() == want.tobytes()


PHILOX_KAT = [   # Random123's published known-answer vectors for philox4x32, 10 rounds: counter, key, expected
    ([0x00000000] * 4, [0x00000000] * 2, [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


def test_philox_known_answers(oracle):
    for c, k, want in PHILOX_KAT:
        assert oracle.philox_block(c, k).tolist() == want
    # the path stream: draw i of (seed, pixel, sample) is word i % 4 of the block with counter (i / 4, pixel, sample, 0)
    seed, pixel, sample = 0x0123456789ABCDEF, 4242, 17
    units = oracle.philox_units(seed, pixel, sample, 12)
    for blk in range(3):
        words = oracle.philox_block([blk, pixel, sample, 0], [seed & 0xFFFFFFFF, seed >> 32])
        assert (units[4 * blk:4 * blk + 4] == np.ldexp(words.astype(np.float32), -32)).all()

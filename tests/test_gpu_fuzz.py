"""Randomised differential test: seeded random two-level scenes (meshes of several kinds, analytic spheres, transforms
with rotation, non-uniform and mirrored scale, coincident instances, BVH and List on either level) and random rays,
on the GPU through the C ABI in both traversal orders, against the CPU checker -- and against the reference's own
Aggregate / Instance / Tri_Mesh / Sphere (aggregate.h:41, instance.h:32-40) where oracle/_ref/libptref.so is present.
Bar: hit flag, distance bits, instance and primitive equal; bit-exact."""
import numpy as np
import pytest

import scenes
from conftest import bits
from pathtracer_amd import abi, capi, synth

pytestmark = pytest.mark.gpu

MODES = [abi.PTX_TRAVERSE_REFERENCE, abi.PTX_TRAVERSE_CULLED, abi.PTX_TRAVERSE_CULLED | abi.PTX_NO_PERSISTENT]


def random_scene(seed, max_leaf=None):
    """max_leaf: leaf size of the soup mesh instead of a drawn one (the reference's Tri_Mesh always builds with 4)."""
    g = np.random.Generator(np.random.Philox(key=1000 + seed))
    use_bvh = seed % 5 != 0          # every fifth scene: List<Instance>
    mesh_bvh = seed % 7 != 0         # every seventh: List<Triangle>
    sb = abi.SceneBuilder(use_bvh=use_bvh)
    mat = sb.add_material(abi.PTX_MAT_LAMBERTIAN, sb.add_constant_texture((0.5, 0.5, 0.5)))
    meshes = [
        sb.add_mesh(*scenes.quad(), use_bvh=mesh_bvh),
        sb.add_mesh(*scenes.icosphere(int(g.integers(0, 3))), use_bvh=mesh_bvh),
        sb.add_mesh(*synth.soup(int(g.integers(20, 400)), key=seed, extent=1.0, half=float(g.random() * 0.2 + 0.02)), use_bvh=mesh_bvh, max_leaf_size=int(g.integers(1, 9)) if max_leaf is None else (int(g.integers(1, 9)), max_leaf)[1]),
    ]
    n_inst = int(g.integers(1, 40))
    transforms = []
    for k in range(n_inst):
        pos = g.random(3) * 4 - 2
        axis = g.standard_normal(3)
        scale = g.random(3) * 1.5 + 0.1
        if g.random() < 0.25:
            scale[int(g.integers(0, 3))] *= -1.0          # mirrored instance
        if g.random() < 0.15:
            scale[:] = scale[0]                            # uniform scale
        T = scenes.trs(pos, axis=axis, deg=float(g.random() * 360), scale=scale)
        if g.random() < 0.1:
            T = abi.IDENTITY16                             # no transform at all
        if transforms and g.random() < 0.15:
            T = transforms[int(g.integers(0, len(transforms)))]   # coincident with an earlier instance: distance ties
        transforms.append(T)
        if g.random() < 0.25:
            sb.add_instance(-1, mat, T, sphere_radius=float(g.random() * 0.8 + 0.1))
        else:
            sb.add_instance(meshes[int(g.integers(0, len(meshes)))], mat, T)
    return sb


def random_rays(seed, n):
    g = np.random.Generator(np.random.Philox(key=5000 + seed))
    r = np.zeros(n, dtype=abi.RAY_DTYPE)
    r["o"] = (g.random((n, 3)) * 7 - 3.5).astype(np.float32)
    target = g.random((n, 3)) * 4 - 2
    d = target - r["o"]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r["d"] = synth._unit_f32(d)
    r["tmin"], r["tmax"] = 0.0, np.inf
    k = n // 4     # a quarter with finite bounds, some axis-parallel, some starting far away
    r["tmin"][:k] = 1e-3
    r["tmax"][:k] = (g.random(k) * 6).astype(np.float32)
    r["d"][k:k + 64] = np.eye(3, dtype=np.float32)[np.arange(64) % 3] * np.where(np.arange(64) % 2, 1, -1)[:, None].astype(np.float32)
    r["o"][k + 64:k + 128] *= np.float32(50.0)
    return r


@pytest.mark.parametrize("seed", range(24))
def test_random_scene_against_checker(oracle, seed):
    sb = random_scene(seed)
    desc = sb.desc()
    rays = random_rays(seed, 1 << 13)
    osc = oracle.scene(desc)
    want = osc.trace_closest(rays)
    shadow = rays.copy()
    shadow["tmin"], shadow["tmax"] = 1e-4, np.float32(2.5)
    want_any = osc.trace_any(shadow)
    osc.close()
    sc = capi.Scene(desc, keep=sb)
    for mode in MODES:
        got = sc.trace_closest(rays, flags=mode)
        hit = want["hits"]["prim"] != abi.PTX_NO_HIT
        assert np.array_equal(got["hits"]["prim"] != abi.PTX_NO_HIT, hit), f"seed {seed} mode {mode}: hit flags differ"
        assert np.array_equal(bits(got["hits"]["distance"]), bits(want["hits"]["distance"])), f"seed {seed} mode {mode}: distance bits differ"
        assert np.array_equal(got["hits"]["prim"], want["hits"]["prim"]) and np.array_equal(got["inst"][hit], want["inst"][hit]), f"seed {seed} mode {mode}"
        assert np.array_equal(bits(got["bary"][hit]), bits(want["bary"][hit]))
        assert np.array_equal(sc.trace_any(shadow, flags=mode), want_any), f"seed {seed} mode {mode}: occlusion differs"
    sc.close()
    assert 0.02 < hit.mean() < 1.0


@pytest.mark.parametrize("seed", range(0, 24, 3))
def test_random_scene_against_the_reference_itself(ref, seed):
    """The same scenes through the reference's own classes: hit flag and distance bits (its Trace carries no primitive id).
    Leaf size 4 as in the reference's Tri_Mesh (tri_mesh.cpp:173): trees of different leaf sizes answer alike only while
    every primitive lies inside its leaf's box, which the reference's builder does not guarantee."""
    sb = random_scene(seed, max_leaf=4)
    desc = sb.desc()
    rays = random_rays(seed, 1 << 13)
    rsc = ref.scene(desc)
    want = rsc.hit(rays)
    rsc.close()
    sc = capi.Scene(desc, keep=sb)
    for mode in MODES:
        got = sc.trace_closest(rays, flags=mode)["hits"]
        hit = want["hit"] != 0
        assert np.array_equal(got["prim"] != abi.PTX_NO_HIT, hit), f"seed {seed} mode {mode}: hit flags differ"
        assert np.array_equal(bits(got["distance"])[hit], bits(want["distance"])[hit]), f"seed {seed} mode {mode}: distance bits differ"
    sc.close()

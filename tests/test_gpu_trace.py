"""GPU parity: closest-hit and occlusion queries through the C ABI against the CPU checker.

Bit-exact bar: Trace::distance bits, primitive position in the BVH-ordered array, instance
position, and the barycentrics of the winning triangle.
"""
import numpy as np
import pytest

import scenes
from conftest import bits
from pathtracer_amd import abi, capi, synth

pytestmark = pytest.mark.gpu

MODES = [abi.PTX_TRAVERSE_REFERENCE, abi.PTX_TRAVERSE_CULLED,
         abi.PTX_TRAVERSE_REFERENCE | abi.PTX_NO_PERSISTENT, abi.PTX_TRAVERSE_CULLED | abi.PTX_NO_PERSISTENT,
         abi.PTX_TRAVERSE_CULLED | abi.PTX_RAYS_COHERENT]


def assert_same_hits(got, want, what):
    gh, wh = got["hits"], want["hits"]
    bad = np.flatnonzero((bits(gh["distance"]) != bits(wh["distance"])) | (gh["prim"] != wh["prim"]))
    assert len(bad) == 0, f"{what}: {len(bad)} of {len(gh)} rays differ, first {bad[:5]}: got {gh[bad[:5]]} want {wh[bad[:5]]}"
    hit = wh["prim"] != abi.PTX_NO_HIT
    assert np.array_equal(bits(got["bary"][hit]), bits(want["bary"][hit])), f"{what}: barycentrics differ"
    if got.get("inst") is not None and want.get("inst") is not None:
        assert np.array_equal(got["inst"][hit], want["inst"][hit]), f"{what}: instance ids differ"


MESHES = {
    "soup_small_tris": lambda: synth.soup(20000),
    "soup_big_tris": lambda: synth.soup(5000, key=5, half=0.5),
    "icosphere": lambda: scenes.icosphere(4),
    "quad_flat": lambda: scenes.quad(),
    "one_triangle": lambda: (np.array([[0, 0, 0, 1, 0, 0, 0, 0], [1, 0, 0, 1, 0, 0, 0, 0], [0, 1, 0, 1, 0, 0, 0, 0]], dtype=np.float32),
                             np.array([[0, 1, 2]], dtype=np.uint32)),
    "empty": lambda: (np.zeros((0, 8), dtype=np.float32), np.zeros((0, 3), dtype=np.uint32)),
}


@pytest.mark.parametrize("name", list(MESHES))
@pytest.mark.parametrize("use_bvh", [True, False])
def test_mesh_closest_hit_matches_checker(oracle, name, use_bvh):
    verts, idx = MESHES[name]()
    if not use_bvh and len(idx) > 6000:
        pytest.skip("List<Triangle> on a large mesh: minutes of brute force for no extra coverage")
    sb = abi.SceneBuilder()
    md = sb.mesh_desc(verts, idx, use_bvh=use_bvh)
    want_scene = oracle.mesh_scene(md)
    scene = capi.Scene(mesh_desc=md, keep=sb)
    assert np.array_equal(scene.mesh_order(len(idx)), want_scene.mesh_order(len(idx)))
    lo = verts[:, :3].min(0) if len(verts) else np.zeros(3)
    hi = verts[:, :3].max(0) if len(verts) else np.ones(3)
    rays = synth.rays(1 << 17, key=31)   # large enough for the library's ray-grouping pass to run
    rays["o"] = lo + (rays["o"] / 10.0) * (hi - lo) * 1.2 - 0.1 * (hi - lo)
    # axis-aligned directions and origins on box planes: the 0/0 and x/0 paths of the slab test
    rays["d"][:64] = np.eye(3, dtype=np.float32)[np.arange(64) % 3]
    rays["o"][:32] = np.where(np.arange(32)[:, None] % 2 == 0, lo, hi).astype(np.float32)
    want = want_scene.trace_closest(rays)
    for mode in MODES:
        got = scene.trace_closest(rays, flags=mode)
        assert_same_hits(got, want, f"{name} bvh={use_bvh} mode={mode}")
    scene.close()


@pytest.mark.parametrize("name", ["soup_small_tris", "soup_big_tris", "icosphere"])
def test_mesh_occlusion_matches_checker(oracle, name):
    verts, idx = MESHES[name]()
    sb = abi.SceneBuilder()
    md = sb.mesh_desc(verts, idx)
    want_scene = oracle.mesh_scene(md)
    scene = capi.Scene(mesh_desc=md, keep=sb)
    rays = synth.shadow_rays(1 << 14, key=77)
    lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
    rays["o"] = lo + (rays["o"] / 10.0) * (hi - lo)
    rays["tmax"] *= np.float32(np.linalg.norm(hi - lo) / 17.0)
    want = want_scene.trace_any(rays)
    assert 0 < want.sum() < len(want)
    for mode in MODES:
        got = scene.trace_any(rays, flags=mode)
        assert np.array_equal(got, want), f"{name} mode={mode}: {np.count_nonzero(got != want)} differ"
    scene.close()


@pytest.mark.parametrize("use_bvh,mesh_bvh", [(True, True), (False, True), (True, False)])
def test_instanced_scene_matches_checker(oracle, use_bvh, mesh_bvh):
    sb = scenes.instanced_scene(use_bvh=use_bvh, mesh_bvh=mesh_bvh)
    desc = sb.desc()
    want_scene = oracle.scene(desc)
    scene = capi.Scene(desc=desc, keep=sb)
    n_inst = len(sb.instances)
    assert np.array_equal(scene.instance_order(n_inst), want_scene.instance_order(n_inst))
    rays = np.concatenate([scenes.scene_rays(1 << 13, seed=3, inside=True), scenes.scene_rays(1 << 12, seed=4, inside=False)])
    # axis-parallel directions (zero components) and origins on the room's planes leave the range where the
    # short division form is proven exact: such rays are finished by the plain-division kernels
    rays["d"][::9] = np.array([[0, -1, 0], [1, 0, 0], [0, 0, -1]], dtype=np.float32)[np.arange(len(rays[::9])) % 3]
    rays["o"][::27, 1] = -1.0
    flat = np.arange(5, len(rays), 31)
    flat = flat[flat % 9 != 0]            # not the axis-parallel ones: a zero vector is no direction
    d = rays["d"][flat].astype(np.float64)
    d[:, 2] = 0.0
    rays["d"][flat] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    want = want_scene.trace_closest(rays)
    assert (want["hits"]["prim"] != abi.PTX_NO_HIT).mean() > 0.5
    for mode in MODES:
        got = scene.trace_closest(rays, flags=mode)
        assert_same_hits(got, want, f"instanced top_bvh={use_bvh} mesh_bvh={mesh_bvh} mode={mode}")
    shadow = rays.copy()
    shadow["tmin"], shadow["tmax"] = 1e-5, 0.8
    want_any = want_scene.trace_any(shadow)
    for mode in MODES:
        assert np.array_equal(scene.trace_any(shadow, flags=mode), want_any)
    scene.close()


def test_injected_tree_is_traversed_as_given(oracle):
    """A tree built elsewhere (here: by the host builder) handed in through PtxMeshDesc.nodes."""
    verts, idx = synth.soup(3000, key=12, half=0.2)
    b = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4)
    order, nodes = b.order(), b.nodes()
    sb = abi.SceneBuilder()
    md = sb.mesh_desc(verts, idx[order], nodes=nodes)
    scene = capi.Scene(mesh_desc=md, keep=sb)
    want_scene = oracle.mesh_scene(sb.mesh_desc(verts, idx))
    rays = synth.rays(4096, key=8)
    assert_same_hits(scene.trace_closest(rays), want_scene.trace_closest(rays), "injected")
    scene.close()


def test_errors_are_reported_not_fatal():
    sb = abi.SceneBuilder()
    verts, idx = scenes.quad()
    bad = idx.copy()
    bad[0, 0] = 99
    with pytest.raises(capi.PtxError, match="index out of range"):
        capi.Scene(mesh_desc=sb.mesh_desc(verts, bad), keep=sb)
    with pytest.raises(capi.PtxError, match="device"):
        capi.Scene(mesh_desc=sb.mesh_desc(verts, idx), device=4096, keep=sb)
    with pytest.raises(capi.PtxError, match="max_leaf_size"):
        capi.Scene(mesh_desc=sb.mesh_desc(verts, idx, max_leaf_size=64), keep=sb)


def test_many_instances_bit_exact(oracle):
    """216 rotated, scaled instances of one mesh plus spheres: a deep instance-level tree (BVH<Instance>, max leaf 1).
    Closest hits in both orders equal the CPU checker's bit for bit."""
    sb = abi.SceneBuilder()
    ico = sb.add_mesh(*scenes.icosphere(2))
    m = sb.add_material(abi.PTX_MAT_LAMBERTIAN, sb.add_constant_texture((0.5, 0.5, 0.5)))
    for ix in range(6):
        for iy in range(6):
            for iz in range(6):
                k = ix + 6 * iy + 36 * iz
                if k % 7 == 3:
                    sb.add_instance(-1, m, scenes.trs((ix * 1.1, iy * 1.1, iz * 1.1), scale=(0.4, 0.4, 0.4)), sphere_radius=1.0)
                else:
                    sb.add_instance(ico, m, scenes.trs((ix * 1.1, iy * 1.1, iz * 1.1), axis=(1, 2, 3), deg=13.0 * k, scale=(0.45, 0.35 + 0.01 * ix, 0.5)))
    desc = sb.desc()
    g = np.random.Generator(np.random.Philox(key=77))
    n = 1 << 16
    rays = np.zeros(n, dtype=abi.RAY_DTYPE)
    rays["o"] = (g.random((n, 3), dtype=np.float32) * 9 - 1.5).astype(np.float32)
    d = g.standard_normal((n, 3))
    rays["d"] = synth._unit_f32(d / np.linalg.norm(d, axis=1, keepdims=True))
    rays["tmin"], rays["tmax"] = 0.0, np.inf
    osc = oracle.scene(desc)
    want = osc.trace_closest(rays)
    osc.close()
    sc = capi.Scene(desc, keep=sb)
    for flags in (abi.PTX_TRAVERSE_REFERENCE, abi.PTX_TRAVERSE_CULLED, abi.PTX_TRAVERSE_CULLED | abi.PTX_NO_PERSISTENT):
        got = sc.trace_closest(rays, flags=flags)
        assert (bits(got["hits"]["distance"]) == bits(want["hits"]["distance"])).all()
        assert (got["hits"]["prim"] == want["hits"]["prim"]).all()
        assert (got["inst"] == want["inst"]).all()
    sc.close()
    assert (want["hits"]["prim"] != abi.PTX_NO_HIT).mean() > 0.3


def test_wide_tree_switch_is_bit_identical():
    """PTX_WIDE_TREE=1 (an experiment switch: the four-wide form of the tree in the replenishing kernel) gives the
    reference order's answers on every ray.  Read at library load, so it runs in a process of its own."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PTX_WIDE_TREE="1")
    out = subprocess.check_output([sys.executable, os.path.join(root, "tools", "time_closest.py"), str(1 << 20), "2"], env=env, text=True, timeout=300)
    assert "mismatches vs reference order 0" in out, out

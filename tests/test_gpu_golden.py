"""The HIP path against the committed vectors the reference's own code produced
(tests/golden/*.npz, written by tests/golden/make_golden.py).  Bit-exact: distance bits and
primitive ids, in the reference-order traversal and in the culled traversal."""
import os

import numpy as np
import pytest

from pathtracer_amd import abi, capi, synth
import scenes
from conftest import bits
from test_oracle_golden import load, positions_to_verts, rays_of

pytestmark = pytest.mark.gpu

MODES = [abi.PTX_TRAVERSE_REFERENCE, abi.PTX_TRAVERSE_CULLED, abi.PTX_TRAVERSE_CULLED | abi.PTX_NO_PERSISTENT,
         abi.PTX_TRAVERSE_CULLED | abi.PTX_RAYS_COHERENT]


def gpu_mesh_hits(verts, idx, rays, flags, use_bvh=True, nodes=None):
    sc = capi.Scene.from_mesh(verts, idx, use_bvh=use_bvh, max_leaf_size=4, nodes=nodes)
    hits = sc.trace_closest(rays, flags=flags)["hits"]
    order = sc.mesh_order(len(idx)) if nodes is None else np.arange(len(idx), dtype=np.uint32)
    sc.close()
    prim = hits["prim"]
    hit = prim != abi.PTX_NO_HIT
    return hits["distance"], np.where(hit, order[np.minimum(prim, len(order) - 1)], abi.PTX_NO_HIT), hit


@pytest.mark.parametrize("flags", MODES)
def test_fuzz_mesh(flags):
    g = load("hits_fuzz462.npz")
    verts = positions_to_verts(g["positions"])
    idx = np.arange(len(verts), dtype=np.uint32).reshape(-1, 3)
    dist, prim, hit = gpu_mesh_hits(verts, idx, rays_of(g["rays"]), flags)
    assert (bits(dist) == g["distance_bits"]).all()
    assert (prim == g["prim"]).all()


@pytest.mark.parametrize("flags", MODES)
def test_sphere_mesh(flags):
    g = load("hits_sphere3.npz")
    dist, prim, hit = gpu_mesh_hits(g["verts"], g["indices"], rays_of(g["rays"]), flags)
    assert (bits(dist) == g["distance_bits"]).all()
    assert (prim == g["prim"]).all()
    gl = load("hits_sphere3_list.npz")
    dist, prim, hit = gpu_mesh_hits(g["verts"], g["indices"], rays_of(g["rays"]), flags, use_bvh=False)
    assert (bits(dist) == gl["distance_bits"]).all()
    assert (prim == gl["prim"]).all()


@pytest.mark.parametrize("flags", MODES)
def test_injected_soup(flags):
    g = load("hits_soup20k.npz")
    verts, idx = synth.soup(20000)
    bvh = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4)
    nodes, order = bvh.nodes(), bvh.order()
    dist, prim, hit = gpu_mesh_hits(verts, idx[order], synth.rays(len(g["prim"])), flags, nodes=nodes)
    prim = np.where(hit, order[np.minimum(prim, len(order) - 1)], abi.PTX_NO_HIT)
    assert (bits(dist) == g["distance_bits"]).all()
    assert (prim == g["prim"]).all()


@pytest.mark.parametrize("name,use_bvh", [("bvh", True), ("list", False)])
@pytest.mark.parametrize("flags", [abi.PTX_TRAVERSE_REFERENCE, abi.PTX_TRAVERSE_CULLED])
def test_instanced_scene(name, use_bvh, flags):
    g = load(f"scene_hits_{name}.npz")
    sb = scenes.instanced_scene(use_bvh=use_bvh, mesh_bvh=use_bvh)
    sc = capi.Scene(sb.desc(), keep=sb)
    hits = sc.trace_closest(scenes.scene_rays(len(g["hit"])), flags=flags)["hits"]
    sc.close()
    hit = hits["prim"] != abi.PTX_NO_HIT
    assert (hit == (g["hit"] != 0)).all()
    assert (bits(hits["distance"])[hit] == g["distance_bits"][hit]).all()


@pytest.mark.parametrize("flags", MODES)
def test_cow_mesh_with_grazing_rays(flags):
    """The reference's cow (shared vertices, thin triangles), half of the rays skimming its surface: where a distance
    computed for a near-parallel ray/triangle pair is least accurate, which is what the culled order's guard band is for."""
    g = load("hits_cow.npz")
    verts, idx = scenes.cow_mesh()
    dist, prim, hit = gpu_mesh_hits(verts, idx, rays_of(g["rays"]), flags)
    assert (bits(dist) == g["distance_bits"]).all()
    assert (prim == g["prim"]).all()


@pytest.mark.parametrize("flags", MODES)
def test_sphere5_mesh(flags):
    from test_oracle_golden import sphere5_mesh
    verts, idx, g = sphere5_mesh()
    dist, prim, hit = gpu_mesh_hits(verts, idx, rays_of(g["rays"]), flags)
    assert (bits(dist) == g["distance_bits"]).all()
    assert (prim == g["prim"]).all()


def test_culled_equals_reference_order_on_grazing_rays():
    """262,144 rays skimming the cow and the level-5 sphere (no fixture: the two traversal orders against each other)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden_rays import grazing_rays
    from test_oracle_golden import sphere5_mesh
    for verts, idx in (scenes.cow_mesh(), sphere5_mesh()[:2]):
        rays = grazing_rays(verts, idx, 1 << 18, 77)
        sc = capi.Scene.from_mesh(verts, idx)
        a = sc.trace_closest(rays, flags=abi.PTX_TRAVERSE_REFERENCE)
        for flags in (abi.PTX_TRAVERSE_CULLED, abi.PTX_TRAVERSE_CULLED | abi.PTX_NO_PERSISTENT):
            b = sc.trace_closest(rays, flags=flags)
            assert (bits(a["hits"]["distance"]) == bits(b["hits"]["distance"])).all()
            assert (a["hits"]["prim"] == b["hits"]["prim"]).all()
        sc.close()
        assert (a["hits"]["prim"] != abi.PTX_NO_HIT).mean() > 0.5


def test_philox_known_answers_on_device(oracle):
    """The device's Philox-4x32-10 against Random123's published vectors, and the path stream against the checker's."""
    from test_oracle_golden import PHILOX_KAT
    got = capi.selftest_philox([c for c, _, _ in PHILOX_KAT], [k for _, k, _ in PHILOX_KAT])
    assert got.tolist() == [w for _, _, w in PHILOX_KAT]
    g = np.random.Generator(np.random.Philox(key=3))
    counters = g.integers(0, 1 << 32, (4096, 4), dtype=np.uint64).astype(np.uint32)
    keys = g.integers(0, 1 << 32, (4096, 2), dtype=np.uint64).astype(np.uint32)
    got = capi.selftest_philox(counters, keys)
    want = np.stack([oracle.philox_block(c, k) for c, k in zip(counters[:256], keys[:256])])
    assert (got[:256] == want).all()
    for seed, pixel, sample in ((0xABCDEF12345, 0, 0), (0xFFFFFFFFFFFFFFFF, 123456, 1023), (7, 8294399, 5)):
        assert (bits(capi.selftest_path_units(seed, pixel, sample, 41)) == bits(oracle.philox_units(seed, pixel, sample, 41))).all()


def test_soup_against_the_reference_itself(ref):
    """65,536 rays of the benchmark's ray batch against the benchmark's 1M-triangle soup: the reference's own
    BVH<Triangle>::hit (the builder's tree injected through the public nodes / primitives members, bvh.h:53-55) against
    the GPU, distance bits and primitive.  Runs where oracle/_ref/libptref.so is present; skipped elsewhere."""
    verts, idx = synth.soup(1_000_000)
    bvh = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4)
    nodes, order = bvh.nodes(), bvh.order()
    rays = synth.rays(1 << 16, key=synth.RAY_KEY)
    rm = ref.mesh(verts, idx[order], id_coded=True, nodes=nodes)
    want = rm.hit(rays, nthreads=16)["hits"]
    rm.close()
    sc = capi.Scene.from_mesh(verts, idx[order], nodes=nodes)
    for flags in (abi.PTX_TRAVERSE_REFERENCE, abi.PTX_TRAVERSE_CULLED):
        got = sc.trace_closest(rays, flags=flags)["hits"]
        assert (bits(got["distance"]) == bits(want["distance"])).all()
        assert (got["prim"] == want["prim"]).all()
    sc.close()
    assert (want["prim"] != abi.PTX_NO_HIT).mean() > 0.5

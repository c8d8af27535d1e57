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

"""The GPU builder (ptx_bvh_build_gpu: BVH::build / buildRecursive, bvh.cpp:31-243, a tree level at a time) against the
host builder, whose node arrays equal the reference builder's byte for byte (test_oracle_golden.py): node array,
primitive order, statistics and depth must be identical."""
import numpy as np
import pytest

import scenes
from pathtracer_amd import abi, capi, synth

pytestmark = pytest.mark.gpu


def same_tree(host, gpu):
    assert gpu.built_on_device
    hn, gn = host.nodes(), gpu.nodes()
    assert len(hn) == len(gn), (len(hn), len(gn))
    assert hn.tobytes() == gn.tobytes(), f"first differing node {np.flatnonzero(hn.view(np.uint8).reshape(len(hn), -1) != gn.view(np.uint8).reshape(len(gn), -1))[:1]}"
    assert np.array_equal(host.order(), gpu.order())
    assert np.array_equal(host.stats(), gpu.stats())
    assert host.max_depth == gpu.max_depth


def random_boxes(n, key, spread=10.0, size=0.05):
    g = np.random.Generator(np.random.Philox(key=key))
    c = (g.random((n, 3), dtype=np.float32) * np.float32(spread)).astype(np.float32)
    h = (g.random((n, 3), dtype=np.float32) * np.float32(size) + np.float32(1e-4)).astype(np.float32)
    return np.concatenate([c - h, c + h], axis=1).astype(np.float32)


MESHES = {
    "soup_20k": lambda: synth.soup(20000),
    "soup_big_tris": lambda: synth.soup(5000, key=5, half=0.5),
    "soup_300k": lambda: synth.soup(300_000, key=9),
    "icosphere4": lambda: scenes.icosphere(4),
    "cow": lambda: scenes.cow_mesh(),
    "quad": lambda: scenes.quad(),
}


@pytest.mark.parametrize("name", list(MESHES))
@pytest.mark.parametrize("max_leaf", [1, 4, 8])
def test_triangle_meshes(name, max_leaf):
    verts, idx = MESHES[name]()
    if np.signbit(verts[:, :3][verts[:, :3] == 0]).any():
        pytest.skip("the mesh holds a negative zero: built on the host by design")
    same_tree(capi.Bvh(verts=verts, indices=idx, max_leaf_size=max_leaf), capi.Bvh(verts=verts, indices=idx, max_leaf_size=max_leaf, gpu_device=0))


@pytest.mark.parametrize("n", [1, 2, 3, 5, 17, 255, 256, 257, 4096, 100_003])
def test_box_counts_around_block_sizes(n):
    boxes = random_boxes(n, key=n)
    same_tree(capi.Bvh(boxes=boxes, max_leaf_size=2), capi.Bvh(boxes=boxes, max_leaf_size=2, gpu_device=0))


def test_coincident_and_clustered_boxes():
    """Boxes that no split separates (the reference drops them, bvh.cpp:183-188), centres on bin boundaries, heavy
    clusters (predicate and bin disagree by rounding)."""
    g = np.random.Generator(np.random.Philox(key=3))
    same = np.tile(np.array([[1, 1, 1, 2, 2, 2]], dtype=np.float32), (37, 1))
    grid = np.stack(np.meshgrid(np.arange(16), np.arange(16), np.arange(8), indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float32) * np.float32(0.125)
    grid_boxes = np.concatenate([grid, grid + np.float32(0.125)], axis=1)
    cluster = random_boxes(50_000, key=4, spread=1e-3, size=1e-5) + np.float32(3.0)
    boxes = np.concatenate([same, grid_boxes, cluster, random_boxes(20_000, key=6)]).astype(np.float32)
    boxes = boxes[g.permutation(len(boxes))]
    host, gpu = capi.Bvh(boxes=boxes, max_leaf_size=4), capi.Bvh(boxes=boxes, max_leaf_size=4, gpu_device=0)
    same_tree(host, gpu)
    assert host.stats()[2] > 0      # some primitives were dropped, identically


def test_negative_zero_and_empty_inputs_go_to_the_host():
    boxes = random_boxes(1000, key=8)
    boxes[17, 1] = np.float32(-0.0)
    gpu = capi.Bvh(boxes=boxes, max_leaf_size=4, gpu_device=0)
    assert not gpu.built_on_device
    host = capi.Bvh(boxes=boxes, max_leaf_size=4)
    assert host.nodes().tobytes() == gpu.nodes().tobytes() and np.array_equal(host.order(), gpu.order())
    empty = capi.Bvh(boxes=np.zeros((0, 6), dtype=np.float32), max_leaf_size=4, gpu_device=0)
    assert not empty.built_on_device and len(empty.nodes()) == 1


def test_million_triangle_soup_and_traversal():
    """The benchmark's 1M-triangle soup: identical tree; injected into a scene it gives the hits of the scene that
    built its own tree."""
    verts, idx = synth.soup(1_000_000)
    host = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4)
    gpu = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4, gpu_device=0)
    same_tree(host, gpu)


def test_gpu_builder_against_the_reference_builder_fixtures():
    """Node arrays the reference's own BVH<Triangle>::build produced (tests/golden/builder.npz, builder2.npz: length,
    first 64 nodes, SHA-256 of the whole array) for its fuzz mesh, sphere meshes, a 20k soup and the cow."""
    import hashlib
    from test_oracle_golden import load, positions_to_verts, sphere5_mesh
    f, s = load("hits_fuzz462.npz"), load("hits_sphere3.npz")
    verts_f = positions_to_verts(f["positions"])
    cases = [
        (load("builder.npz"), "fuzz462", (verts_f, np.arange(len(verts_f), dtype=np.uint32).reshape(-1, 3))),
        (load("builder.npz"), "sphere3", (s["verts"], s["indices"])),
        (load("builder.npz"), "soup20k", synth.soup(20000)),
        (load("builder2.npz"), "cow", scenes.cow_mesh()),
        (load("builder2.npz"), "sphere5", sphere5_mesh()[:2]),
    ]
    on_device = 0
    for g, name, (verts, idx) in cases:
        b = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4, gpu_device=0)
        nodes = b.nodes()
        on_device += b.built_on_device
        assert len(nodes) == int(g[f"{name}_n_nodes"]), name
        assert nodes[:64].tobytes() == g[f"{name}_nodes_head"].tobytes(), name
        assert hashlib.sha256(nodes.tobytes()).digest() == g[f"{name}_nodes_sha256"].tobytes(), name
        b.close()
    assert on_device >= 3      # meshes with a negative zero among their coordinates are built on the host by design


def test_ten_million_boxes():
    """The size of the 10M+ configuration: several thousand scan tiles, bin tables of millions of nodes per level."""
    boxes = random_boxes(10_000_000, key=77, spread=100.0, size=0.05)
    host = capi.Bvh(boxes=boxes, max_leaf_size=4)
    gpu = capi.Bvh(boxes=boxes, max_leaf_size=4, gpu_device=0)
    same_tree(host, gpu)


def test_scene_creation_falls_back_to_the_host_builder(monkeypatch):
    """Scene creation sends meshes of 2^17 triangles and more to the GPU builder on its own initiative; when that
    attempt fails (transient device memory on a full GPU) the host builder, which gives the same tree byte for byte,
    takes over.  Only an explicit PTX_GPU_BUILD=1 makes the GPU builder's failure the call's failure."""
    verts, idx = synth.soup(150_000, key=31)
    rays = synth.rays(1 << 14, key=32)
    sc = capi.Scene.from_mesh(verts, idx)
    assert sc.mesh_built_on_device()
    want_nodes, want = sc.mesh_nodes(), sc.trace_closest(rays)["hits"]
    sc.close()
    monkeypatch.setenv("PTX_GPU_BUILD_FORCE_FAIL", "1")
    sc = capi.Scene.from_mesh(verts, idx)
    assert not sc.mesh_built_on_device()
    assert sc.mesh_nodes().tobytes() == want_nodes.tobytes()
    got = sc.trace_closest(rays)["hits"]
    assert got.tobytes() == want.tobytes()
    sc.close()
    monkeypatch.setenv("PTX_GPU_BUILD", "1")
    with pytest.raises(capi.PtxError):
        capi.Scene.from_mesh(verts, idx)

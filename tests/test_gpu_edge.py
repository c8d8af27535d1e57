"""GPU parity on the inputs that break traversals: empty and ragged batches, rays with NaN, infinite,
zero or huge components, flipped distance bounds, coincident and degenerate triangles, the smallest
and largest leaves the layout holds -- each against the CPU checker, bit for bit -- and, at a size
the checker cannot cover, properties that hold whatever the size is.
"""
import numpy as np
import pytest

import scenes
from conftest import bits
from pathtracer_amd import abi, capi, synth

pytestmark = pytest.mark.gpu

MODES = [abi.PTX_TRAVERSE_REFERENCE, abi.PTX_TRAVERSE_CULLED, abi.PTX_TRAVERSE_CULLED | abi.PTX_NO_PERSISTENT]


def same_floats(a, b):
    """Bit for bit, except that any NaN equals any NaN: which NaN an invalid operation returns (sign, payload)
    is the processor's choice -- x86 gives the negative default NaN, the GPU the positive one."""
    return (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))


def same(got, want, what):
    gh, wh = got["hits"], want["hits"]
    bad = np.flatnonzero(~same_floats(gh["distance"], wh["distance"]) | (gh["prim"] != wh["prim"]))
    assert len(bad) == 0, f"{what}: {len(bad)} of {len(gh)} differ, first {bad[:5]}: got {gh[bad[:5]]} want {wh[bad[:5]]}"
    hit = wh["prim"] != abi.PTX_NO_HIT
    assert same_floats(got["bary"][hit], want["bary"][hit]).all(), f"{what}: barycentrics differ"


def mesh_pair(oracle, verts, idx, **kw):
    sb = abi.SceneBuilder()
    md = sb.mesh_desc(verts, idx, **kw)
    return capi.Scene(mesh_desc=md, keep=sb), oracle.mesh_scene(md)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 4097, 65535, 65536, 65537, 100003])
def test_ragged_batch_sizes(oracle, n):
    """Batch lengths around the wave size, the claim-chunk size and the threshold of the grouping pass."""
    scene, want_scene = mesh_pair(oracle, *synth.soup(20000))
    rays = synth.rays(n, key=1000 + n)
    want = want_scene.trace_closest(rays)
    for mode in MODES:
        same(scene.trace_closest(rays, flags=mode), want, f"n={n} mode={mode}")
    shadow = synth.shadow_rays(n, key=2000 + n)
    want_any = want_scene.trace_any(shadow)
    for mode in MODES:
        assert np.array_equal(scene.trace_any(shadow, flags=mode), want_any)
    scene.close()


def hostile_rays(n, key):
    """Ordinary rays with every 7th one spoiled in one of twelve ways."""
    rays = synth.rays(n, key=key)
    k = np.arange(0, n, 7)
    kind = (k // 7) % 12
    f = np.float32
    rays["d"][k[kind == 0], 0] = np.nan
    rays["o"][k[kind == 1], 1] = np.nan
    rays["o"][k[kind == 2], 2] = np.inf
    rays["d"][k[kind == 3]] = 0.0                         # no direction at all
    rays["d"][k[kind == 4]] = f(1e-30) * rays["d"][k[kind == 4]]   # tiny, un-normalised
    rays["o"][k[kind == 5]] = f(1e30)                     # far outside the exact-division range
    rays["tmax"][k[kind == 6]] = 0.0
    rays["tmin"][k[kind == 7]], rays["tmax"][k[kind == 7]] = 5.0, 1.0   # flipped bounds
    rays["d"][k[kind == 8]] = np.array([0, 0, 1], dtype=f)              # two zero components
    rays["d"][k[kind == 9]] = -rays["d"][k[kind == 9]] * f(3.0)         # un-normalised
    rays["tmax"][k[kind == 10]] = np.nan
    rays["o"][k[kind == 11]] = np.array([5, 5, 5], dtype=f) + f(1e-38)  # denormal offsets
    return rays


def test_hostile_rays_on_a_mesh(oracle):
    scene, want_scene = mesh_pair(oracle, *synth.soup(20000))
    for n, key in ((5000, 5), (1 << 17, 6)):     # below and above the threshold of the grouping pass
        rays = hostile_rays(n, key)
        want = want_scene.trace_closest(rays)
        for mode in MODES + [abi.PTX_TRAVERSE_CULLED | abi.PTX_RAYS_COHERENT]:
            same(scene.trace_closest(rays, flags=mode), want, f"hostile n={n} mode={mode}")
        want_any = want_scene.trace_any(rays)
        for mode in MODES:
            assert np.array_equal(scene.trace_any(rays, flags=mode), want_any), f"hostile occlusion n={n} mode={mode}"
    scene.close()


def test_hostile_rays_on_an_instanced_scene(oracle):
    sb = scenes.instanced_scene()
    desc = sb.desc()
    want_scene = oracle.scene(desc)
    scene = capi.Scene(desc=desc, keep=sb)
    rays = hostile_rays(1 << 14, 9)
    rays["o"] = (rays["o"] - 5.0) * np.float32(0.4)      # into the room
    want = want_scene.trace_closest(rays)
    for mode in MODES:
        got = scene.trace_closest(rays, flags=mode)
        same(got, want, f"hostile instanced mode={mode}")
        hit = want["hits"]["prim"] != abi.PTX_NO_HIT
        assert np.array_equal(got["inst"][hit], want["inst"][hit])
    scene.close()


def test_coincident_and_degenerate_triangles(oracle):
    """Every triangle three times over (ties between equal distances), plus zero-area and point triangles."""
    verts, idx = scenes.icosphere(3)
    verts = np.concatenate([verts, verts[:3] * 0 + verts[:1]])     # three copies of one point
    n0 = len(verts) - 3
    extra = np.array([[n0, n0 + 1, n0 + 2],          # a point
                      [0, 0, 1], [2, 3, 3],          # zero-area slivers (two corners equal)
                      ], dtype=np.uint32)
    idx = np.concatenate([idx, idx, extra, idx[::-1]])
    scene, want_scene = mesh_pair(oracle, verts, idx)
    rays = synth.rays(1 << 15, key=77)
    rays["o"] = (rays["o"] - 5.0) * np.float32(0.5)
    rays["d"][::3] = synth._unit_f32(-rays["o"][::3].astype(np.float64) / np.maximum(1e-9, np.linalg.norm(rays["o"][::3].astype(np.float64), axis=1, keepdims=True)))
    want = want_scene.trace_closest(rays)
    assert (want["hits"]["prim"] != abi.PTX_NO_HIT).mean() > 0.2
    for mode in MODES:
        same(scene.trace_closest(rays, flags=mode), want, f"coincident mode={mode}")
    scene.close()


@pytest.mark.parametrize("max_leaf", [1, 2, 15])
def test_leaf_sizes(oracle, max_leaf):
    scene, want_scene = mesh_pair(oracle, *synth.soup(6000, key=3, half=0.3), max_leaf_size=max_leaf)
    rays = synth.rays(1 << 14, key=11)
    want = want_scene.trace_closest(rays)
    for mode in MODES:
        same(scene.trace_closest(rays, flags=mode), want, f"max_leaf={max_leaf} mode={mode}")
    scene.close()


def test_full_size_properties():
    """1 M triangles, 4 M rays -- more than the CPU checker covers in a test: the two traversal orders agree
    bit for bit, a second run repeats the first, permuting the rays permutes the answers, and a ray whose
    upper bound is cut to just above its hit distance still reports that hit while one cut below it reports a
    farther hit or none."""
    import torch
    verts, idx = synth.soup(1_000_000)
    scene = capi.Scene.from_mesh(verts, idx, device=0)
    n = 1 << 22
    dev = torch.device("cuda", 0)
    rays_h = synth.rays(n, key=4242)
    rays = torch.from_numpy(rays_h.view(np.float32).reshape(-1, 8)).to(dev)
    ref = torch.empty((n, 2), dtype=torch.int32, device=dev)
    cul = torch.empty_like(ref)
    again = torch.empty_like(ref)
    scene.trace_closest_device(rays, ref, flags=abi.PTX_TRAVERSE_REFERENCE, n=n)
    scene.trace_closest_device(rays, cul, flags=abi.PTX_TRAVERSE_CULLED, n=n)
    scene.trace_closest_device(rays, again, flags=abi.PTX_TRAVERSE_CULLED, n=n)
    torch.cuda.synchronize()
    assert torch.equal(ref, cul) and torch.equal(cul, again)
    hit = cul[:, 1] != -1
    assert 0.5 < float(hit.float().mean()) < 0.95

    perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    shuffled = rays[perm].contiguous()
    out = torch.empty_like(ref)
    scene.trace_closest_device(shuffled, out, flags=abi.PTX_TRAVERSE_CULLED, n=n)
    torch.cuda.synchronize()
    assert torch.equal(out, cul[perm])

    dist = cul[:, 0].view(torch.float32)
    above = rays.clone()
    above[:, 7] = torch.where(hit, torch.nextafter(dist, torch.full_like(dist, float("inf"))), above[:, 7])
    scene.trace_closest_device(above, out, flags=abi.PTX_TRAVERSE_CULLED, n=n)
    torch.cuda.synchronize()
    assert torch.equal(out, cul)          # distance <= dist_bounds.y keeps the hit (tri_mesh.cpp:93)
    below = rays.clone()
    below[:, 7] = torch.where(hit, torch.nextafter(dist, torch.zeros_like(dist)), below[:, 7])
    scene.trace_closest_device(below, out, flags=abi.PTX_TRAVERSE_CULLED, n=n)
    torch.cuda.synchronize()
    lost = out[:, 1] == -1
    assert bool((lost | (out[:, 1] != cul[:, 1]) | ~hit).all())
    assert bool((out[:, 0].view(torch.float32)[hit & ~lost] <= below[:, 7][hit & ~lost]).all())
    scene.close()


def test_zero_direction_components_in_bulk(oracle):
    """Batches of axis-aligned rays (orthographic views, straight-down shadow rays) -- direction components that are
    exactly +0 or -0, one or two of them, origins inside, outside and exactly ON box planes (0/0 in BBox::hit,
    bbox.h:98-120) -- walk the replenishing kernels' zero-aware second pass.  Mesh and instanced scene (axis-aligned
    instance transforms keep the zeros in object space), every mode, against the checker: bit-exact."""
    g = np.random.Generator(np.random.Philox(key=404))
    n = 1 << 17
    dirs = np.array([[0, -1, 0], [0, 1, 0], [1, 0, 0], [0, 0, -1], [-0.0, -1, 0], [0, -1, -0.0], [-0.0, 0, 1],
                     [0.6, 0, -0.8], [0, 0.6, 0.8], [-0.8, -0.6, 0], [0.6, -0.0, 0.8]], dtype=np.float32)

    def axis_rays(lo, hi, snap):
        r = np.zeros(n, dtype=abi.RAY_DTYPE)
        r["o"] = (lo - 0.2 * (hi - lo) + g.random((n, 3)) * 1.4 * (hi - lo)).astype(np.float32)
        # a share of the origins sits exactly on coordinates that are box planes of the tree
        k = n // 8
        pick = g.integers(0, len(snap), (k, 3))
        for a in range(3):
            on = g.random(k) < 0.5
            r["o"][:k, a] = np.where(on, snap[pick[:, a], a], r["o"][:k, a])
        r["d"] = dirs[g.integers(0, len(dirs), n)]
        r["tmin"], r["tmax"] = 0.0, np.inf
        r["tmax"][::5] = np.float32(0.7) * np.float32(np.linalg.norm(hi - lo))
        return r

    verts, idx = synth.soup(30000, key=41, extent=4.0, half=0.1)
    sb = abi.SceneBuilder()
    md = sb.mesh_desc(verts, idx)
    rays = axis_rays(verts[:, :3].min(0), verts[:, :3].max(0), verts[:, :3])
    want_scene = oracle.mesh_scene(md)
    want, want_any = want_scene.trace_closest(rays), want_scene.trace_any(rays)
    scene = capi.Scene(mesh_desc=md, keep=sb)
    for mode in MODES:
        got = scene.trace_closest(rays, flags=mode)
        assert np.array_equal(bits(got["hits"]["distance"]), bits(want["hits"]["distance"])) and np.array_equal(got["hits"]["prim"], want["hits"]["prim"]), f"mesh mode {mode}"
        assert np.array_equal(scene.trace_any(rays, flags=mode), want_any), f"mesh occlusion mode {mode}"
    scene.close()
    assert 0.2 < (want["hits"]["prim"] != abi.PTX_NO_HIT).mean() < 0.999

    # instanced: axis-aligned placements (translation, 90-degree turns, scale) keep zero components zero in object space
    sb = abi.SceneBuilder()
    m = sb.add_material(abi.PTX_MAT_LAMBERTIAN, sb.add_constant_texture((0.5, 0.5, 0.5)))
    soup = sb.add_mesh(*synth.soup(2000, key=42, extent=1.0, half=0.08))
    q = sb.add_mesh(*scenes.quad())
    ico = sb.add_mesh(*scenes.icosphere(2))
    for k in range(27):
        pos = (float(k % 3) * 1.5, float((k // 3) % 3) * 1.5, float(k // 9) * 1.5)
        what = k % 4
        if what == 0:
            sb.add_instance(soup, m, scenes.trs(pos))
        elif what == 1:
            sb.add_instance(q, m, scenes.trs(pos, axis=(1, 0, 0), deg=0.0, scale=(0.7, 1.0, 0.5)))
        elif what == 2:
            sb.add_instance(ico, m, scenes.trs(pos, scale=(0.4, 0.6, 0.5)))
        else:
            sb.add_instance(-1, m, scenes.trs(pos, scale=(0.5, 0.5, 0.5)), sphere_radius=1.0)
    desc = sb.desc()
    lo, hi = np.array([-1.0, -1.0, -1.0]), np.array([4.5, 4.5, 4.5])
    snap = np.array([[0.0, 0.0, 0.0], [1.5, 1.5, 1.5], [3.0, 3.0, 3.0], [0.7, 0.0, 0.5], [2.2, 1.5, 2.0]], dtype=np.float32)
    rays = axis_rays(lo, hi, snap)
    osc = oracle.scene(desc)
    want, want_any = osc.trace_closest(rays), osc.trace_any(rays)
    osc.close()
    sc = capi.Scene(desc, keep=sb)
    for mode in MODES:
        got = sc.trace_closest(rays, flags=mode)
        hit = want["hits"]["prim"] != abi.PTX_NO_HIT
        assert np.array_equal(bits(got["hits"]["distance"]), bits(want["hits"]["distance"])) and np.array_equal(got["hits"]["prim"], want["hits"]["prim"]), f"scene mode {mode}"
        assert np.array_equal(got["inst"][hit], want["inst"][hit])
        assert np.array_equal(sc.trace_any(rays, flags=mode), want_any), f"scene occlusion mode {mode}"
    sc.close()
    assert 0.1 < hit.mean() < 0.999

"""ptx_trace_closest_noalloc (the entry point for callers that cannot allocate) against ptx_trace_closest:
same distance bits, primitive, barycentrics and instance on every ray, on calls in a row that share the
library's fixed device scratch array -- the later ones with smaller inputs, so stale contents of the
array from the larger call are in place when they run."""
import numpy as np
import pytest

import scenes
from pathtracer_amd import abi, capi, synth

pytestmark = pytest.mark.gpu


def _device_pair(scene, rays_h, flags, noalloc):
    import torch
    dev = torch.device("cuda", 0)
    n = len(rays_h)
    rays = torch.from_numpy(rays_h.view(np.float32).reshape(-1, 8).copy()).to(dev)
    hits = torch.full((n, 2), 0x7FC00001, dtype=torch.int32, device=dev)     # poison: every slot must be written
    bary = torch.full((n, 2), -7.0, dtype=torch.float32, device=dev)
    inst = torch.full((n,), 0x12345678, dtype=torch.int32, device=dev)
    if noalloc:
        scene.trace_closest_noalloc(rays, hits, bary, inst, flags=flags | abi.PTX_MEM_DEVICE, n=n)
    else:
        scene.trace_closest_device(rays, hits, bary, inst, flags=flags, n=n)
    torch.cuda.synchronize()
    return hits.cpu().numpy(), bary.cpu().numpy().view(np.uint32), inst.cpu().numpy()


def _rays(n, key):
    rays = synth.rays(n, key=key)
    # a sprinkling of rays outside the exact-division range (axis-parallel directions): the set-aside list
    rays["d"][::97] = np.eye(3, dtype=np.float32)[np.arange(len(rays[::97])) % 3]
    return rays


def test_noalloc_equals_allocating_entry_point_on_calls_in_a_row():
    verts, idx = synth.soup(200000)
    scene = capi.Scene.from_mesh(verts, idx)
    assert capi.lib().ptx_noalloc_max_rays() == 1 << 20
    hit_any = False
    # largest first (fills the whole array), then smaller ones: one still grouped, one below the grouping threshold
    for k, n in enumerate((1 << 20, 70001, 1000)):
        rays = _rays(n, key=900 + k)
        for flags in (abi.PTX_TRAVERSE_CULLED, abi.PTX_TRAVERSE_REFERENCE, abi.PTX_TRAVERSE_CULLED | abi.PTX_RAYS_COHERENT):
            want = _device_pair(scene, rays, flags, noalloc=False)
            got = _device_pair(scene, rays, flags, noalloc=True)
            for g, w, what in zip(got, want, ("hits", "bary", "inst")):
                assert np.array_equal(g, w), f"n={n} flags={flags}: {what} differ on {np.count_nonzero((g != w).reshape(len(g), -1).any(axis=1))} rays"
            hit_any |= bool((want[0][:, 1] != -1).any())
    assert hit_any
    scene.close()


def test_noalloc_limits_and_other_scene_kinds(oracle):
    import torch
    verts, idx = synth.soup(5000)
    scene = capi.Scene.from_mesh(verts, idx)
    dev = torch.device("cuda", 0)
    rays = torch.zeros((16, 8), dtype=torch.float32, device=dev)
    hits = torch.zeros((16, 2), dtype=torch.int32, device=dev)
    with pytest.raises(capi.PtxError, match="exceed the fixed scratch"):
        scene.trace_closest_noalloc(rays, hits, n=(1 << 20) + 1)
    with pytest.raises(capi.PtxError, match="device pointers"):
        scene.trace_closest_noalloc(rays, hits, flags=0)
    scene.close()
    # two-level scenes and lists take the one-ray-per-lane kernels, which need no working buffers
    sb = scenes.instanced_scene(use_bvh=True, mesh_bvh=True)
    sc = capi.Scene(desc=sb.desc(), keep=sb)
    r = scenes.scene_rays(1 << 13, seed=5, inside=True)
    for flags in (abi.PTX_TRAVERSE_CULLED, abi.PTX_TRAVERSE_REFERENCE):
        want = _device_pair(sc, r, flags, noalloc=False)
        got = _device_pair(sc, r, flags, noalloc=True)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
    sc.close()

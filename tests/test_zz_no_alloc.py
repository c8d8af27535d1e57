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


def test_noalloc_allocates_nothing():
    """The entry point's defining property: on a scene that no allocating call has touched (so none of the scene's
    grown-on-demand buffers exist), free device memory is the same before and after every call, at the largest size, a
    grouped smaller one and one below the grouping threshold, synchronous and PTX_ASYNC; only afterwards does the
    allocating entry point run on the same scene, and its answers are the same."""
    import torch
    dev = torch.device("cuda", 0)
    verts, idx = synth.soup(150000, key=77)
    scene = capi.Scene.from_mesh(verts, idx)
    scene.set_stream(torch.cuda.current_stream().cuda_stream)
    sizes = (1 << 20, 70001, 1000)
    batches = [_rays(n, key=950 + k) for k, n in enumerate(sizes)]
    rays = [torch.from_numpy(r.view(np.float32).reshape(-1, 8).copy()).to(dev) for r in batches]
    hits = [torch.zeros((n, 2), dtype=torch.int32, device=dev) for n in sizes]
    bary = [torch.zeros((n, 2), dtype=torch.float32, device=dev) for n in sizes]
    inst = [torch.zeros((n,), dtype=torch.int32, device=dev) for n in sizes]
    want = [(torch.zeros_like(h), torch.zeros_like(b), torch.zeros_like(i)) for h, b, i in zip(hits, bary, inst)]
    # code objects load on a kernel's first launch: one call per mode before measuring
    for flags in (abi.PTX_TRAVERSE_CULLED, abi.PTX_TRAVERSE_REFERENCE):
        scene.trace_closest_noalloc(rays[0], hits[0], bary[0], inst[0], flags=flags | abi.PTX_MEM_DEVICE, n=sizes[0])
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for flags in (abi.PTX_TRAVERSE_CULLED, abi.PTX_TRAVERSE_REFERENCE, abi.PTX_TRAVERSE_CULLED | abi.PTX_RAYS_COHERENT, abi.PTX_TRAVERSE_CULLED | abi.PTX_ASYNC):
        for k, n in enumerate(sizes):
            scene.trace_closest_noalloc(rays[k], hits[k], bary[k], inst[k], flags=flags | abi.PTX_MEM_DEVICE, n=n)
            if flags & abi.PTX_ASYNC:
                torch.cuda.current_stream().synchronize()
            assert torch.cuda.mem_get_info(0)[0] == free0, f"flags={flags} n={n}: free device memory changed across an allocation-free call"
    torch.cuda.synchronize()
    for k, n in enumerate(sizes):       # now the allocating entry point, for the comparison
        scene.trace_closest_device(rays[k], want[k][0], want[k][1], want[k][2], flags=abi.PTX_TRAVERSE_CULLED, n=n)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(0)[0] < free0        # ... which does grow the scene's buffers
    for k in range(len(sizes)):
        assert torch.equal(hits[k], want[k][0]) and torch.equal(bary[k].view(torch.int32), want[k][1].view(torch.int32)) and torch.equal(inst[k], want[k][2])
    scene.close()


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

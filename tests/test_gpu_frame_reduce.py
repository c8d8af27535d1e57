"""The film's sum across workers (pathtracer.cpp:407-427, 566-584: every finished tile is added into one accumulator;
pathtracer.h:80-83: integer sums, so the order does not matter): ptx_frame_reduce's three ways of reaching a part
(where it lies, staged through a copy) and the RCCL collectives, all against plain integer addition on the host."""
import numpy as np
import pytest

import scenes
from pathtracer_amd import abi, capi

pytestmark = pytest.mark.gpu

W, H = 250, 130      # an odd number of int64 words per film: the adding kernel's scalar tail is exercised


@pytest.fixture(scope="module")
def scene():
    sb = scenes.instanced_scene()
    sc = capi.Scene(sb.desc(), keep=sb)
    yield sc
    sc.close()


def random_film(key):
    g = np.random.Generator(np.random.Philox(key=key))
    acc = g.integers(-(1 << 40), 1 << 40, 3 * W * H, dtype=np.int64)
    smp = g.integers(0, 1 << 20, W * H, dtype=np.uint64).astype(np.uint32)
    return acc, smp


def films(scene, n, key0):
    out, host = [], []
    for k in range(n):
        f = capi.Frame(scene, W, H)
        a, s = random_film(key0 + k)
        f.upload(a, s)
        out.append(f)
        host.append((a, s))
    return out, host


@pytest.mark.parametrize("n_parts", [1, 3, 8, 11])
def test_frame_reduce_adds_parts_where_they_lie(scene, n_parts, monkeypatch):
    monkeypatch.delenv("PTX_FRAME_REDUCE_STAGE", raising=False)
    parts, host = films(scene, n_parts, 100)
    dst = capi.Frame(scene, W, H)
    base = random_film(7)
    dst.upload(*base)
    info = dst.reduce(parts)
    acc, smp = dst.download()
    assert np.array_equal(acc, base[0] + sum(a for a, _ in host))
    assert np.array_equal(smp, (base[1].astype(np.uint64) + sum(s.astype(np.uint64) for _, s in host)).astype(np.uint32))
    assert (info.parts_local, info.parts_peer_direct, info.parts_staged) == (n_parts, 0, 0)
    assert info.add_launches == (n_parts + 7) // 8 and info.allocations == 0 and info.bytes_over_links == 0
    for f, (a, s) in zip(parts, host):      # parts unchanged
        fa, fs = f.download()
        assert np.array_equal(fa, a) and np.array_equal(fs, s)
        f.close()
    dst.close()


def test_frame_reduce_staging_path_allocates_once(scene, monkeypatch):
    """The path for films on GPUs the destination cannot address (copies into staging films), forced on one GPU: same
    sums, and a destination that has served one call allocates nothing in the next -- a render with a hundred progress
    reports makes its staging allocations once."""
    monkeypatch.setenv("PTX_FRAME_REDUCE_STAGE", "1")
    parts, host = films(scene, 3, 200)
    dst = capi.Frame(scene, W, H)
    want_a, want_s = np.zeros(3 * W * H, dtype=np.int64), np.zeros(W * H, dtype=np.uint64)
    for call in range(100):
        info = dst.reduce(parts)
        want_a += sum(a for a, _ in host)
        want_s += sum(s.astype(np.uint64) for _, s in host)
        assert (info.parts_local, info.parts_peer_direct, info.parts_staged) == (0, 0, 3)
        assert info.allocations == (7 if call == 0 else 0), (call, info.allocations)     # staging memory, 3 streams, 3 events
    acc, smp = dst.download()
    assert np.array_equal(acc, want_a) and np.array_equal(smp, want_s.astype(np.uint32))
    for f in parts:
        f.close()
    dst.close()


def test_rccl_collectives_on_a_one_rank_communicator(scene):
    """ptx_frame_reduce_rccl / ptx_frame_allreduce through RCCL itself with the one rank a one-GPU box allows: the
    result must equal ptx_frame_reduce's bit for bit (more ranks need more GPUs: not measured here)."""
    comm = capi.Comm([0])
    assert comm.size == 1 and "rccl" in comm.library
    parts, host = films(scene, 1, 300)
    dst = capi.Frame(scene, W, H)
    dst.upload(*random_film(9))              # overwritten: dst = sum of parts
    comm.reduce(dst, parts)
    acc, smp = dst.download()
    assert np.array_equal(acc, host[0][0]) and np.array_equal(smp, host[0][1])
    peer = capi.Frame(scene, W, H)
    peer.reduce(parts)
    pa, ps = peer.download()
    assert np.array_equal(acc, pa) and np.array_equal(smp, ps)
    comm.reduce(parts[0], parts)             # in place
    comm.allreduce(parts)
    a, s = parts[0].download()
    assert np.array_equal(a, host[0][0]) and np.array_equal(s, host[0][1])
    with pytest.raises(capi.PtxError):
        comm.reduce(dst, parts + parts)      # one film per rank
    with pytest.raises(capi.PtxError):
        capi.Comm([0, 0])                    # one rank per GPU
    for f in parts + [dst, peer]:
        f.close()
    comm.close()


def test_render_into_films_and_reduce(scene):
    """A render dealt to three workers' films on the GPU and reduced equals the one-film render bit for bit."""
    cam = abi.PtxCamera()
    cam.vertical_fov, cam.aspect_ratio = 60.0, W / H
    cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth = W, H, 60, 4
    T = scenes.trs((0, 0, 3.5))
    for i in range(16):
        cam.camera_to_world[i] = float(T[i])
    tiles = capi.make_tiles(W, H, 60, 11)
    whole = capi.Frame(scene, W, H)
    whole.render_tiles(cam, tiles, 5)
    want = whole.download()
    parts = [capi.Frame(scene, W, H) for _ in range(3)]
    for k, f in enumerate(parts):
        f.render_tiles(cam, tiles[k::3], 5)
    total = capi.Frame(scene, W, H)
    total.reduce(parts)
    got = total.download()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and int(want[1].min()) == 60
    for f in parts + [whole, total]:
        f.close()

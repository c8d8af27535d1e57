"""World-size-2 rehearsal of the N > 1 path on CPU (gloo): tiles dealt across ranks, private
accumulators, one integer all-reduce -- the reduced frame must equal a single-process render of all
tiles bit for bit (integer addition commutes; pathtracer.h:80-83 relies on the same property)."""
import os
import socket
import subprocess
import sys

import numpy as np

from pathtracer_amd import abi, capi, dist as pdist, synth
import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x5EED
FILM = (150, 110, 60)      # 2 x 2 tiles x 2 sample ranges = 8 tiles


def make_camera():
    w, h, spp = FILM
    cam = abi.PtxCamera()
    cam.vertical_fov, cam.aspect_ratio = 60.0, w / h
    cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth = w, h, spp, 3
    T = scenes.trs((0, 0, 3.4))
    for i in range(16):
        cam.camera_to_world[i] = float(T[i])
    return cam


def test_deal_tiles_and_ray_ranges():
    tiles = capi.make_tiles(1024, 1024, 256, 1)
    for world in (1, 2, 3, 8):
        parts = [pdist.deal_tiles(tiles, r, world) for r in range(world)]
        assert sum(len(p) for p in parts) == len(tiles)
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
        merged = np.concatenate(parts)
        assert sorted(map(bytes, merged)) == sorted(map(bytes, tiles))
        ranges = [pdist.ray_range(1000003, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == 1000003
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))


def test_two_ranks_reduce_to_the_single_process_frame(oracle, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "frame.npz")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), out], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = np.load(out)
    w, h, spp = FILM
    sb = scenes.instanced_scene()
    sc = oracle.scene(sb.desc())
    acc, smp = sc.render_tiles(make_camera(), capi.make_tiles(w, h, spp, 99), 1, SEED)
    assert (smp == spp).all()
    assert (got["smp"] == smp).all()
    assert (got["acc"] == acc).all()
    # the two ray ranges tile the batch and their hits concatenate to the single-process result
    rays = synth.rays(5001)
    assert got["ranges"].tolist() == [[0, 2501], [2501, 5001]]
    assert got["hits"].tobytes() == sc.trace_closest(rays)["hits"].tobytes()
    sc.close()

"""Time from ptx_scene_create_mesh's call to a resident scene, for triangle soups:  python3 tools/time_scene_create.py [n_tris ...]
(PTX_BUILD_PROFILE=1 prints the laps of the construction passes; PTX_GPU_BUILD=0 keeps the host builder)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import capi, synth

for n in [int(a) for a in sys.argv[1:]] or [1_000_000, 10_000_000]:
    verts, idx = synth.soup(n)
    capi.Scene.from_mesh(verts[:300], idx[:100], device=0).close()
    t0 = time.perf_counter()
    sc = capi.Scene.from_mesh(verts, idx, device=0)
    dt = time.perf_counter() - t0
    print(f"{n} triangles: scene resident after {dt * 1e3:.0f} ms", flush=True)
    sc.close()

"""Build times of the host builder and the GPU builder on triangle soups:  python3 tools/time_builder.py [n_tris ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pathtracer_amd import capi, synth

for n in [int(a) for a in sys.argv[1:]] or [1_000_000, 10_000_000]:
    verts, idx = synth.soup(n)
    g = np.random.Generator(np.random.Philox(key=1))
    lo = np.minimum(np.minimum(verts[idx[:, 0], :3], verts[idx[:, 1], :3]), verts[idx[:, 2], :3])
    hi = np.maximum(np.maximum(verts[idx[:, 0], :3], verts[idx[:, 1], :3]), verts[idx[:, 2], :3])
    boxes = np.concatenate([lo, hi], axis=1).astype(np.float32)
    capi.Bvh(boxes=boxes[:1000], max_leaf_size=4, gpu_device=0)   # context, code load
    t = {}
    for what, dev in (("host", None), ("gpu", 0), ("gpu again", 0)):
        t0 = time.perf_counter()
        b = capi.Bvh(boxes=boxes, max_leaf_size=4, gpu_device=dev)
        t[what] = time.perf_counter() - t0
        if what == "host":
            ref = (b.nodes().tobytes(), b.order().tobytes())
        else:
            assert b.built_on_device and (b.nodes().tobytes(), b.order().tobytes()) == ref
    print(f"{n} boxes: host {t['host']*1e3:.1f} ms, gpu {t['gpu']*1e3:.1f} ms, gpu again {t['gpu again']*1e3:.1f} ms (upload of the boxes and download of the tree included), identical trees, depth {b.max_depth}", flush=True)

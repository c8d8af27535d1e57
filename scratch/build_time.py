"""Host builder timing and a digest of its output (same digest whatever PTX_BUILD_THREADS is)."""
import time, numpy as np, sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import capi, synth
capi.lib()
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
verts, idx = synth.soup(nt)
for rep in range(2):
    t0 = time.perf_counter(); b = capi.Bvh(verts=verts, indices=idx, max_leaf_size=4); dt = time.perf_counter() - t0
    print(os.environ.get("PTX_BUILD_THREADS"), nt, "build %.2f s" % dt, len(b.nodes()), hashlib.sha1(b.nodes().tobytes() + b.order().tobytes()).hexdigest()[:12], flush=True)

import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pathtracer_amd import abi, capi, synth
verts, idx = synth.soup(1_000_000)
scene = capi.Scene.from_mesh(verts, idx, device=0)
rays = synth.rays(1 << 18)
for mode, name in ((abi.PTX_TRAVERSE_REFERENCE, 'reference'), (abi.PTX_TRAVERSE_CULLED, 'culled')):
    st = scene.trace_closest_stats(rays, flags=mode)
    max_sp = st[:, 0] >> 24
    st = st.copy(); st[:, 0] &= 0xFFFFFF
    st = st.astype(np.float64)
    w = st.reshape(-1, 64, 3)
    print(name, 'per ray mean nodes/leaves/tris', st.mean(0), 'max', st.max(0))
    print('   stack depth: mean max_sp %.2f, p50 %d p90 %d p99 %d max %d' % (max_sp.mean(), *np.percentile(max_sp, [50, 90, 99]), max_sp.max()))
    print('   per wave: mean of max', w.max(1).mean(0), ' SIMD efficiency (mean/max)', (w.mean(1)/np.maximum(w.max(1),1)).mean(0))

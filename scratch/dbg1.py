import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from _oracle import Oracle
from pathtracer_amd import abi, capi, synth
verts, idx = synth.soup(20000)
sb = abi.SceneBuilder(); md = sb.mesh_desc(verts, idx)
scene = capi.Scene(mesh_desc=md, device=0, keep=sb)
ws = Oracle().mesh_scene(md)
rays = synth.rays(1 << 14, key=1234)
want = ws.trace_closest(rays)
for mode in (0, 16):
    got = scene.trace_closest(rays, flags=mode)
    gd, wd = got['hits']['distance'].view(np.uint32), want['hits']['distance'].view(np.uint32)
    bad = np.flatnonzero((gd != wd) | (got['hits']['prim'] != want['hits']['prim']))
    print('mode', mode, 'bad', len(bad), 'of', len(rays), 'hits', (want['hits']['prim']!=0xFFFFFFFF).sum())
    for b in bad[:8]:
        print(b, 'got', got['hits'][b], 'want', want['hits'][b], rays[b])

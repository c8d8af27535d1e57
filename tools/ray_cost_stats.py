"""Per-ray traversal cost on the config-3 workload against what can be known before tracing (the chord of the ray
through the root box): python3 tools/ray_cost_stats.py [n_rays]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pathtracer_amd import abi, capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
verts, idx = synth.soup(1_000_000)
scene = capi.Scene.from_mesh(verts, idx, device=0)
rays = synth.rays(n)
st = scene.trace_closest_stats(rays, flags=abi.PTX_TRAVERSE_CULLED).astype(np.float64)
hits = scene.trace_closest(rays, flags=abi.PTX_TRAVERSE_CULLED)["hits"]
steps = st[:, 0] + st[:, 2]          # interior nodes fetched + triangles tested
lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
o, d = rays["o"].astype(np.float64), rays["d"].astype(np.float64)
with np.errstate(divide="ignore", invalid="ignore"):
    t1, t2 = (lo - o) / d, (hi - o) / d
t_exit = np.minimum.reduce(np.maximum(t1, t2), axis=1)
t_enter = np.maximum(np.maximum.reduce(np.minimum(t1, t2), axis=1), 0)
chord = np.clip(t_exit - t_enter, 0, None)
hit = hits["prim"] != abi.PTX_NO_HIT
print(f"rays {n}: mean steps {steps.mean():.1f}, median {np.median(steps):.0f}, p90 {np.percentile(steps, 90):.0f}, p99 {np.percentile(steps, 99):.0f}, max {steps.max():.0f}; hit fraction {hit.mean():.3f}")
print(f"mean over groups of 64 consecutive rays of the group's largest step count: {steps[: n // 64 * 64].reshape(-1, 64).max(1).mean():.1f}")
print(f"correlation of steps with chord {np.corrcoef(steps, chord)[0, 1]:.3f}, with min(chord, hit distance) {np.corrcoef(steps, np.where(hit, np.minimum(chord, hits['distance']), chord))[0, 1]:.3f}")
q = np.quantile(chord, np.linspace(0, 1, 11))
for k in range(10):
    m = (chord >= q[k]) & (chord <= q[k + 1])
    print(f"chord decile {k} [{q[k]:.2f}, {q[k + 1]:.2f}]: mean steps {steps[m].mean():.1f}, p90 {np.percentile(steps[m], 90):.0f}, p99 {np.percentile(steps[m], 99):.0f}, hit fraction {hit[m].mean():.2f}")

"""Seeded synthetic workloads: the triangle soup and the incoherent ray batches.

Everything is generated with numpy's Philox bit generator from a documented key, in
float32 with the operation order of the reference's Ray constructor (lib/ray.h:16-20:
dir.unit() = each component divided by sqrt(x*x + y*y + z*z)), so the same arrays can be
rebuilt anywhere and fed to the GPU path, the CPU checker and the reference alike.
"""
import numpy as np

from . import abi

SOUP_KEY = 0x5C077D          # triangle soup
RAY_KEY = 0x5C077D + 1       # closest-hit ray batch
SHADOW_KEY = 0x5C077D + 2    # occlusion ray batch


def _rng(key):
    return np.random.Generator(np.random.Philox(key=key))


def soup(n_tris, key=SOUP_KEY, extent=10.0, half=0.04):
    """n_tris triangles: centre uniform in [0,extent)^3, vertices centre + U[-half,half)^3,
    normals (0,1,0), uv 0 (the shape of tests/a3/test.a3.task3.bvh.fuzz.cpp:13-39, scaled so a
    ray meets O(1) triangles).  Returns (verts [3n,8] f32, indices [n,3] u32)."""
    g = _rng(key)
    centre = g.random((n_tris, 1, 3), dtype=np.float32) * np.float32(extent)
    off = (g.random((n_tris, 3, 3), dtype=np.float32) * np.float32(2 * half)) - np.float32(half)
    verts = np.zeros((n_tris * 3, 8), dtype=np.float32)
    verts[:, :3] = (centre + off).reshape(-1, 3)
    verts[:, 4] = 1.0
    return verts, np.arange(n_tris * 3, dtype=np.uint32).reshape(-1, 3)


def _unit_f32(d):
    d = d.astype(np.float32)
    n = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])   # vec3.h:155-160
    return d / n[:, None]


def rays(n, key=RAY_KEY, extent=10.0, chunk=1 << 22):
    """n rays: origin uniform in [0,extent)^3, direction uniform on the sphere, passed through
    Ray's constructor (unit()); dist_bounds (0, inf)."""
    g = _rng(key)
    out = np.zeros(n, dtype=abi.RAY_DTYPE)
    for b in range(0, n, chunk):
        m = min(chunk, n - b)
        out["o"][b:b + m] = g.random((m, 3), dtype=np.float32) * np.float32(extent)
        d = g.standard_normal((m, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        out["d"][b:b + m] = _unit_f32(d)
    out["tmin"] = 0.0
    out["tmax"] = np.inf
    return out


def shadow_rays(n, key=SHADOW_KEY, extent=10.0, chunk=1 << 22):
    """Like rays(), with dist_bounds (1e-5, U[0.5, 5)): the shape of sum_delta_lights' shadow
    ray (pathtracer.cpp:657: Vec2{EPS_F, distance - EPS_F})."""
    g = _rng(key)
    out = rays(n, key=key + 0x1000, extent=extent, chunk=chunk)
    out["tmin"] = np.float32(1e-5)
    out["tmax"] = np.float32(0.5) + g.random(n, dtype=np.float32) * np.float32(4.5)
    return out

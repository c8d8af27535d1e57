"""Seeded ray generators shared by tests/golden/make_golden.py and the tests (inputs only)."""
import numpy as np

from pathtracer_amd import abi, synth


def mesh_rays(verts, n, seed):
    """Rays aimed into the mesh's bounding box from around it (about half of them hit something)."""
    g = np.random.Generator(np.random.Philox(key=seed))
    lo, hi = verts[:, :3].min(axis=0), verts[:, :3].max(axis=0)
    c, e = (lo + hi) / 2, (hi - lo)
    o = c + (g.random((n, 3)) * 2 - 1) * e * 1.2
    t = lo + g.random((n, 3)) * e
    d = t - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros(n, dtype=abi.RAY_DTYPE)
    r["o"] = o.astype(np.float32)
    r["d"] = synth._unit_f32(d)
    r["tmin"], r["tmax"] = 0.0, np.inf
    # a quarter of the rays get finite bounds, as shadow rays have
    r["tmin"][::4] = 1e-5
    r["tmax"][::4] = (g.random(len(r["tmax"][::4])) * np.linalg.norm(e)).astype(np.float32)
    return r


def grazing_rays(verts, idx, n, seed):
    """Rays that skim the mesh: from a point just off a random triangle's plane towards a point of a neighbouring
    triangle's plane, nearly parallel to both -- thin triangles hit at shallow angles."""
    g = np.random.Generator(np.random.Philox(key=seed))
    tri = idx[g.integers(0, len(idx), n)]
    a, b, c = (verts[tri[:, k], :3].astype(np.float64) for k in range(3))
    nrm = np.cross(b - a, c - a)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    w = g.random((n, 3)) + 0.05
    w /= w.sum(axis=1, keepdims=True)
    target = a * w[:, :1] + b * w[:, 1:2] + c * w[:, 2:]
    along = (b - a) * (g.random((n, 1)) * 2 - 1) + (c - a) * (g.random((n, 1)) * 2 - 1)
    along /= np.maximum(np.linalg.norm(along, axis=1, keepdims=True), 1e-30)
    size = np.linalg.norm(verts[:, :3].max(axis=0) - verts[:, :3].min(axis=0))
    o = target - along * (0.05 + g.random((n, 1))) * size * 0.3 + nrm * (g.random((n, 1)) * 2 - 1) * size * 2e-3
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros(n, dtype=abi.RAY_DTYPE)
    r["o"] = o.astype(np.float32)
    r["d"] = synth._unit_f32(d)
    r["tmin"], r["tmax"] = 0.0, np.inf
    return r

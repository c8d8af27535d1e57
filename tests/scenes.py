"""Small seeded scenes shared by the CPU and GPU parity tests (inputs only, no expectations)."""
import os

import numpy as np

from pathtracer_amd import abi, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rotation(axis, deg):
    axis = np.asarray(axis, dtype=np.float64)
    axis /= np.linalg.norm(axis)
    c, s = np.cos(np.radians(deg)), np.sin(np.radians(deg))
    x, y, z = axis
    R = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                  [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                  [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]])
    return R


def trs(translate=(0, 0, 0), axis=(0, 1, 0), deg=0.0, scale=(1, 1, 1)):
    """Column-major 4x4 (16 floats) = translate * rotate * scale."""
    M = np.eye(4)
    M[:3, :3] = rotation(axis, deg) @ np.diag(scale)
    M[:3, 3] = translate
    return M.T.astype(np.float32).reshape(16)   # column-major: element [col*4+row]


def quad(y=0.0, half=1.0):
    """Two triangles in the plane y = const, normals +y."""
    v = np.zeros((4, 8), dtype=np.float32)
    v[:, :3] = [[-half, y, -half], [-half, y, half], [half, y, -half], [half, y, half]]
    v[:, 4] = 1.0
    v[:, 6:] = [[0, 0], [0, 1], [1, 0], [1, 1]]
    return v, np.array([[0, 1, 2], [2, 1, 3]], dtype=np.uint32)


def icosphere(subdiv=2, radius=1.0):
    t = (1 + 5 ** 0.5) / 2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    f = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
         [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]]
    v = list(v / np.linalg.norm(v, axis=1, keepdims=True))
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        f = nf
    v = np.array(v)
    verts = np.zeros((len(v), 8), dtype=np.float32)
    verts[:, :3] = v * radius
    verts[:, 3:6] = v
    verts[:, 6] = 0.5 + np.arctan2(v[:, 2], v[:, 0]) / (2 * np.pi)
    verts[:, 7] = 0.5 + np.arcsin(np.clip(v[:, 1], -1, 1)) / np.pi
    return verts, np.array(f, dtype=np.uint32)


def instanced_scene(use_bvh=True, with_spheres=True, mesh_bvh=True, fresnel_mode=abi.PTX_FRESNEL_GCC_INT_ABS):
    """Boxy room of quads, rotated/non-uniformly scaled icospheres, a soup, analytic spheres."""
    sb = abi.SceneBuilder(use_bvh=use_bvh, fresnel_mode=fresnel_mode)
    white = sb.add_constant_texture((0.8, 0.8, 0.8))
    red = sb.add_constant_texture((0.8, 0.2, 0.2))
    light = sb.add_constant_texture((1.0, 1.0, 1.0), scale=4.0)
    m_white = sb.add_material(abi.PTX_MAT_LAMBERTIAN, white)
    m_red = sb.add_material(abi.PTX_MAT_LAMBERTIAN, red)
    m_mirror = sb.add_material(abi.PTX_MAT_MIRROR, white)
    m_glass = sb.add_material(abi.PTX_MAT_GLASS, white, white, 1.5)
    m_refract = sb.add_material(abi.PTX_MAT_REFRACT, -1, white, 1.3)
    m_light = sb.add_material(abi.PTX_MAT_EMISSIVE, light)
    q = sb.add_mesh(*quad(), use_bvh=mesh_bvh)
    ico = sb.add_mesh(*icosphere(2), use_bvh=mesh_bvh)
    soup = sb.add_mesh(*synth.soup(600, key=99, extent=2.0, half=0.15), use_bvh=mesh_bvh)
    sb.add_instance(q, m_white, trs((0, -1, 0), scale=(2, 1, 2)))                       # floor
    sb.add_instance(q, m_white, trs((0, 1, 0), axis=(1, 0, 0), deg=180, scale=(2, 1, 2)))  # ceiling
    sb.add_instance(q, m_red, trs((-2, 0, 0), axis=(0, 0, 1), deg=-90, scale=(1, 1, 2)))
    sb.add_instance(q, m_white, trs((2, 0, 0), axis=(0, 0, 1), deg=90, scale=(1, 1, 2)))
    sb.add_instance(q, m_white, trs((0, 0, -2), axis=(1, 0, 0), deg=90, scale=(2, 1, 1)))
    sb.add_instance(q, m_light, trs((0, 0.98, 0), axis=(1, 0, 0), deg=180, scale=(0.5, 1, 0.5)))
    sb.add_instance(ico, m_mirror, trs((-0.9, -0.5, -0.6), axis=(1, 2, 3), deg=33, scale=(0.5, 0.4, 0.6)))
    sb.add_instance(ico, m_glass, trs((0.8, -0.45, 0.3), scale=(0.55, 0.55, 0.55)))
    sb.add_instance(ico, -1)                                                            # identity transform, default material
    sb.add_instance(soup, m_refract, trs((-1.0, -0.9, -1.0), axis=(0, 1, 0), deg=10))
    if with_spheres:
        sb.add_instance(-1, m_mirror, trs((0.2, 0.4, -0.8), scale=(0.3, 0.3, 0.3)), sphere_radius=1.0)
        sb.add_instance(-1, m_white, abi.IDENTITY16, sphere_radius=0.25)
    return sb


def scene_rays(n, seed=7, extent=2.2, inside=True):
    """Incoherent rays through the room, constructed like Ray's ctor (unit())."""
    g = np.random.Generator(np.random.Philox(key=seed))
    r = np.zeros(n, dtype=abi.RAY_DTYPE)
    r["o"] = (g.random((n, 3), dtype=np.float32) * 2 - 1) * np.float32(extent if not inside else 0.95)
    d = g.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r["d"] = synth._unit_f32(d)
    r["tmin"], r["tmax"] = 0.0, np.inf
    return r


def cow_mesh():
    """(verts, indices) of the reference's cow (media/js3d/cow.js3d: 5,856 triangles, shared vertices, thin
    triangles), taken from the committed scene descriptor of refract_cow_cbox.js3d."""
    g = np.load(os.path.join(GOLDEN, "scene_refract_cow_cbox.npz"), allow_pickle=False)
    cow_i = max(range(int(g["n_meshes"])), key=lambda i: len(g[f"mesh{i}_indices"]))
    return g[f"mesh{cow_i}_verts"], g[f"mesh{cow_i}_indices"]


def cow_grid_transforms(n):
    """local_to_world of the n^3 cows of the grid scenes, in the order the instances are added."""
    return [trs((ix * 1.2, iy * 1.2, iz * 1.2), axis=(0, 1, 0), deg=17.0 * (ix + 2 * iy + 3 * iz), scale=(0.9, 0.9, 0.9))
            for ix in range(n) for iy in range(n) for iz in range(n)]


def _cow_grid_room(sb, n, q, white):
    """Floor, lamp and camera shared by the instanced and the flattened cow grid."""
    lamp = sb.add_material(abi.PTX_MAT_EMISSIVE, sb.add_constant_texture((1, 1, 1), 6.0))
    L = n * 1.2
    sb.add_instance(q, white, trs((L / 2, -0.8, L / 2), scale=(L, 1, L)))
    sb.add_instance(q, lamp, trs((L / 2, L + 1.0, L / 2), axis=(1, 0, 0), deg=180, scale=(L / 2, 1, L / 2)))
    cam = abi.PtxCamera()
    cam.vertical_fov = 50.0
    T = trs((L / 2, L * 0.7, L * 2.4), axis=(1, 0, 0), deg=-12)
    for i in range(16):
        cam.camera_to_world[i] = float(T[i])
    return cam


def instanced_cows(n, two_tone=True):
    """n^3 rotated instances of the cow mesh on a grid (10.1 M instanced triangles at n = 12), Lambertian (alternating
    white and red, or all white), over a floor quad and under one emissive quad; returns (SceneBuilder, PtxCamera
    without film settings)."""
    sb = abi.SceneBuilder()
    cow = sb.add_mesh(*cow_mesh())
    q = sb.add_mesh(*quad())
    white = sb.add_material(abi.PTX_MAT_LAMBERTIAN, sb.add_constant_texture((0.7, 0.7, 0.7)))
    red = sb.add_material(abi.PTX_MAT_LAMBERTIAN, sb.add_constant_texture((0.7, 0.3, 0.3)))
    for k, T in enumerate(cow_grid_transforms(n)):
        ix, iy, iz = k // (n * n), (k // n) % n, k % n
        sb.add_instance(cow, white if (ix + iy + iz) % 2 or not two_tone else red, T)
    cam = _cow_grid_room(sb, n, q, white)
    return sb, cam


def flattened_cows(n):
    """The cow grid of instanced_cows(n) baked into ONE mesh in world space (10,119,168 triangles and 30,357,504
    vertices at n = 12: about 215 MB of tree nodes, 486 MB of triangle geometry and 607 MB of attributes on the GPU --
    a tree that lives in HBM, not in the caches).  Vertex k of cow c is vertex c*17568 + k: position = the instance's
    float32 local_to_world applied in float64 and rounded to float32 once, normal = its rotation applied likewise.
    Returns (verts, indices)."""
    cv, ci = cow_mesh()
    Ts = cow_grid_transforms(n)
    nv, nt = len(cv), len(ci)
    verts = np.empty((len(Ts) * nv, 8), dtype=np.float32)
    idx = np.empty((len(Ts) * nt, 3), dtype=np.uint32)
    p64, n64 = cv[:, :3].astype(np.float64), cv[:, 3:6].astype(np.float64)
    for c, T in enumerate(Ts):
        M = T.astype(np.float64).reshape(4, 4).T          # column-major 16 floats -> row-major 4x4
        out = verts[c * nv:(c + 1) * nv]
        out[:, :3] = (p64 @ M[:3, :3].T + M[:3, 3]).astype(np.float32)
        out[:, 3:6] = (n64 @ (M[:3, :3] / 0.9).T).astype(np.float32)
        out[:, 6:] = cv[:, 6:]
        idx[c * nt:(c + 1) * nt] = ci + np.uint32(c * nv)
    return verts, idx


def flattened_cow_scene(n, verts=None, idx=None, nodes=None):
    """The flattened cow grid as a lit scene: the one big mesh under an identity instance (Lambertian), the floor quad
    and the emissive quad of instanced_cows; returns (SceneBuilder, PtxCamera without film settings)."""
    if verts is None:
        verts, idx = flattened_cows(n)
    sb = abi.SceneBuilder()
    big = sb.add_mesh(verts, idx, nodes=nodes)
    q = sb.add_mesh(*quad())
    white = sb.add_material(abi.PTX_MAT_LAMBERTIAN, sb.add_constant_texture((0.7, 0.7, 0.7)))
    sb.add_instance(big, white)
    cam = _cow_grid_room(sb, n, q, white)
    return sb, cam


# ---- adversarial rays for the culled traversal order: rays that lie (almost) in the plane of a triangle they pass
def skimming_rays(verts, idx, n, seed, lift):
    """From just off a random triangle's plane (`lift` x the mesh's size) towards a point of that triangle, along the plane."""
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
    o = target - along * (0.05 + g.random((n, 1))) * size * 0.3 + nrm * (g.random((n, 1)) * 2 - 1) * size * lift
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros(n, dtype=abi.RAY_DTYPE)
    r["o"] = o.astype(np.float32)
    r["d"] = synth._unit_f32(d)
    r["tmin"], r["tmax"] = 0.0, np.inf
    return r


def slivers(n, key):
    """Long thin triangles (aspect 1:200) in random orientations inside a unit cube: many overlap along any line."""
    g = np.random.Generator(np.random.Philox(key=key))
    c = g.random((n, 3))
    u = g.standard_normal((n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = np.cross(u, g.standard_normal((n, 3)))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    p = np.stack([c - u * 0.1, c + u * 0.1, c + v * 0.001], axis=1).reshape(-1, 3)
    verts = np.zeros((3 * n, 8), dtype=np.float32)
    verts[:, :3] = p
    verts[:, 4] = 1.0
    return verts, np.arange(3 * n, dtype=np.uint32).reshape(n, 3)

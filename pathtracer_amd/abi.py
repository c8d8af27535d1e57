"""ctypes mirror of include/ptx.h: the POD structs and constants of the C ABI.

Only layouts live here; loading the library and calling it is in `capi.py`.
"""
import ctypes as C

import numpy as np

PTX_OK = 0
PTX_NO_HIT = 0xFFFFFFFF

PTX_MEM_DEVICE = 1 << 0
PTX_ASYNC = 1 << 1
PTX_TRAVERSE_REFERENCE = 0 << 4
PTX_TRAVERSE_CULLED = 1 << 4
PTX_NO_PERSISTENT = 1 << 8
PTX_RAYS_COHERENT = 1 << 9
PTX_RENDER_REFERENCE_ORDER = 1 << 10
PTX_RENDER_TRACE_UNUSED_RAYS = 1 << 11

PTX_MAT_LAMBERTIAN, PTX_MAT_MIRROR, PTX_MAT_REFRACT, PTX_MAT_GLASS, PTX_MAT_EMISSIVE = range(5)
PTX_TEX_CONSTANT, PTX_TEX_IMAGE = range(2)
PTX_SAMPLER_NEAREST, PTX_SAMPLER_BILINEAR = range(2)
PTX_LIGHT_POINT, PTX_LIGHT_DIRECTIONAL, PTX_LIGHT_SPOT = range(3)
PTX_ENV_HEMISPHERE, PTX_ENV_SPHERE = range(2)
PTX_FRESNEL_GCC_INT_ABS, PTX_FRESNEL_FLOAT_ABS = range(2)

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)

# numpy views of the array-of-struct buffers
RAY_DTYPE = np.dtype([("o", "<f4", 3), ("d", "<f4", 3), ("tmin", "<f4"), ("tmax", "<f4")])
HIT_DTYPE = np.dtype([("distance", "<f4"), ("prim", "<u4")])
NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("start", "<u4"), ("size", "<u4"), ("l", "<u4"), ("r", "<u4")])
TILE_DTYPE = np.dtype([("seed", "<u4"), ("x_begin", "<u4"), ("x_end", "<u4"), ("y_begin", "<u4"), ("y_end", "<u4"),
                       ("s_begin", "<u4"), ("s_end", "<u4")])
assert RAY_DTYPE.itemsize == 32 and HIT_DTYPE.itemsize == 8 and NODE_DTYPE.itemsize == 40 and TILE_DTYPE.itemsize == 28


class PtxRay(C.Structure):
    _fields_ = [("o", C.c_float * 3), ("d", C.c_float * 3), ("tmin", C.c_float), ("tmax", C.c_float)]


class PtxHit(C.Structure):
    _fields_ = [("distance", C.c_float), ("prim", C.c_uint32)]


class PtxBvhNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("start", C.c_uint32), ("size", C.c_uint32),
                ("l", C.c_uint32), ("r", C.c_uint32)]


class PtxMeshDesc(C.Structure):
    _fields_ = [("verts", f32p), ("n_verts", C.c_uint32), ("indices", u32p), ("n_tris", C.c_uint32),
                ("use_bvh", C.c_uint32), ("max_leaf_size", C.c_uint32), ("nodes", C.POINTER(PtxBvhNode)),
                ("n_nodes", C.c_uint32)]


class PtxTextureDesc(C.Structure):
    _fields_ = [("type", C.c_uint32), ("color", C.c_float * 3), ("scale", C.c_float), ("sampler", C.c_uint32),
                ("w", C.c_uint32), ("h", C.c_uint32), ("pixels", f32p)]


class PtxMaterialDesc(C.Structure):
    _fields_ = [("type", C.c_uint32), ("tex_a", C.c_int32), ("tex_b", C.c_int32), ("ior", C.c_float)]


class PtxInstanceDesc(C.Structure):
    _fields_ = [("mesh", C.c_int32), ("sphere_radius", C.c_float), ("material", C.c_int32), ("T", C.c_float * 16)]


class PtxDeltaLightDesc(C.Structure):
    _fields_ = [("type", C.c_uint32), ("color", C.c_float * 3), ("intensity", C.c_float), ("inner_angle", C.c_float),
                ("outer_angle", C.c_float), ("T", C.c_float * 16)]


class PtxEnvLightDesc(C.Structure):
    _fields_ = [("type", C.c_uint32), ("tex", C.c_int32)]


class PtxSceneDesc(C.Structure):
    _fields_ = [("meshes", C.POINTER(PtxMeshDesc)), ("n_meshes", C.c_uint32),
                ("instances", C.POINTER(PtxInstanceDesc)), ("n_instances", C.c_uint32),
                ("textures", C.POINTER(PtxTextureDesc)), ("n_textures", C.c_uint32),
                ("materials", C.POINTER(PtxMaterialDesc)), ("n_materials", C.c_uint32),
                ("delta_lights", C.POINTER(PtxDeltaLightDesc)), ("n_delta_lights", C.c_uint32),
                ("env_lights", C.POINTER(PtxEnvLightDesc)), ("n_env_lights", C.c_uint32),
                ("use_bvh", C.c_uint32), ("fresnel_mode", C.c_uint32)]


class PtxCamera(C.Structure):
    _fields_ = [("vertical_fov", C.c_float), ("aspect_ratio", C.c_float), ("film_width", C.c_uint32),
                ("film_height", C.c_uint32), ("film_samples", C.c_uint32), ("film_max_ray_depth", C.c_uint32),
                ("camera_to_world", C.c_float * 16)]


class PtxTile(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("x_begin", C.c_uint32), ("x_end", C.c_uint32), ("y_begin", C.c_uint32),
                ("y_end", C.c_uint32), ("s_begin", C.c_uint32), ("s_end", C.c_uint32)]


class PtxCounters(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_any", C.c_uint64), ("paths", C.c_uint64),
                ("render_main_rays", C.c_uint64), ("render_direct_rays", C.c_uint64)]


class PtxFrameReduceInfo(C.Structure):
    _fields_ = [("parts_local", C.c_uint32), ("parts_peer_direct", C.c_uint32), ("parts_staged", C.c_uint32), ("add_launches", C.c_uint32),
                ("allocations", C.c_uint32), ("pad", C.c_uint32), ("bytes_over_links", C.c_uint64)]


IDENTITY16 = (1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0)


def fptr(a):
    """float* into a C-contiguous float32 numpy array (kept alive by the caller)."""
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(f32p)


def uptr(a):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u32p)


class SceneBuilder:
    """Assembles a PtxSceneDesc from numpy arrays and keeps every buffer alive.

    The same object feeds the product library and, in tests, the CPU checker: both
    read the POD description declared in include/ptx.h.
    """

    def __init__(self, use_bvh=True, fresnel_mode=PTX_FRESNEL_GCC_INT_ABS):
        self.meshes, self.instances, self.textures, self.materials = [], [], [], []
        self.delta_lights, self.env_lights = [], []
        self.use_bvh, self.fresnel_mode = use_bvh, fresnel_mode
        self._keep = []

    def mesh_desc(self, verts, indices, use_bvh=True, max_leaf_size=0, nodes=None):
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 8)
        indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        self._keep += [verts, indices]
        d = PtxMeshDesc(fptr(verts), len(verts), uptr(indices), len(indices), 1 if use_bvh else 0, max_leaf_size, None, 0)
        if nodes is not None:
            nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
            self._keep.append(nodes)
            d.nodes = nodes.ctypes.data_as(C.POINTER(PtxBvhNode))
            d.n_nodes = len(nodes)
        return d

    def add_mesh(self, verts, indices, use_bvh=True, max_leaf_size=0, nodes=None):
        self.meshes.append(self.mesh_desc(verts, indices, use_bvh, max_leaf_size, nodes))
        return len(self.meshes) - 1

    def add_constant_texture(self, color, scale=1.0):
        self.textures.append(PtxTextureDesc(PTX_TEX_CONSTANT, (C.c_float * 3)(*color), scale, 0, 0, 0, None))
        return len(self.textures) - 1

    def add_image_texture(self, pixels, sampler=PTX_SAMPLER_BILINEAR):
        pixels = np.ascontiguousarray(pixels, dtype=np.float32)
        h, w, _ = pixels.shape
        self._keep.append(pixels)
        self.textures.append(PtxTextureDesc(PTX_TEX_IMAGE, (C.c_float * 3)(0, 0, 0), 1.0, sampler, w, h, fptr(pixels)))
        return len(self.textures) - 1

    def add_material(self, mtype, tex_a=-1, tex_b=-1, ior=1.5):
        self.materials.append(PtxMaterialDesc(mtype, tex_a, tex_b, ior))
        return len(self.materials) - 1

    def add_instance(self, mesh, material, T=IDENTITY16, sphere_radius=1.0):
        T = np.asarray(T, dtype=np.float32).reshape(16)
        self.instances.append(PtxInstanceDesc(mesh, sphere_radius, material, (C.c_float * 16)(*T)))
        return len(self.instances) - 1

    def add_delta_light(self, ltype, color, intensity, T=IDENTITY16, inner=30.0, outer=45.0):
        T = np.asarray(T, dtype=np.float32).reshape(16)
        self.delta_lights.append(PtxDeltaLightDesc(ltype, (C.c_float * 3)(*color), intensity, inner, outer, (C.c_float * 16)(*T)))

    def add_env_light(self, etype, tex):
        self.env_lights.append(PtxEnvLightDesc(etype, tex))

    def desc(self):
        def arr(t, items):
            a = (t * max(1, len(items)))(*items)
            self._keep.append(a)
            return a

        d = PtxSceneDesc()
        d.meshes, d.n_meshes = arr(PtxMeshDesc, self.meshes), len(self.meshes)
        d.instances, d.n_instances = arr(PtxInstanceDesc, self.instances), len(self.instances)
        d.textures, d.n_textures = arr(PtxTextureDesc, self.textures), len(self.textures)
        d.materials, d.n_materials = arr(PtxMaterialDesc, self.materials), len(self.materials)
        d.delta_lights, d.n_delta_lights = arr(PtxDeltaLightDesc, self.delta_lights), len(self.delta_lights)
        d.env_lights, d.n_env_lights = arr(PtxEnvLightDesc, self.env_lights), len(self.env_lights)
        d.use_bvh, d.fresnel_mode = (1 if self.use_bvh else 0), self.fresnel_mode
        self._keep.append(d)
        return d


def desc_to_arrays(desc, cameras=None):
    """Flattens a PtxSceneDesc (and optionally {name: PtxCamera}) into a dict of numpy arrays (np.savez-able)."""
    out = {"n_meshes": np.uint32(desc.n_meshes), "use_bvh": np.uint32(desc.use_bvh), "fresnel_mode": np.uint32(desc.fresnel_mode)}
    for i in range(desc.n_meshes):
        m = desc.meshes[i]
        out[f"mesh{i}_verts"] = np.ctypeslib.as_array(m.verts, shape=(m.n_verts, 8)).copy() if m.n_verts else np.zeros((0, 8), np.float32)
        out[f"mesh{i}_indices"] = np.ctypeslib.as_array(m.indices, shape=(m.n_tris, 3)).copy() if m.n_tris else np.zeros((0, 3), np.uint32)
        out[f"mesh{i}_opts"] = np.array([m.use_bvh, m.max_leaf_size], dtype=np.uint32)
    out["instances"] = np.array([[in_.mesh, in_.material] for in_ in desc.instances[:desc.n_instances]], dtype=np.int32).reshape(-1, 2)
    out["instance_radius"] = np.array([in_.sphere_radius for in_ in desc.instances[:desc.n_instances]], dtype=np.float32)
    out["instance_T"] = np.array([list(in_.T) for in_ in desc.instances[:desc.n_instances]], dtype=np.float32).reshape(-1, 16)
    tex = desc.textures[:desc.n_textures]
    out["textures"] = np.array([[t.type, t.sampler, t.w, t.h] for t in tex], dtype=np.uint32).reshape(-1, 4)
    out["texture_color"] = np.array([[t.color[0], t.color[1], t.color[2], t.scale] for t in tex], dtype=np.float32).reshape(-1, 4)
    for i, t in enumerate(tex):
        if t.type == PTX_TEX_IMAGE:
            out[f"tex{i}_pixels"] = np.ctypeslib.as_array(t.pixels, shape=(t.h, t.w, 3)).copy()
    mats = desc.materials[:desc.n_materials]
    out["materials"] = np.array([[m.type, m.tex_a, m.tex_b] for m in mats], dtype=np.int32).reshape(-1, 3)
    out["material_ior"] = np.array([m.ior for m in mats], dtype=np.float32)
    dl = desc.delta_lights[:desc.n_delta_lights]
    out["delta_light_type"] = np.array([l.type for l in dl], dtype=np.uint32)
    out["delta_light_params"] = np.array([[l.color[0], l.color[1], l.color[2], l.intensity, l.inner_angle, l.outer_angle] for l in dl], dtype=np.float32).reshape(-1, 6)
    out["delta_light_T"] = np.array([list(l.T) for l in dl], dtype=np.float32).reshape(-1, 16)
    out["env_lights"] = np.array([[l.type, l.tex] for l in desc.env_lights[:desc.n_env_lights]], dtype=np.int32).reshape(-1, 2)
    if cameras:
        names = sorted(cameras)
        out["camera_names"] = np.array(names)
        out["camera_lens"] = np.array([[cameras[n].vertical_fov, cameras[n].aspect_ratio] for n in names], dtype=np.float32)
        out["camera_film"] = np.array([[cameras[n].film_width, cameras[n].film_height, cameras[n].film_samples, cameras[n].film_max_ray_depth] for n in names], dtype=np.uint32)
        out["camera_T"] = np.array([list(cameras[n].camera_to_world) for n in names], dtype=np.float32)
    return out


def builder_from_arrays(a):
    """Inverse of desc_to_arrays: (SceneBuilder, {name: PtxCamera})."""
    sb = SceneBuilder(use_bvh=bool(a["use_bvh"]), fresnel_mode=int(a["fresnel_mode"]))
    for i in range(int(a["n_meshes"])):
        opts = a[f"mesh{i}_opts"]
        sb.add_mesh(a[f"mesh{i}_verts"], a[f"mesh{i}_indices"], use_bvh=bool(opts[0]), max_leaf_size=int(opts[1]))
    for i, (t, c) in enumerate(zip(a["textures"], a["texture_color"])):
        if int(t[0]) == PTX_TEX_IMAGE:
            sb.add_image_texture(a[f"tex{i}_pixels"], int(t[1]))
        else:
            sb.add_constant_texture(tuple(float(x) for x in c[:3]), float(c[3]))
    for m, ior in zip(a["materials"], a["material_ior"]):
        sb.add_material(int(m[0]), int(m[1]), int(m[2]), float(ior))
    for (mesh, mat), radius, T in zip(a["instances"], a["instance_radius"], a["instance_T"]):
        sb.add_instance(int(mesh), int(mat), T, float(radius))
    for t, p, T in zip(a["delta_light_type"], a["delta_light_params"], a["delta_light_T"]):
        sb.add_delta_light(int(t), tuple(float(x) for x in p[:3]), float(p[3]), T, float(p[4]), float(p[5]))
    for t, tex in a["env_lights"]:
        sb.add_env_light(int(t), int(tex))
    cams = {}
    if "camera_names" in a:
        for n, lens, film, T in zip(a["camera_names"], a["camera_lens"], a["camera_film"], a["camera_T"]):
            cam = PtxCamera()
            cam.vertical_fov, cam.aspect_ratio = float(lens[0]), float(lens[1])
            cam.film_width, cam.film_height, cam.film_samples, cam.film_max_ray_depth = (int(x) for x in film)
            for k in range(16):
                cam.camera_to_world[k] = float(T[k])
            cams[str(n)] = cam
    return sb, cams

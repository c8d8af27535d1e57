"""Loader and thin Python mirror of the C ABI in include/ptx.h (libptx_amd.so).

There is no CPU fallback: if the library is missing, or a call fails, this module raises.
Arrays are numpy (host pointers) or torch CUDA tensors (device pointers, PTX_MEM_DEVICE).
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTX_AMD_LIB") or os.path.join(_HERE, "lib", "libptx_amd.so")   # the override serves tuning experiments

vp = C.c_void_p
_lib = None


class PtxError(RuntimeError):
    pass


def lib():
    """The loaded C ABI library.  Raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PtxError(f"{LIB_PATH} is missing: build it with `make -C pathtracer_amd/csrc` (there is no CPU fallback)")
    # One HIP/HSA runtime per process: torch ships its own copy of the ROCm runtime libraries, and a
    # process that loads the system copy first and torch's second ends up with two HSA runtimes,
    # the second of which sees no GPU.  Loading torch first (when it is installed) makes this
    # library resolve its libamdhip64 dependency to the copy torch already mapped.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.ptx_last_error.restype = C.c_char_p
    L.ptx_version.restype = C.c_char_p
    L.ptx_device_count.restype = C.c_int
    L.ptx_bvh_build.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.ptx_bvh_build_triangles.argtypes = [vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.ptx_bvh_build_gpu.argtypes = [C.c_int, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.ptx_bvh_build_triangles_gpu.argtypes = [C.c_int, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.ptx_bvh_built_on_device.argtypes = [vp]
    for f in ("ptx_bvh_num_nodes", "ptx_bvh_num_prims", "ptx_bvh_max_depth"):
        getattr(L, f).restype = C.c_uint32
        getattr(L, f).argtypes = [vp]
    L.ptx_bvh_get_nodes.argtypes = [vp, vp]
    L.ptx_bvh_get_order.argtypes = [vp, vp]
    L.ptx_bvh_get_stats.argtypes = [vp, vp]
    L.ptx_bvh_destroy.argtypes = [vp]
    L.ptx_bvh_destroy.restype = None
    L.ptx_scene_create.argtypes = [C.c_int, C.POINTER(abi.PtxSceneDesc), C.POINTER(vp)]
    L.ptx_scene_create_mesh.argtypes = [C.c_int, C.POINTER(abi.PtxMeshDesc), C.POINTER(vp)]
    L.ptx_scene_destroy.argtypes = [vp]
    L.ptx_scene_destroy.restype = None
    L.ptx_scene_set_stream.argtypes = [vp, vp]
    L.ptx_scene_mesh_num_nodes.restype = C.c_uint32
    L.ptx_scene_mesh_num_nodes.argtypes = [vp, C.c_uint32]
    L.ptx_scene_mesh_get_nodes.argtypes = [vp, C.c_uint32, vp]
    L.ptx_scene_mesh_get_order.argtypes = [vp, C.c_uint32, vp]
    L.ptx_scene_mesh_built_on_device.argtypes = [vp, C.c_uint32]
    L.ptx_scene_get_instance_order.argtypes = [vp, vp]
    L.ptx_trace_closest.argtypes = [vp, vp, C.c_uint64, vp, vp, vp, C.c_uint32]
    L.ptx_trace_closest_noalloc.argtypes = [vp, vp, C.c_uint64, vp, vp, vp, C.c_uint32]
    L.ptx_noalloc_max_rays.restype = C.c_uint64
    L.ptx_noalloc_max_rays.argtypes = []
    L.ptx_trace_any.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32]
    L.ptx_trace_closest_stats.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32]
    L.ptx_selftest_division.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    L.ptx_selftest_philox.argtypes = [C.c_int, vp, vp, C.c_uint32, vp]
    L.ptx_selftest_path_units.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.ptx_render_tiles.argtypes = [vp, C.POINTER(abi.PtxCamera), vp, C.c_uint32, C.c_uint64, vp, vp, C.c_uint32]
    L.ptx_make_tiles.restype = C.c_uint32
    L.ptx_make_tiles.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint32]
    L.ptx_accumulator_to_image.argtypes = [vp, vp, C.c_uint64, vp]
    L.ptx_tonemap.argtypes = [vp, C.c_uint64, C.c_float, vp]
    L.ptx_get_counters.argtypes = [vp, C.POINTER(abi.PtxCounters)]
    L.ptx_frame_create.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.ptx_frame_accum.restype = vp
    L.ptx_frame_accum.argtypes = [vp]
    L.ptx_frame_samples.restype = vp
    L.ptx_frame_samples.argtypes = [vp]
    L.ptx_frame_upload.argtypes = [vp, vp, vp]
    L.ptx_frame_download.argtypes = [vp, vp, vp]
    L.ptx_frame_clear.argtypes = [vp]
    L.ptx_frame_reduce.argtypes = [vp, C.POINTER(vp), C.c_uint32]
    L.ptx_frame_reduce_info.argtypes = [vp, C.POINTER(abi.PtxFrameReduceInfo)]
    L.ptx_frame_destroy.argtypes = [vp]
    L.ptx_frame_destroy.restype = None
    L.ptx_comm_create.argtypes = [C.POINTER(C.c_int), C.c_uint32, C.POINTER(vp)]
    L.ptx_comm_size.restype = C.c_uint32
    L.ptx_comm_size.argtypes = [vp]
    L.ptx_comm_library.restype = C.c_char_p
    L.ptx_comm_library.argtypes = [vp]
    L.ptx_comm_destroy.argtypes = [vp]
    L.ptx_comm_destroy.restype = None
    L.ptx_frame_reduce_rccl.argtypes = [vp, vp, C.POINTER(vp), C.c_uint32]
    L.ptx_frame_allreduce.argtypes = [vp, C.POINTER(vp), C.c_uint32]
    L.ptx_get_kernel_times.argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    L.ptx_js3d_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.ptx_js3d_free.argtypes = [vp]
    L.ptx_js3d_free.restype = None
    L.ptx_js3d_scene_desc.argtypes = [vp]
    L.ptx_js3d_scene_desc.restype = C.POINTER(abi.PtxSceneDesc)
    L.ptx_js3d_set_modes.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.ptx_js3d_set_modes.restype = None
    L.ptx_js3d_num_cameras.argtypes = [vp]
    L.ptx_js3d_num_cameras.restype = C.c_uint32
    L.ptx_js3d_num_meshes.argtypes = [vp]
    L.ptx_js3d_num_meshes.restype = C.c_uint32
    for f in ("ptx_js3d_camera_name", "ptx_js3d_mesh_name", "ptx_js3d_instance_name"):
        getattr(L, f).argtypes = [vp, C.c_uint32]
        getattr(L, f).restype = C.c_char_p
    L.ptx_js3d_warnings.argtypes = [vp]
    L.ptx_js3d_warnings.restype = C.c_char_p
    L.ptx_js3d_get_camera.argtypes = [vp, C.c_char_p, C.POINTER(abi.PtxCamera)]
    L.ptx_write_png.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32, C.c_int]
    _lib = L
    return L


def check(rc):
    if rc != abi.PTX_OK:
        raise PtxError(f"ptx error {rc}: {lib().ptx_last_error().decode()}")


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _addr(x):
    """(address, is_device) of a numpy array or a torch tensor; None -> (None, None)."""
    if x is None:
        return None, None
    if _is_torch(x):
        assert x.is_contiguous()
        return x.data_ptr(), x.is_cuda
    assert x.flags["C_CONTIGUOUS"]
    return x.ctypes.data, False


class Bvh:
    """PT::BVH<P>::build (bvh.cpp:31-243) over boxes or over an indexed triangle mesh, by the host builder or
    (gpu_device given) by the GPU builder; both give the same tree."""

    def __init__(self, boxes=None, verts=None, indices=None, max_leaf_size=4, gpu_device=None):
        self.h = vp()
        if boxes is not None:
            boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6)
            if gpu_device is None:
                check(lib().ptx_bvh_build(boxes.ctypes.data, len(boxes), max_leaf_size, C.byref(self.h)))
            else:
                check(lib().ptx_bvh_build_gpu(gpu_device, boxes.ctypes.data, len(boxes), max_leaf_size, C.byref(self.h)))
        else:
            verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 8)
            indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
            if gpu_device is None:
                check(lib().ptx_bvh_build_triangles(verts.ctypes.data, len(verts), indices.ctypes.data, len(indices), max_leaf_size, C.byref(self.h)))
            else:
                check(lib().ptx_bvh_build_triangles_gpu(gpu_device, verts.ctypes.data, len(verts), indices.ctypes.data, len(indices), max_leaf_size, C.byref(self.h)))

    @property
    def built_on_device(self):
        return bool(lib().ptx_bvh_built_on_device(self.h))

    def close(self):
        if self.h:
            lib().ptx_bvh_destroy(self.h)
            self.h = vp()

    __del__ = close

    @property
    def max_depth(self):
        return lib().ptx_bvh_max_depth(self.h)

    def nodes(self):
        out = np.zeros(lib().ptx_bvh_num_nodes(self.h), dtype=abi.NODE_DTYPE)
        check(lib().ptx_bvh_get_nodes(self.h, out.ctypes.data))
        return out

    def order(self):
        out = np.zeros(lib().ptx_bvh_num_prims(self.h), dtype=np.uint32)
        check(lib().ptx_bvh_get_order(self.h, out.ctypes.data))
        return out

    def stats(self):
        out = np.zeros(3, dtype=np.uint64)
        check(lib().ptx_bvh_get_stats(self.h, out.ctypes.data))
        return out


class Scene:
    """A scene resident on one GPU (ptx_scene_t)."""

    def __init__(self, desc=None, mesh_desc=None, device=0, keep=None):
        self.h = vp()
        self._keep = keep
        if mesh_desc is not None:
            check(lib().ptx_scene_create_mesh(device, C.byref(mesh_desc), C.byref(self.h)))
        else:
            check(lib().ptx_scene_create(device, C.byref(desc), C.byref(self.h)))

    @classmethod
    def from_mesh(cls, verts, indices, use_bvh=True, max_leaf_size=0, nodes=None, device=0):
        sb = abi.SceneBuilder()
        return cls(mesh_desc=sb.mesh_desc(verts, indices, use_bvh, max_leaf_size, nodes), device=device, keep=sb)

    def close(self):
        if self.h:
            lib().ptx_scene_destroy(self.h)
            self.h = vp()

    def __del__(self):
        try:
            self.close()
        except TypeError:
            pass   # interpreter shutdown: module globals are already gone and the process's exit frees the scene

    def set_stream(self, stream_handle):
        check(lib().ptx_scene_set_stream(self.h, stream_handle))

    def mesh_nodes(self, mesh=0):
        out = np.zeros(lib().ptx_scene_mesh_num_nodes(self.h, mesh), dtype=abi.NODE_DTYPE)
        check(lib().ptx_scene_mesh_get_nodes(self.h, mesh, out.ctypes.data))
        return out

    def mesh_order(self, n_tris, mesh=0):
        out = np.zeros(n_tris, dtype=np.uint32)
        check(lib().ptx_scene_mesh_get_order(self.h, mesh, out.ctypes.data))
        return out

    def mesh_built_on_device(self, mesh=0):
        return bool(lib().ptx_scene_mesh_built_on_device(self.h, mesh))

    def instance_order(self, n):
        out = np.zeros(n, dtype=np.uint32)
        check(lib().ptx_scene_get_instance_order(self.h, out.ctypes.data))
        return out

    def trace_closest(self, rays, flags=0, want_bary=True, want_inst=True):
        """Host arrays in, host arrays out: dict(hits, bary, inst)."""
        rays = np.ascontiguousarray(rays)
        assert rays.dtype == abi.RAY_DTYPE
        n = len(rays)
        hits = np.zeros(n, dtype=abi.HIT_DTYPE)
        bary = np.zeros((n, 2), dtype=np.float32) if want_bary else None
        inst = np.zeros(n, dtype=np.uint32) if want_inst else None
        check(lib().ptx_trace_closest(self.h, rays.ctypes.data, n, hits.ctypes.data, _addr(bary)[0], _addr(inst)[0], flags))
        return dict(hits=hits, bary=bary, inst=inst)

    def trace_closest_device(self, rays, hits, bary=None, inst=None, flags=0, n=None):
        """Torch CUDA tensors in and out (nothing is copied): rays uint8/float32 view of PtxRay[n], hits of PtxHit[n]."""
        n = n if n is not None else rays.numel() * rays.element_size() // 32
        check(lib().ptx_trace_closest(self.h, rays.data_ptr(), n, hits.data_ptr(), _addr(bary)[0], _addr(inst)[0], flags | abi.PTX_MEM_DEVICE))

    def trace_closest_noalloc(self, rays, hits, bary=None, inst=None, flags=abi.PTX_MEM_DEVICE, n=None):
        """ptx_trace_closest_noalloc: torch CUDA tensors in and out; the library allocates nothing.  `flags` is passed
        as given (PTX_MEM_DEVICE is required by the entry point)."""
        n = n if n is not None else rays.numel() * rays.element_size() // 32
        check(lib().ptx_trace_closest_noalloc(self.h, rays.data_ptr(), n, hits.data_ptr(), _addr(bary)[0], _addr(inst)[0], flags))

    def trace_closest_stats(self, rays, flags=0):
        """Per ray: interior nodes fetched, leaves entered, triangles tested, deepest traversal stack."""
        rays = np.ascontiguousarray(rays)
        assert rays.dtype == abi.RAY_DTYPE
        raw = np.zeros((len(rays), 3), dtype=np.uint32)
        check(lib().ptx_trace_closest_stats(self.h, rays.ctypes.data, len(rays), raw.ctypes.data, flags))
        out = np.zeros((len(rays), 4), dtype=np.uint32)
        out[:, 0], out[:, 1], out[:, 2], out[:, 3] = raw[:, 0] & 0xFFFFFF, raw[:, 1], raw[:, 2], raw[:, 0] >> 24
        return out

    def trace_any(self, rays, flags=0):
        rays = np.ascontiguousarray(rays)
        assert rays.dtype == abi.RAY_DTYPE
        out = np.zeros(len(rays), dtype=np.uint8)
        check(lib().ptx_trace_any(self.h, rays.ctypes.data, len(rays), out.ctypes.data, flags))
        return out

    def trace_any_device(self, rays, occluded, flags=0, n=None):
        n = n if n is not None else rays.numel() * rays.element_size() // 32
        check(lib().ptx_trace_any(self.h, rays.data_ptr(), n, occluded.data_ptr(), flags | abi.PTX_MEM_DEVICE))

    def render_tiles(self, cam, tiles, seed, accum=None, samples=None, flags=0):
        npix = cam.film_width * cam.film_height
        tiles = np.ascontiguousarray(tiles, dtype=abi.TILE_DTYPE)
        if accum is None:
            accum = np.zeros(npix * 3, dtype=np.int64)
            samples = np.zeros(npix, dtype=np.uint32)
        a, dev = _addr(accum)
        if dev:
            flags |= abi.PTX_MEM_DEVICE
        check(lib().ptx_render_tiles(self.h, C.byref(cam), tiles.ctypes.data, len(tiles), seed, a, _addr(samples)[0], flags))
        return accum, samples

    def kernel_times_ms(self, last=64):
        """Device durations (ms) of the last traversal-kernel launches, oldest first (waits for the stream)."""
        out = np.zeros(last, dtype=np.float32)
        n = C.c_uint32()
        check(lib().ptx_get_kernel_times(self.h, out.ctypes.data, last, C.byref(n)))
        return out[:n.value]

    def counters(self):
        c = abi.PtxCounters()
        check(lib().ptx_get_counters(self.h, C.byref(c)))
        return c


class Frame:
    """A film resident on a scene's GPU (ptx_frame_t): Pathtracer::accumulator / accumulator_samples (pathtracer.h:80-83)."""

    def __init__(self, scene, width, height):
        self.h = vp()
        self.scene, self.n_pixels = scene, width * height
        check(lib().ptx_frame_create(scene.h, width, height, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().ptx_frame_destroy(self.h)
            self.h = vp()

    __del__ = close

    def render_tiles(self, cam, tiles, seed, flags=0):
        """ptx_render_tiles into this film (device pointers, nothing crosses PCIe)."""
        tiles = np.ascontiguousarray(tiles, dtype=abi.TILE_DTYPE)
        check(lib().ptx_render_tiles(self.scene.h, C.byref(cam), tiles.ctypes.data, len(tiles), seed, lib().ptx_frame_accum(self.h), lib().ptx_frame_samples(self.h),
                                     flags | abi.PTX_MEM_DEVICE))

    def upload(self, accum, samples):
        accum, samples = np.ascontiguousarray(accum, dtype=np.int64), np.ascontiguousarray(samples, dtype=np.uint32)
        assert accum.size == 3 * self.n_pixels and samples.size == self.n_pixels
        check(lib().ptx_frame_upload(self.h, accum.ctypes.data, samples.ctypes.data))

    def download(self):
        accum, samples = np.zeros(3 * self.n_pixels, dtype=np.int64), np.zeros(self.n_pixels, dtype=np.uint32)
        check(lib().ptx_frame_download(self.h, accum.ctypes.data, samples.ctypes.data))
        return accum, samples

    def clear(self):
        check(lib().ptx_frame_clear(self.h))

    def reduce(self, parts):
        """self += sum of parts (ptx_frame_reduce); returns what the call did (PtxFrameReduceInfo)."""
        arr = (vp * max(1, len(parts)))(*[p.h for p in parts])
        check(lib().ptx_frame_reduce(self.h, arr, len(parts)))
        info = abi.PtxFrameReduceInfo()
        check(lib().ptx_frame_reduce_info(self.h, C.byref(info)))
        return info


class Comm:
    """One RCCL rank per listed GPU of this process (ptx_comm_t)."""

    def __init__(self, devices):
        self.h = vp()
        arr = (C.c_int * len(devices))(*devices)
        check(lib().ptx_comm_create(arr, len(devices), C.byref(self.h)))

    def close(self):
        if self.h:
            lib().ptx_comm_destroy(self.h)
            self.h = vp()

    __del__ = close

    @property
    def size(self):
        return lib().ptx_comm_size(self.h)

    @property
    def library(self):
        return lib().ptx_comm_library(self.h).decode()

    def reduce(self, dst, parts):
        """dst = sum of parts (ncclReduce to the rank on dst's GPU)."""
        arr = (vp * len(parts))(*[p.h for p in parts])
        check(lib().ptx_frame_reduce_rccl(self.h, dst.h, arr, len(parts)))

    def allreduce(self, frames):
        """Every film ends holding the sum of all (ncclAllReduce in place)."""
        arr = (vp * len(frames))(*[f.h for f in frames])
        check(lib().ptx_frame_allreduce(self.h, arr, len(frames)))


class Js3d:
    """A scene file of the reference's JSON format, flattened to a PtxSceneDesc (ptx_js3d_t)."""

    def __init__(self, path):
        self.h = vp()
        check(lib().ptx_js3d_load(os.fsencode(path), C.byref(self.h)))

    def close(self):
        if self.h:
            lib().ptx_js3d_free(self.h)
            self.h = vp()

    __del__ = close

    @property
    def desc(self):
        """PtxSceneDesc whose arrays live as long as this object."""
        return lib().ptx_js3d_scene_desc(self.h).contents

    def set_modes(self, use_bvh=True, fresnel_mode=abi.PTX_FRESNEL_GCC_INT_ABS):
        lib().ptx_js3d_set_modes(self.h, 1 if use_bvh else 0, fresnel_mode)

    def camera_names(self):
        return [lib().ptx_js3d_camera_name(self.h, i).decode() for i in range(lib().ptx_js3d_num_cameras(self.h))]

    def camera(self, name):
        cam = abi.PtxCamera()
        check(lib().ptx_js3d_get_camera(self.h, name.encode(), C.byref(cam)))
        return cam

    def mesh_names(self):
        return [lib().ptx_js3d_mesh_name(self.h, i).decode() for i in range(lib().ptx_js3d_num_meshes(self.h))]

    def instance_names(self):
        return [lib().ptx_js3d_instance_name(self.h, i).decode() for i in range(self.desc.n_instances)]

    def warnings(self):
        return lib().ptx_js3d_warnings(self.h).decode()

    def mesh_arrays(self, i):
        """(verts n x 8 float32, indices m x 3 uint32) of mesh i, copied out."""
        m = self.desc.meshes[i]
        verts = np.ctypeslib.as_array(m.verts, shape=(m.n_verts, 8)).copy() if m.n_verts else np.zeros((0, 8), np.float32)
        idx = np.ctypeslib.as_array(m.indices, shape=(m.n_tris, 3)).copy() if m.n_tris else np.zeros((0, 3), np.uint32)
        return verts, idx

    def scene(self, device=0):
        return Scene(self.desc, device=device, keep=self)


def write_png(path, rgba, width, height, flip_rows=True):
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    assert rgba.size == width * height * 4
    check(lib().ptx_write_png(os.fsencode(path), rgba.ctypes.data, width, height, 1 if flip_rows else 0))


def selftest_division(pairs, seed=1, device=0):
    """Number of operand pairs (of `pairs`) on which the short division form differs from IEEE `/`."""
    bad = C.c_uint64()
    check(lib().ptx_selftest_division(device, pairs, seed, C.byref(bad)))
    return bad.value


def selftest_philox(counters, keys, device=0):
    """Philox-4x32-10 blocks computed on the device: counters (n x 4 uint32), keys (n x 2 uint32) -> n x 4 uint32."""
    counters = np.ascontiguousarray(counters, dtype=np.uint32).reshape(-1, 4)
    keys = np.ascontiguousarray(keys, dtype=np.uint32).reshape(-1, 2)
    out = np.zeros_like(counters)
    check(lib().ptx_selftest_philox(device, counters.ctypes.data, keys.ctypes.data, len(counters), out.ctypes.data))
    return out


def selftest_path_units(seed, pixel, sample, n, device=0):
    out = np.zeros(n, dtype=np.float32)
    check(lib().ptx_selftest_path_units(device, seed, pixel, sample, n, out.ctypes.data))
    return out


def make_tiles(w, h, samples, fixed_seed):
    n = lib().ptx_make_tiles(w, h, samples, fixed_seed, None, 0)
    out = np.zeros(n, dtype=abi.TILE_DTYPE)
    lib().ptx_make_tiles(w, h, samples, fixed_seed, out.ctypes.data, n)
    return out


def accumulator_to_image(accum, samples):
    accum = np.ascontiguousarray(accum, dtype=np.int64)
    samples = np.ascontiguousarray(samples, dtype=np.uint32)
    rgb = np.zeros((len(samples), 3), dtype=np.float32)
    check(lib().ptx_accumulator_to_image(accum.ctypes.data, samples.ctypes.data, len(samples), rgb.ctypes.data))
    return rgb


def tonemap(rgb, exposure=1.0):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32).reshape(-1, 3)
    out = np.zeros((len(rgb), 4), dtype=np.uint8)
    check(lib().ptx_tonemap(rgb.ctypes.data, len(rgb), exposure, out.ctypes.data))
    return out

/* ptx.h -- C ABI of the MI355X path-tracing hot path (libptx_amd.so).
 *
 * The reference (hazard-10/PathTracer, a Scotty3D derivative) has no plugin or FFI
 * layer: its boundary is the C++ class API PT::Pathtracer / PT::BVH / PT::Tri_Mesh.
 * This header is the flat boundary those classes' GPU-backed equivalents call; each
 * entry point cites the reference interface it stands in for (paths relative to the
 * reference checkout).  Everything is POD: plain pointers and sizes, no C++ types.
 *
 * Conventions
 *  - Every function returning int returns PTX_OK (0) or a negative PtxStatus; the
 *    message for the calling thread's last failure is ptx_last_error().  Nothing
 *    throws, aborts or exits across this boundary (the reference's assert/die() do,
 *    lib/log.h:45-56).
 *  - The caller owns every buffer it passes and may free it when the call returns.
 *    The library owns device memory reachable from a handle until *_destroy.
 *  - Output buffers are caller-allocated.  Pointers are host pointers unless the
 *    call's `flags` carry PTX_MEM_DEVICE, in which case rays/hits/accumulators are
 *    device pointers on the scene's device and nothing is copied.  Host ray batches of
 *    8 M rays and more are staged in chunks of about 4 M rays, the upload of one chunk
 *    beside the traversal of the one before and the download of the one before that;
 *    page-locked host buffers (hipHostMalloc, hipHostRegister) transfer fastest.
 *  - Calls are synchronous with respect to the host unless PTX_ASYNC is set, in
 *    which case work is only enqueued on the scene's stream (device pointers only).
 *  - A handle may be used by one host thread at a time; use one handle per GPU.
 *  - Matrices are 16 floats, column-major (element [col*4 + row]), as lib/mat4.h.
 */
#ifndef PTX_AMD_H
#define PTX_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum PtxStatus {
	PTX_OK = 0,
	PTX_ERR_INVALID = -1,      /* bad argument (null pointer, index out of range, bad enum) */
	PTX_ERR_DEVICE = -2,       /* a HIP call failed; message holds hipGetErrorString */
	PTX_ERR_NO_DEVICE = -3,    /* no GPU visible / device index out of range */
	PTX_ERR_UNSUPPORTED = -4,  /* valid input the GPU path cannot run (tree too deep, leaf too large) */
	PTX_ERR_OOM = -5
} PtxStatus;

/* ------------------------------------------------------------------ flags */
enum {
	PTX_MEM_DEVICE = 1u << 0,       /* ray / hit / accumulator pointers are device pointers */
	PTX_ASYNC = 1u << 1,            /* enqueue only; caller synchronises the stream */
	/* Traversal semantics (closest-hit):
	 * REFERENCE visits exactly the nodes PT::BVH::hitRecursive visits (bvh.cpp:272-375): every
	 *   node whose ancestor chain passes BBox::hit(ray,(0,inf)), left before right, no culling
	 *   by the running closest distance.  Identical to the reference by construction.
	 * CULLED visits children near-to-far and skips a subtree whose slab entry is beyond the
	 *   running closest distance by more than a guard band (1 % of that distance); ties are still resolved as the
	 *   reference resolves them (largest position in the BVH-ordered primitive array wins,
	 *   trace.h:28-36).  Same answers as REFERENCE wherever the distances Triangle::hit computes mean
	 *   something: its distance carries a relative error of about 2^-22 over the sine of the angle between
	 *   ray and triangle plane, and the guard band covers angles down to 0.002 degrees.  Where a ray lies IN
	 *   the plane of a triangle it passes (to within about 1e-6 rad) all of Triangle::hit's determinants are
	 *   rounding noise, the reference may accept the triangle at a noise distance, and only a walk that tests
	 *   every triangle the reference tests reproduces that: REFERENCE does, CULLED may skip the triangle.
	 *   Measured on adversarial ray sets and on 16.7 M random rays per benchmark run (0 differences there).
	 *   A ray with a NaN or infinite component, or whose hit distance comes out NaN, is walked in
	 *   the reference order whatever this flag says: with NaN distances Trace::min's winner is a
	 *   matter of test order.
	 */
	PTX_TRAVERSE_REFERENCE = 0u << 4,
	PTX_TRAVERSE_CULLED = 1u << 4,
	PTX_TRAVERSE_MASK = 3u << 4,
	/* Run the one-ray-per-lane kernel even where the replenishing kernel applies (A/B timing and
	 * tests; results are identical). */
	PTX_NO_PERSISTENT = 1u << 8,
	/* The batch is already ordered so that neighbouring rays travel together (camera rays in
	 * pixel order, ...): skip the library's own grouping pass (results are identical). */
	PTX_RAYS_COHERENT = 1u << 9,
	/* ptx_render_tiles only: walk every ray of the render (extend and shadow rays) in the reference's visit order.
	 * The render's default is the culled order (PTX_TRAVERSE_REFERENCE is 0 and cannot be told from "no flag");
	 * both give the same image bit for bit, this one is the slow cross-check. */
	PTX_RENDER_REFERENCE_ORDER = 1u << 10,
	/* ptx_render_tiles only: also trace the rays whose answer cannot reach the image.  By default two kinds are
	 * left out, with the same accumulator bit for bit: (1) the scattered ray of a path's last vertex -- its ray has
	 * depth 0, and sample_indirect_lighting uses only `.second` of trace(), which is {} at depth 0
	 * (pathtracer.cpp:118,163); (2) in scenes without environment lights and with few, small Emissive instances,
	 * a direct-lighting ray (pathtracer.cpp:64-71: depth 0, only the emission of its first hit is used) that
	 * touches no primitive of an Emissive instance, when the weight it would be multiplied by is finite.
	 * This flag is the cross-check. */
	PTX_RENDER_TRACE_UNUSED_RAYS = 1u << 11
};

/* ------------------------------------------------------------------ rays and hits */
/* lib/ray.h:11-49.  `d` must already be what Ray's constructor would hold (unit length,
 * ray.h:19): hit() takes a constructed Ray and never renormalises. */
typedef struct PtxRay {
	float o[3];
	float d[3];
	float tmin, tmax; /* Ray::dist_bounds */
} PtxRay;             /* 32 bytes */

/* pathtracer/trace.h:10-44 reduced to what identifies a hit.  `distance` is
 * Trace::distance bit for bit, including the reference's miss values: +inf when no
 * primitive was tested, 0 when primitives were tested and all missed (Trace::min of two
 * misses is a default Trace, trace.h:35).  `prim` is the position of the hit primitive in
 * the BVH-ordered primitive array of its mesh (BVH::primitives after build, bvh.h:53), or
 * PTX_NO_HIT. */
#define PTX_NO_HIT 0xFFFFFFFFu
typedef struct PtxHit {
	float distance;
	uint32_t prim;
} PtxHit; /* 8 bytes */

/* ------------------------------------------------------------------ host BVH (PT::BVH<P>::build, bvh.cpp:31-243) */
/* One node of the canonical binary tree, the reference's BVH::Node (bvh.h:15-23) with
 * 32-bit indices.  Leaves have l == r (bvh.h:20). */
typedef struct PtxBvhNode {
	float bmin[3], bmax[3];
	uint32_t start, size, l, r;
} PtxBvhNode; /* 40 bytes */

typedef struct PtxBvh* ptx_bvh_t;

/* Builds the binned-SAH tree over boxes (n x 6 floats: min xyz, max xyz), 8 bins per axis,
 * exactly as BVH::build/buildRecursive do (bvh.cpp:31-243): child sizes and boxes from the bin
 * counts, primitives partitioned by the reference's separately evaluated predicate -- so the
 * node array equals the reference's byte for byte.  One guard: the bin index is clamped to
 * [0,7] where the reference would index bins[] out of range.  ptx_bvh_get_stats reports how
 * often the clamp applied, how often predicate and bin disagreed, and primitives dropped by the
 * reference's "no split separates" branch.  max_leaf_size: 4 for triangles
 * (tri_mesh.cpp:173), 1 for instances (bvh.h:26). */
int ptx_bvh_build(const float* boxes, uint32_t n, uint32_t max_leaf_size, ptx_bvh_t* out);
/* Convenience: boxes are Triangle::bbox (tri_mesh.cpp:10-44) of an indexed mesh whose
 * vertices are 8 floats each (position, normal, uv: Tri_Mesh_Vert, tri_mesh.h:13-17). */
int ptx_bvh_build_triangles(const float* verts, uint32_t n_verts, const uint32_t* indices, uint32_t n_tris,
                            uint32_t max_leaf_size, ptx_bvh_t* out);
/* The same builder on the GPU `device` (BVH::build / buildRecursive, bvh.cpp:31-243, all nodes of a tree level refined
 * together): node array, primitive order and statistics equal ptx_bvh_build's byte for byte.  Inputs that hold a
 * negative zero or a NaN among the box bounds (where the reference's std::min / std::max depend on the order of
 * their arguments) and empty inputs are built on the host; ptx_bvh_built_on_device tells which happened. */
int ptx_bvh_build_gpu(int device, const float* boxes, uint32_t n, uint32_t max_leaf_size, ptx_bvh_t* out);
int ptx_bvh_build_triangles_gpu(int device, const float* verts, uint32_t n_verts, const uint32_t* indices, uint32_t n_tris,
                                uint32_t max_leaf_size, ptx_bvh_t* out);
int ptx_bvh_built_on_device(ptx_bvh_t bvh);                 /* 1 / 0 */
uint32_t ptx_bvh_num_nodes(ptx_bvh_t bvh);
uint32_t ptx_bvh_num_prims(ptx_bvh_t bvh);
uint32_t ptx_bvh_max_depth(ptx_bvh_t bvh);
int ptx_bvh_get_nodes(ptx_bvh_t bvh, PtxBvhNode* out);      /* num_nodes entries */
int ptx_bvh_get_order(ptx_bvh_t bvh, uint32_t* out);        /* num_prims entries: input index of the i-th BVH-ordered primitive */
/* stats[0] primitives whose bin index had to be clamped, [1] primitives the reference's own
 * partition predicate (bvh.cpp:191-197) would have sent to the other side, [2] primitives
 * under nodes no split could separate (the reference drops them: bvh.cpp:183-188, 213-221). */
int ptx_bvh_get_stats(ptx_bvh_t bvh, uint64_t stats[3]);
void ptx_bvh_destroy(ptx_bvh_t bvh);

/* ------------------------------------------------------------------ scene description (Pathtracer::build_scene, pathtracer.cpp:182-387) */
typedef struct PtxMeshDesc {                 /* PT::Tri_Mesh(const Indexed_Mesh&, bool), tri_mesh.cpp:160-177 */
	const float* verts;                      /* n_verts x 8 floats: position, normal, uv */
	uint32_t n_verts;
	const uint32_t* indices;                 /* n_tris x 3 */
	uint32_t n_tris;
	uint32_t use_bvh;                        /* 0: List<Triangle> (--no_bvh), else BVH<Triangle> */
	uint32_t max_leaf_size;                  /* 0 means 4 */
	/* Optional prebuilt tree (BVH::nodes / BVH::primitives are public, bvh.h:53-55): when
	 * `nodes` is non-null the triangles in `indices` are taken to be in BVH order already
	 * and no build happens. */
	const PtxBvhNode* nodes;
	uint32_t n_nodes;
} PtxMeshDesc;

enum { PTX_MAT_LAMBERTIAN = 0, PTX_MAT_MIRROR = 1, PTX_MAT_REFRACT = 2, PTX_MAT_GLASS = 3, PTX_MAT_EMISSIVE = 4 };
enum { PTX_TEX_CONSTANT = 0, PTX_TEX_IMAGE = 1 };
enum { PTX_SAMPLER_NEAREST = 0, PTX_SAMPLER_BILINEAR = 1 };
enum { PTX_LIGHT_POINT = 0, PTX_LIGHT_DIRECTIONAL = 1, PTX_LIGHT_SPOT = 2 };
enum { PTX_ENV_HEMISPHERE = 0, PTX_ENV_SPHERE = 1 };
/* Materials::schlick (scene/material.cpp:58-71) calls unqualified abs(): with g++ it is
 * abs(int) and Glass always reflects off-normal; with the float overload it is Schlick. */
enum { PTX_FRESNEL_GCC_INT_ABS = 0, PTX_FRESNEL_FLOAT_ABS = 1 };

typedef struct PtxTextureDesc {              /* scene/texture.h: Textures::Constant / Textures::Image */
	uint32_t type;
	float color[3];
	float scale;
	uint32_t sampler, w, h;
	const float* pixels;                     /* w*h*3 floats, row 0 at the bottom (hdr_image.h:13) */
} PtxTextureDesc;

typedef struct PtxMaterialDesc {             /* scene/material.h */
	uint32_t type;
	int32_t tex_a;                           /* albedo | reflectance | emissive */
	int32_t tex_b;                           /* transmittance (Refract, Glass) */
	float ior;
} PtxMaterialDesc;

typedef struct PtxInstanceDesc {             /* PT::Instance, pathtracer/instance.h:15-40 */
	int32_t mesh;                            /* index into meshes, or -1 for a Shapes::Sphere of `sphere_radius` */
	float sphere_radius;
	int32_t material;                        /* index into materials; -1 = default Lambertian(0) (pathtracer.cpp:240-252) */
	float T[16];                             /* local_to_world */
} PtxInstanceDesc;

typedef struct PtxDeltaLightDesc {           /* scene/delta_light.h + PT::Light_Instance */
	uint32_t type;
	float color[3];
	float intensity, inner_angle, outer_angle;
	float T[16];
} PtxDeltaLightDesc;

typedef struct PtxEnvLightDesc {             /* scene/env_light.h */
	uint32_t type;
	int32_t tex;
} PtxEnvLightDesc;

typedef struct PtxSceneDesc {
	const PtxMeshDesc* meshes;
	uint32_t n_meshes;
	const PtxInstanceDesc* instances;
	uint32_t n_instances;
	const PtxTextureDesc* textures;
	uint32_t n_textures;
	const PtxMaterialDesc* materials;
	uint32_t n_materials;
	const PtxDeltaLightDesc* delta_lights;
	uint32_t n_delta_lights;
	const PtxEnvLightDesc* env_lights;
	uint32_t n_env_lights;
	uint32_t use_bvh;                        /* top level: BVH<Instance> or List<Instance> (pathtracer.cpp:381-385) */
	uint32_t fresnel_mode;                   /* PTX_FRESNEL_* */
} PtxSceneDesc;

typedef struct PtxScene* ptx_scene_t;

int ptx_device_count(void);
const char* ptx_last_error(void);
const char* ptx_version(void);

/* Pathtracer::build_scene (pathtracer.cpp:182-387): builds every mesh BVH and the
 * instance BVH on the host, flattens them and uploads to `device`. */
int ptx_scene_create(int device, const PtxSceneDesc* desc, ptx_scene_t* out);
/* A scene that is one untransformed Tri_Mesh: hit queries on it are PT::Tri_Mesh::hit
 * (tri_mesh.cpp:193-196) rather than Aggregate::hit. */
int ptx_scene_create_mesh(int device, const PtxMeshDesc* mesh, ptx_scene_t* out);
void ptx_scene_destroy(ptx_scene_t scene);
/* Work is enqueued on `hip_stream` (a hipStream_t; NULL = the null stream). */
int ptx_scene_set_stream(ptx_scene_t scene, void* hip_stream);
/* For a mesh scene: the tree the library built (same layout as ptx_bvh_get_*). */
uint32_t ptx_scene_mesh_num_nodes(ptx_scene_t scene, uint32_t mesh);
int ptx_scene_mesh_get_nodes(ptx_scene_t scene, uint32_t mesh, PtxBvhNode* out);
int ptx_scene_mesh_get_order(ptx_scene_t scene, uint32_t mesh, uint32_t* out);
/* 1 when mesh `mesh`'s tree was built by the GPU builder (meshes of 131,072 triangles and more, unless PTX_GPU_BUILD
 * says otherwise), 0 when by the host builder -- also after the GPU builder failed for want of device memory and the
 * library fell back to the host builder, which gives the same tree byte for byte. */
int ptx_scene_mesh_built_on_device(ptx_scene_t scene, uint32_t mesh);
/* BVH-ordered instance array of the top level: input index of the i-th instance. */
int ptx_scene_get_instance_order(ptx_scene_t scene, uint32_t* out);

/* ------------------------------------------------------------------ ray queries */
/* Closest hit: Aggregate::hit -> BVH<Instance>::hit -> Instance::hit -> Tri_Mesh::hit ->
 * BVH<Triangle>::hit -> Triangle::hit (aggregate.h:41, bvh.cpp:246-375, instance.h:32-40,
 * tri_mesh.cpp:46-111, 193-196).  hits[i] as documented at PtxHit; bary (nullable) receives
 * Triangle::hit's (u, v) per ray; inst (nullable) the position of the hit instance in the
 * BVH-ordered instance array. */
int ptx_trace_closest(ptx_scene_t scene, const PtxRay* rays, uint64_t n, PtxHit* hits, float* bary, uint32_t* inst,
                      uint32_t flags);
/* Occlusion: the reference's only any-hit query is sum_delta_lights' shadow ray,
 * `scene.hit(shadow_ray).hit` (pathtracer.cpp:657-660).  occluded[i] = that boolean,
 * computed with early exit on the first accepted triangle. */
int ptx_trace_any(ptx_scene_t scene, const PtxRay* rays, uint64_t n, uint8_t* occluded, uint32_t flags);

/* ptx_trace_closest for callers that cannot allocate -- code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  Same results as
 * ptx_trace_closest bit for bit; the kernels' working buffers (ray-grouping keys and order, claim
 * counters, the list of rays set aside for the plain-division kernel) come from a fixed device array
 * that the library holds on every GPU it runs on (about 20 MB), so the call performs no hipMalloc /
 * hipFree and creates no events (it is not recorded by ptx_get_kernel_times).
 * Restrictions: rays, hits, bary and inst must be device pointers (PTX_MEM_DEVICE is required);
 * n <= PTX_NOALLOC_MAX_RAYS, larger batches are rejected with PTX_ERR_INVALID.  All calls on one GPU
 * share that GPU's array: two calls that overlap in time on the same GPU (different streams or
 * scenes) race; issue them on one stream, or wait for one before starting the next. */
#define PTX_NOALLOC_MAX_RAYS (1ull << 20)
int ptx_trace_closest_noalloc(ptx_scene_t scene, const PtxRay* rays, uint64_t n, PtxHit* hits, float* bary, uint32_t* inst,
                              uint32_t flags);
uint64_t ptx_noalloc_max_rays(void);

/* Diagnostics (no counterpart in the reference).  ptx_trace_closest_stats: the closest-hit
 * traversal instrumented; stats3[3*i + {0,1,2}] = interior nodes fetched (two box tests each; low
 * 24 bits, the high 8 bits hold the deepest traversal stack), leaves entered, triangles tested
 * for ray i.  ptx_selftest_division: runs `pairs` operand
 * pairs through the traversal's short division form and the IEEE quotient on the device and
 * reports how many differ in any bit. */
int ptx_trace_closest_stats(ptx_scene_t scene, const PtxRay* rays, uint64_t n, uint32_t* stats3, uint32_t flags);
int ptx_selftest_division(int device, uint64_t pairs, uint64_t seed, uint64_t* mismatches);
/* The render's random stream on the device (it replaces the reference's per-tile mt19937, util/rand.cpp:17-40).
 * ptx_selftest_philox: n Philox-4x32-10 blocks, counters4[4*i..] and keys2[2*i..] -> out4[4*i..], for checking the
 * generator against the published known-answer vectors.  ptx_selftest_path_units: the first n RNG::unit() values of
 * the stream of (seed, pixel, sample) exactly as ptx_render_tiles consumes them. */
int ptx_selftest_philox(int device, const uint32_t* counters4, const uint32_t* keys2, uint32_t n, uint32_t* out4);
int ptx_selftest_path_units(int device, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float* out);

/* ------------------------------------------------------------------ rendering */
typedef struct PtxCamera {                   /* scene/camera.h:12-30 + Pathtracer::set_camera (pathtracer.cpp:389-396) */
	float vertical_fov, aspect_ratio;
	uint32_t film_width, film_height, film_samples, film_max_ray_depth;
	float camera_to_world[16];
} PtxCamera;

typedef struct PtxTile {                     /* Pathtracer::Tile, pathtracer.cpp:536-546 */
	uint32_t seed, x_begin, x_end, y_begin, y_end, s_begin, s_end;
} PtxTile;

/* Pathtracer::do_trace + accumulate (pathtracer.cpp:444-477, 407-427) for a batch of tiles:
 * for every pixel of every tile, the tile's samples are traced (Pathtracer::trace,
 * pathtracer.cpp:126-173) and summed in float in sample order, non-finite samples skipped;
 * then accum_rgb[3*pixel + c] += int64(sum_c * 2^24) and samples[pixel] += s_end - s_begin.
 * accum_rgb (3 x int64 per pixel, 40.24 fixed point) and samples (uint32 per pixel) are
 * film_width*film_height entries and are accumulated into, not cleared
 * (render(..., add_samples = true), pathtracer.cpp:504-519).
 * Random numbers: Philox-4x32-10 with key = seed and one stream per (pixel, sample); draw i of
 * a path is the i-th number the reference's RNG::unit() would have produced on that path.
 * PtxTile::seed is ignored (it seeds the reference's per-tile mt19937).
 * Limits: film_max_ray_depth <= 8191 (PTX_ERR_UNSUPPORTED above); device memory 172 B per path in flight (at most
 * 128 M paths per pass) plus 49 B per path and point / spot / directional light.
 * Flags: PTX_MEM_DEVICE, PTX_ASYNC, PTX_NO_PERSISTENT, PTX_RENDER_REFERENCE_ORDER, PTX_RENDER_TRACE_UNUSED_RAYS. */
int ptx_render_tiles(ptx_scene_t scene, const PtxCamera* camera, const PtxTile* tiles, uint32_t n_tiles, uint64_t seed,
                     int64_t* accum_rgb, uint32_t* samples, uint32_t flags);

/* Pathtracer::accumulator / accumulator_samples (pathtracer.h:80-83) kept on the scene's GPU between
 * calls, so that a render made of many ptx_render_tiles calls (the reference reports progress tile by
 * tile, pathtracer.cpp:568-584) moves the film across PCIe only when somebody looks at it.
 * ptx_frame_create gives a zeroed accumulator of width x height pixels; its device arrays go to
 * ptx_render_tiles with PTX_MEM_DEVICE.  upload/download copy the whole film from/to host arrays of
 * 3*width*height int64 and width*height uint32 (upload: render(..., add_samples = true) continuing
 * a host-side accumulator, pathtracer.cpp:504-519) and return after the copy has completed. */
typedef struct PtxFrame* ptx_frame_t;
int ptx_frame_create(ptx_scene_t scene, uint32_t width, uint32_t height, ptx_frame_t* out);
int64_t* ptx_frame_accum(ptx_frame_t frame);      /* device pointer, int64[3*width*height] */
uint32_t* ptx_frame_samples(ptx_frame_t frame);   /* device pointer, uint32[width*height] */
int ptx_frame_upload(ptx_frame_t frame, const int64_t* accum_rgb, const uint32_t* samples);
int ptx_frame_download(ptx_frame_t frame, int64_t* accum_rgb, uint32_t* samples);
int ptx_frame_clear(ptx_frame_t frame);
/* The reference's tiles run on a thread pool and every finished tile is added into ONE accumulator under a mutex
 * (pathtracer.cpp:566-584, 407-427).  With several GPUs every GPU accumulates the tiles it was dealt in a film of its
 * own (ptx_frame_create on that GPU's scene); ptx_frame_reduce then adds every part into `dst` (dst += sum of parts,
 * parts unchanged): integer sums of 40.24 fixed-point values and of the sample counts, so the frame is the one a single
 * accumulator would hold, whatever the order (pathtracer.h:80-83).  Parts on dst's own GPU and on GPUs it can address
 * as peers (xGMI; peer access is enabled on first use) are read where they lie by one kernel per eight parts; parts it
 * cannot address are copied into staging films on dst's GPU first.  Staging memory, copy streams and events belong to
 * `dst` and are kept until it is destroyed: calls after the first allocate nothing.  The call first waits for what the
 * parts' own scenes have queued on their streams (a render with PTX_ASYNC still in flight); nothing new may be queued
 * on the parts while it runs. */
int ptx_frame_reduce(ptx_frame_t dst, const ptx_frame_t* parts, uint32_t n_parts);
/* What the last ptx_frame_reduce into `dst` did (no counterpart in the reference: its accumulator never leaves one
 * address space). */
typedef struct PtxFrameReduceInfo {
	uint32_t parts_local;        /* parts on dst's GPU, added where they lie */
	uint32_t parts_peer_direct;  /* parts on another GPU, read over the link by the adding kernel (peer access) */
	uint32_t parts_staged;       /* parts copied into a staging film first (no peer access, or PTX_FRAME_REDUCE_STAGE set) */
	uint32_t add_launches;       /* launches of the adding kernel */
	uint32_t allocations;        /* device allocations, streams and events this call created (0 once dst has served a call of this shape) */
	uint32_t pad;
	uint64_t bytes_over_links;   /* film bytes read or copied from other GPUs */
} PtxFrameReduceInfo;
int ptx_frame_reduce_info(ptx_frame_t dst, PtxFrameReduceInfo* out);
/* The same sums as RCCL collectives over xGMI, for a process that drives several GPUs (one host thread and scene per
 * GPU, as PT::Pathtracer::set_devices does).  ptx_comm_create makes one RCCL rank per listed GPU (ncclCommInitAll;
 * every GPU at most once); RCCL itself is loaded at the first call, from the copy the process already holds (PyTorch
 * ships one) or else from the ROCm installation, PTX_RCCL_LIB overriding both.
 * ptx_frame_reduce_rccl: dst = sum of parts (NOT added to what dst held), parts[i] on the communicator's i-th GPU, dst
 * on one of those GPUs; dst may be one of the parts.  One ncclReduce of int64[3*W*H] and one of uint32[W*H] per rank
 * in one group.  ptx_frame_allreduce: every film ends holding the sum of all (ncclAllReduce in place) -- the exchange
 * ranks in separate processes perform through torch.distributed.  Both first wait for the films' own streams and
 * return when the result is complete.
 * They stand where the reference adds a tile under its mutex (pathtracer.cpp:407-427, 566-584). */
typedef struct PtxComm* ptx_comm_t;
int ptx_comm_create(const int* devices, uint32_t n_devices, ptx_comm_t* out);
uint32_t ptx_comm_size(ptx_comm_t comm);
const char* ptx_comm_library(ptx_comm_t comm);       /* the RCCL library file in use */
void ptx_comm_destroy(ptx_comm_t comm);
int ptx_frame_reduce_rccl(ptx_comm_t comm, ptx_frame_t dst, const ptx_frame_t* parts, uint32_t n_parts);
int ptx_frame_allreduce(ptx_comm_t comm, const ptx_frame_t* frames, uint32_t n_frames);
void ptx_frame_destroy(ptx_frame_t frame);

/* Pathtracer::render's tile list (pathtracer.cpp:522-563): 100x100-pixel x 50-sample tiles,
 * seeded from one mt19937(fixed_seed) in y, x, s order, stably sorted centre-out then by
 * s_begin.  Returns the tile count; writes at most `cap` tiles when `out` is non-null. */
uint32_t ptx_make_tiles(uint32_t film_width, uint32_t film_height, uint32_t film_samples, uint32_t fixed_seed, PtxTile* out,
                        uint32_t cap);
/* Pathtracer::accumulator_to_image (pathtracer.cpp:429-442): rgb[3*i+c] = acc / 2^24 / samples in double. */
int ptx_accumulator_to_image(const int64_t* accum_rgb, const uint32_t* samples, uint64_t n_pixels, float* rgb);
/* HDR_Image::tonemap_to (util/hdr_image.cpp:201-225): 1-exp(-c*exposure), sRGB, round(*255); RGBA8. */
int ptx_tonemap(const float* rgb, uint64_t n_pixels, float exposure, uint8_t* rgba);

typedef struct PtxCounters {
	uint64_t rays_closest;   /* closest-hit queries since the scene was created */
	uint64_t rays_any;       /* occlusion queries */
	uint64_t paths;          /* camera samples traced by ptx_render_tiles */
	/* Rays ptx_render_tiles traced for those paths, counted on the device: camera and scattered ("indirect") rays, and
	 * direct-lighting rays (pathtracer.cpp:64-76).  The reference traces 1 + 2 per scattering vertex; rays whose answer
	 * cannot reach the image are left out unless PTX_RENDER_TRACE_UNUSED_RAYS is set.  Shadow rays of delta lights
	 * (pathtracer.cpp:657) are not included. */
	uint64_t render_main_rays;
	uint64_t render_direct_rays;
} PtxCounters;
/* Waits for the scene's stream when a render has run (the ray totals are read back from the device). */
int ptx_get_counters(ptx_scene_t scene, PtxCounters* out);
/* Device time of the single-mesh traversal kernel (the dominant kernel of ptx_trace_closest /
 * ptx_trace_any on a ptx_scene_create_mesh scene), from HIP events recorded on the scene's stream
 * around each launch: the last min(launches, 64, cap) durations in milliseconds, oldest first.
 * Waits for the stream.  Stands where the reference's build_timer / render_timer stand
 * (pathtracer.h:76, completion_time(), pathtracer.cpp:483-485). */
int ptx_get_kernel_times(ptx_scene_t scene, float* ms, uint32_t cap, uint32_t* n_out);

/* ------------------------------------------------------------------ scene files and image output */
/* The reference's JSON scene format ("js3d-v1"): load(filepath, &scene, &animator)
 * (scene/io.cpp:15-70, scene/load-save-json.cpp:108-333) followed by what Pathtracer::build_scene
 * (pathtracer.cpp:182-387) derives from a Scene: half-edge meshes flattened with
 * Indexed_Mesh::from_halfedge_mesh(..., SplitEdges) (geometry/indexed.cpp:10-34), one instance per
 * visible mesh / shape instance with its Transform::local_to_world() (scene/transform.cpp:12-24),
 * materials, textures, delta-light instances and every environment light.  Names are visited in
 * sorted order.  Skinned meshes, particles and the binary .s3d format are not read (a warning is
 * recorded).  The handle owns every array the returned PtxSceneDesc points to. */
typedef struct PtxJs3d* ptx_js3d_t;
int ptx_js3d_load(const char* path, ptx_js3d_t* out);
void ptx_js3d_free(ptx_js3d_t s);
const PtxSceneDesc* ptx_js3d_scene_desc(ptx_js3d_t s);
/* Pathtracer::use_bvh (pathtracer.h:37: meshes and the instance level alike) and the Fresnel mode. */
void ptx_js3d_set_modes(ptx_js3d_t s, uint32_t use_bvh, uint32_t fresnel_mode);
/* Camera instances (scene.instances.cameras, main.cpp:125-136): film and lens values of the
 * camera resource, camera_to_world = the instance transform's local_to_world(). */
uint32_t ptx_js3d_num_cameras(ptx_js3d_t s);
const char* ptx_js3d_camera_name(ptx_js3d_t s, uint32_t i);
int ptx_js3d_get_camera(ptx_js3d_t s, const char* name, PtxCamera* out);
uint32_t ptx_js3d_num_meshes(ptx_js3d_t s);
const char* ptx_js3d_mesh_name(ptx_js3d_t s, uint32_t i);
const char* ptx_js3d_instance_name(ptx_js3d_t s, uint32_t i);
const char* ptx_js3d_warnings(ptx_js3d_t s);
/* stbi_write_png with stbi_flip_vertically_on_write(true) (main.cpp:339-343): 8-bit RGBA, image
 * row 0 is the bottom row (util/hdr_image.h:13) and is written last when flip_rows is non-zero. */
int ptx_write_png(const char* path, const uint8_t* rgba, uint32_t width, uint32_t height, int flip_rows);

#ifdef __cplusplus
}
#endif
#endif /* PTX_AMD_H */

// render.hip -- Pathtracer::do_trace / trace / accumulate as a wavefront pipeline.
//
// The reference traces one camera sample at a time, recursively (pathtracer.cpp:126-173,
// 444-477).  Its recursion is linear: at every vertex one "direct" ray of depth 0, whose
// `.first` (the emission it meets) is used, and one "indirect" ray that continues the path.
// Here every pixel of every tile in the batch is a SLOT.  For sample k of the tiles, all
// slots run in lock step through
//
//   raygen  -> [ extend -> shade/scatter/compact [-> shadow pass -> apply] ] x (max_ray_depth + 1) -> add to slot sum
//
//   raygen   camera ray per slot (Camera::sample_ray, 2 draws), every slot into the active list
//   extend   closest hit of each active path's main ray and of the direct rays the shade stage asked for
//   shade    uses the two hits: adds the direct ray's emission, shades the main ray's vertex (emission, BSDF scatter
//            twice, the delta lights' shadow rays and what each light would add), writes the next main + direct
//            rays; surviving paths and wanted direct rays are appended to their lists with one atomic per block
//            (ballot + prefix popcount + one LDS exchange)
//   shadow   (scenes with point / spot / directional lights) any-hit queries for the shadow rays, then the unoccluded
//            lights' contributions are added in light order
//   Rays whose answer cannot reach the image are not traced (see PTX_RENDER_TRACE_UNUSED_RAYS in ptx.h).
//
// Samples of a slot are added in sample order (k_sum_pass folds a pass's samples into the slot's float sum in
// sample order), so a slot's float sum is the same sequence of additions as the reference's tile-private
// HDR_Image pixel (pathtracer.cpp:447-470).  When the tile's samples are done the sum goes to
// the 40.24 fixed-point accumulator exactly as Pathtracer::accumulate does
// (pathtracer.cpp:407-427); integer atomics make tiles that share pixels commute.
//
// Randomness: one Philox-4x32-10 stream per (pixel, sample); the draws are consumed in the
// order the reference's recursion consumes RNG::unit().
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "scene.h"
#include "scene_persistent.hip.h"
#include "shade.hip.h"
#include "traverse.hip.h"

namespace ptx {

constexpr float RAY_MINIMUM_BOUND = 0.001f;   // pathtracer.cpp:17
// Paths in flight per pass (172 B of state each: 23 GB at the limit).  Measured on a 268 M-path render (config 2):
// 32 M paths per pass 0.348 s, 64 M 0.335 s, 128 M 0.331 s, 256 M 0.331 s.
#ifndef PTX_MAX_PATHS_LOG2
#define PTX_MAX_PATHS_LOG2 27
#endif
constexpr uint64_t MAX_SLOTS_PER_CHUNK = uint64_t(1) << PTX_MAX_PATHS_LOG2;

struct alignas(16) SlotInfo {
	uint32_t pixel, s_begin, s_count, pad;
};

// What a path carries from one round to the next is one 16-byte word: the throughput of its main ray's path so
// far and, packed into the fourth component, Ray::depth of the main ray, the number of random draws consumed, and
// two flags.  Which sample of which film pixel the path is (the random stream's name) follows from its index.
constexpr uint32_t PATH_DEPTH_BITS = 13, PATH_DRAWS_BITS = 17;
constexpr uint32_t PATH_MAX_DEPTH = (1u << PATH_DEPTH_BITS) - 1;   // at most 2 + 6 draws per vertex: 49 148 < 2^17
constexpr uint32_t PATH_CAMERA = 1u << 30, PATH_HAS_DIRECT = 1u << 31;
__device__ __forceinline__ uint32_t path_pack(uint32_t depth, uint32_t draws, uint32_t flags) { return depth | (draws << PATH_DEPTH_BITS) | flags; }
__device__ __forceinline__ uint32_t path_depth(uint32_t w) { return w & PATH_MAX_DEPTH; }
__device__ __forceinline__ uint32_t path_draws(uint32_t w) { return (w >> PATH_DEPTH_BITS) & ((1u << PATH_DRAWS_BITS) - 1); }

struct TileRec {
	uint32_t x_begin, y_begin, w, h, s_begin, s_count, slot_base, pad;
};

// A slot is one pixel of one tile; a path is one sample of a slot.  `group` samples of every slot are in
// flight together: path g * n_slots + slot carries the g-th sample of the current pass of `slot`
// (neighbouring paths are neighbouring pixels).  Small films thus still fill the chip.
// Per-path records are kept as separate arrays of 16-byte (8-byte) pieces: the lanes of a wave work on
// neighbouring paths, so each of their loads and stores covers one contiguous kilobyte.
struct RenderBuffers {
	SlotInfo* slots;
	float4* sums;         // per slot: float sum of its samples so far (rgb)
	float4* rad;          // per path: the radiance gathered so far; what its sample ends with
	float4* state;        // per path: throughput (xyz), packed depth / draws / flags (w)
	float4* wd;           // per path: weight of the pending direct ray's emission (T * attenuation [/ pdf])
	float4* ray_a[2];     // per path, [0] main ray, [1] direct ray: origin, direction.x
	float4* ray_b[2];     //                                         direction.yz, distance bounds
	float4* hit_a[2];     // per path and ray: distance, t, u, v
	uint2* hit_b[2];      //                   primitive (0xFFFFFFFF: none), instance
	uint32_t* active[2];
	uint32_t* direct;     // the paths of the next active list whose direct ray is to be traced
	uint32_t* counts;     // [0], [1]: sizes of active[0], active[1]; [2]: size of direct
	uint32_t n_slots, group;
	// Scenes with point / spot / directional lights: per path and light, the shadow ray of sum_delta_lights
	// (pathtracer.cpp:645-666), what it adds to the path's radiance when nothing blocks it (w: 1 when the light
	// takes part at this vertex, else 0), and the occlusion query's answer.
	PtxRay* shadow_rays;
	float4* shadow_add;
	uint8_t* shadow_occ;
	uint32_t n_lights;
};
// Which sample of which slot a path is.  PATH_BLOCK consecutive slots (neighbouring pixels of a tile) carry all the
// samples of the pass before the next PATH_BLOCK slots begin: path = (block * group + g) * PATH_BLOCK + slot % PATH_BLOCK.
// The lanes of a wave are still neighbouring pixels of one sample (coalesced slot records, near-identical camera
// rays), while the paths in flight on the chip at any time -- a window of a few hundred thousand consecutive paths --
// cover a small patch of the film with all its samples instead of a third of the film with one: their rays start in
// the same part of the scene at every bounce, which is what the caches see.  The last, partial block of slots is laid
// out the same way with its own width.  PATH_BLOCK 0: one block as wide as the film's slots (sample-major).
#ifndef PTX_PATH_BLOCK
#define PTX_PATH_BLOCK 1024   // measured: 64, 1024 and 16384 within 0.3 % of each other, all 1-3.5 % faster than sample-major
#endif
constexpr uint32_t PATH_BLOCK = PTX_PATH_BLOCK;
__device__ __forceinline__ void path_to_slot(const RenderBuffers& rb, uint32_t path, uint32_t& slot, uint32_t& g) {
	if (PATH_BLOCK == 0) {
		slot = path % rb.n_slots, g = path / rb.n_slots;
		return;
	}
	const uint32_t full = rb.n_slots / PATH_BLOCK, span = PATH_BLOCK * rb.group;
	const uint32_t block = path / span;
	if (block < full) {
		slot = block * PATH_BLOCK + path % PATH_BLOCK, g = (path / PATH_BLOCK) % rb.group;
	} else {
		const uint32_t r = path - full * span, rem = rb.n_slots - full * PATH_BLOCK;
		slot = full * PATH_BLOCK + r % rem, g = r / rem;
	}
}
__device__ __forceinline__ size_t slot_to_path(const RenderBuffers& rb, uint32_t slot, uint32_t g) {
	if (PATH_BLOCK == 0) return size_t(g) * rb.n_slots + slot;
	const uint32_t full = rb.n_slots / PATH_BLOCK;
	if (slot < full * PATH_BLOCK) return (size_t(slot / PATH_BLOCK) * rb.group + g) * PATH_BLOCK + slot % PATH_BLOCK;
	const uint32_t rem = rb.n_slots - full * PATH_BLOCK;
	return size_t(full) * PATH_BLOCK * rb.group + size_t(g) * rem + (slot - full * PATH_BLOCK);
}

constexpr size_t PATH_STATE_BYTES = 3 * sizeof(float4) + 2 * (2 * sizeof(float4) + sizeof(float4) + sizeof(uint2)) + 3 * sizeof(uint32_t);

// Records move between registers and memory as whole 16-byte pieces (one vector load or store per
// piece, adjacent lanes adjacent in memory) instead of one dword at a time.
template<typename T>
__device__ __forceinline__ T load_rec(const T* p) {
	static_assert(sizeof(T) % 16 == 0, "whole 16-byte pieces");
	union {
		T v;
		uint4 q[sizeof(T) / 16];
	} u;
	const uint4* src = reinterpret_cast<const uint4*>(p);
#pragma unroll
	for (size_t i = 0; i < sizeof(T) / 16; i++) u.q[i] = src[i];
	return u.v;
}
template<typename T>
__device__ __forceinline__ void store_rec(T* p, const T& v) {
	static_assert(sizeof(T) % 16 == 0, "whole 16-byte pieces");
	union {
		T v;
		uint4 q[sizeof(T) / 16];
	} u;
	u.v = v;
	uint4* dst = reinterpret_cast<uint4*>(p);
#pragma unroll
	for (size_t i = 0; i < sizeof(T) / 16; i++) dst[i] = u.q[i];
}

__global__ void __launch_bounds__(256) k_init_slots(const TileRec* __restrict__ tiles, uint32_t film_w, RenderBuffers rb) {
	const TileRec t = tiles[blockIdx.x];
	for (uint32_t p = threadIdx.x; p < t.w * t.h; p += blockDim.x) {
		const uint32_t x = t.x_begin + p % t.w, y = t.y_begin + p / t.w;
		store_rec(rb.slots + t.slot_base + p, SlotInfo{y * film_w + x, t.s_begin, t.s_count, 0});
		rb.sums[t.slot_base + p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	}
}

// Appends `value` for every thread with `keep` set, in thread order: one atomic per block.  Every
// thread of the block calls it.  (One atomic per wave, all on the one counter, was what the ray
// generation and shading kernels spent most of their time on: same-address atomics retire one at a
// time, some 13 ns each.)
template<int WAVES>
__device__ __forceinline__ void block_append(bool keep, uint32_t value, uint32_t* list, uint32_t* count) {
	__shared__ uint32_t wave_base[WAVES];
	const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const unsigned long long mask = __ballot(keep);
	if (lane == 0) wave_base[wave] = uint32_t(__popcll(mask));
	__syncthreads();
	if (threadIdx.x < 64) {   // first wave: exclusive prefix over the waves' counts, one atomic for the block
		const uint32_t c = lane < WAVES ? wave_base[lane] : 0u;
		uint32_t incl = c;
		for (int d = 1; d < WAVES; d <<= 1) {
			const uint32_t o = __shfl_up(incl, d);
			if (int(lane) >= d) incl += o;
		}
		const uint32_t total = __shfl(incl, WAVES - 1);
		uint32_t base = 0;
		if (lane == 0 && total) base = atomicAdd(count, total);
		base = __shfl(base, 0);
		if (lane < WAVES) wave_base[lane] = base + incl - c;
	}
	__syncthreads();
	if (keep) list[wave_base[wave] + __popcll(mask & ((1ull << lane) - 1ull))] = value;
}

// Two appends at once (the shade stage's survivors and its direct rays): one pair of barriers for both, and the two
// atomics are in flight together.
template<int WAVES>
__device__ __forceinline__ void block_append2(bool keep_a, uint32_t* list_a, uint32_t* count_a, bool keep_b, uint32_t* list_b, uint32_t* count_b, uint32_t value) {
	__shared__ uint32_t wave_base[2][WAVES];
	const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const unsigned long long mask_a = __ballot(keep_a), mask_b = __ballot(keep_b);
	if (lane == 0) wave_base[0][wave] = uint32_t(__popcll(mask_a)), wave_base[1][wave] = uint32_t(__popcll(mask_b));
	__syncthreads();
	if (threadIdx.x < 64) {
		const uint32_t ca = lane < WAVES ? wave_base[0][lane] : 0u, cb = lane < WAVES ? wave_base[1][lane] : 0u;
		uint32_t ia = ca, ib = cb;
		for (int d = 1; d < WAVES; d <<= 1) {
			const uint32_t oa = __shfl_up(ia, d), ob = __shfl_up(ib, d);
			if (int(lane) >= d) ia += oa, ib += ob;
		}
		const uint32_t total_a = __shfl(ia, WAVES - 1), total_b = __shfl(ib, WAVES - 1);
		uint32_t base = 0;
		if (lane == 0 && total_a) base = atomicAdd(count_a, total_a);
		if (lane == 1 && total_b) base = atomicAdd(count_b, total_b);
		const uint32_t base_a = __shfl(base, 0), base_b = __shfl(base, 1);
		if (lane < WAVES) wave_base[0][lane] = base_a + ia - ca, wave_base[1][lane] = base_b + ib - cb;
	}
	__syncthreads();
	const unsigned long long below = (1ull << lane) - 1ull;
	if (keep_a) list_a[wave_base[0][wave] + __popcll(mask_a & below)] = value;
	if (keep_b) list_b[wave_base[1][wave] + __popcll(mask_b & below)] = value;
}

constexpr int RAYGEN_BLOCK = 1024;
// The shade stage at 92-102 VGPRs fits one 1024-thread block per CU, and the CU would idle while that block waits at
// its barriers and for its atomics; two 512-thread blocks overlap (measured: 6 % of a whole render; 256-thread blocks
// lose part of it again to the queue at the list counter).
#ifndef PTX_SHADE_BLOCK
#define PTX_SHADE_BLOCK 512
#endif
constexpr int SHADE_BLOCK = PTX_SHADE_BLOCK;
#ifndef PTX_SHADE_EARLY_LOADS
#define PTX_SHADE_EARLY_LOADS 2   // 0: every load where it is used, 1: main ray and hit early, 2: the slot record too (measured: -0.7 %, -1.1 % on config 2)
#endif
#ifndef PTX_SHADE_LDS_INSTS
#define PTX_SHADE_LDS_INSTS 16
#endif
constexpr int SHADE_LDS_INSTS = PTX_SHADE_LDS_INSTS;   // instance records a shade block copies into LDS (scenes with at most this many; 0: never)
// Experiment switch: waves per SIMD the compiler is asked to fit the shade kernel into (0: its own choice).
#ifndef PTX_SHADE_WAVES
#define PTX_SHADE_WAVES 0
#endif
#if PTX_SHADE_WAVES
#define PTX_SHADE_OCCUPANCY __attribute__((amdgpu_waves_per_eu(PTX_SHADE_WAVES, PTX_SHADE_WAVES)))
#else
#define PTX_SHADE_OCCUPANCY
#endif
#ifndef PTX_SHADE_SORT
#define PTX_SHADE_SORT 0
#endif
static_assert(!PTX_SHADE_SORT || (SHADE_BLOCK / 64) * 8 <= 64, "the slot table of the in-block regrouping is scanned by one wave");

// Camera::sample_ray for samples k0 .. k0 + group - 1 of every slot that has them (pathtracer.cpp:448-454).
__global__ void __launch_bounds__(RAYGEN_BLOCK) k_raygen(RenderBuffers rb, uint32_t k0, CameraRec cam, uint32_t key0, uint32_t key1,
                                                uint32_t* __restrict__ out_list, uint32_t* __restrict__ out_count) {
	const uint32_t path = blockIdx.x * blockDim.x + threadIdx.x;
	bool live = false;
	if (path < rb.n_slots * rb.group) {
		uint32_t slot, g;
		path_to_slot(rb, path, slot, g);
		const uint32_t k = k0 + g;
		const SlotInfo si = load_rec(rb.slots + slot);
		if (k < si.s_count) {
			live = true;
			PathRng rng;
			rng.start(key0, key1, si.pixel, si.s_begin + k, 0);
			const float ox = rng.next_unit(), oy = rng.next_unit();   // Samplers::Rect::sample, samplers.cpp:13 (x first)
			const Ray3 r = camera_ray(cam, si.pixel % cam.film_w, si.pixel / cam.film_w, ox, oy);
			rb.state[path] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(path_pack(cam.max_depth, rng.draws, PATH_CAMERA)));
			rb.rad[path] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			rb.ray_a[0][path] = make_float4(r.ox, r.oy, r.oz, r.dx);
			rb.ray_b[0][path] = make_float4(r.dy, r.dz, r.tmin, r.tmax);
		}
	}
	block_append<RAYGEN_BLOCK / 64>(live, path, out_list, out_count);
}

// The rays of one extend round as a ray source / hit sink of the traversal kernels: the main rays of the active
// paths first (neighbouring pixels, the more coherent half), then the direct rays k_shade asked for.
struct RenderIO {
	RenderBuffers rb;
	const uint32_t* list;
	const uint32_t* n_paths;
	uint32_t main_rays;       // 0: the main rays are not traced this round (nothing would use their hits)
	__device__ __forceinline__ uint32_t n_main() const { return main_rays ? *n_paths : 0u; }
	__device__ __forceinline__ uint64_t count() const { return uint64_t(n_main()) + rb.counts[2]; }
	// a ray's name in the traversal kernels: 2 * path + (0: main ray, 1: direct ray)
	__device__ __forceinline__ void load(uint32_t out, Ray3& r) const {
		const uint32_t path = out >> 1;
		const float4 a = (out & 1u ? rb.ray_a[1] : rb.ray_a[0])[path], b = (out & 1u ? rb.ray_b[1] : rb.ray_b[0])[path];
		r = Ray3{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
	}
	__device__ __forceinline__ bool fetch(uint64_t item, Ray3& r, uint32_t& out) const {
		const uint32_t n = n_main();
		out = item < n ? 2u * list[item] : 2u * rb.direct[item - n] + 1u;
		load(out, r);
		return true;
	}
	__device__ __forceinline__ void store(uint32_t out, const SceneHit& h) const {
		const uint32_t path = out >> 1;
		(out & 1u ? rb.hit_a[1] : rb.hit_a[0])[path] = make_float4(h.distance, h.t, h.u, h.v);
		(out & 1u ? rb.hit_b[1] : rb.hit_b[0])[path] = make_uint2(h.hit ? h.prim : 0xFFFFFFFFu, h.inst);
	}
};

// Closest hits one ray per lane: small batches, scenes the replenishing kernel does not take, reference order.
template<int MODE>
__global__ void __launch_bounds__(256) k_extend(SceneRec sc, RenderIO io) {
	const uint64_t item = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (item >= io.count()) return;
	Ray3 r;
	uint32_t out;
	io.fetch(item, r, out);
	Stack st;
	SceneHit h;
	traverse_scene<MODE, false>(sc, r, st, h);
	if (MODE == MODE_CULLED && needs_reference_order(r, h.hit, h.distance)) traverse_scene<MODE_REFERENCE, false>(sc, r, st, h);
	io.store(out, h);
}

// Whether a direct-lighting ray can bring back light at all: it touches a primitive of an Emissive instance.
// The traversal proper would run these very tests (same object-space ray, same functions) on the primitives whose
// boxes the ray passes, so a hit it finds on an Emissive instance is found here too; a ray for which this returns
// false ends, in the reference, on a surface whose emission is zero or on nothing (pathtracer.cpp:129-163).
__device__ __forceinline__ bool direct_ray_may_see_light(const SceneRec& sc, const ShadeTables& sh, const Ray3& world) {
	for (uint32_t e = 0; e < sh.n_emissive; e++) {
		const InstRec& in = sc.insts[sh.emissive_insts[e]];
		Ray3 r = world;
		if (in.has_transform) r = ray_transform(in.iT, world);
		if (in.mesh < 0) {
			float t;
			if (sphere_test(in.radius, r, t)) return true;
		} else {
			const MeshRec& mesh = sc.meshes[in.mesh];
			const float4* p = reinterpret_cast<const float4*>(sc.tris + mesh.tri_base);
			for (uint32_t k = 0; k < mesh.n_tris; k++) {
				float d, t, u, v;
				if (triangle_test(p[3 * k], p[3 * k + 1], p[3 * k + 2], r, d, t, u, v)) return true;
			}
		}
	}
	return false;
}

// What Pathtracer::trace needs of a Trace (trace.h:20-26) rebuilt from a HitRec.
struct Surface {
	V3f position, normal;
	float u, v;
	int32_t material;
};

// `insts`: the scene's instance table, or the block's copy of it in LDS.
__device__ __forceinline__ Surface rebuild_surface(const InstRec* insts, const SceneRec& sc, const ShadeTables& sh, const Ray3& world, const HitRec& h) {
	Surface s;
	const InstRec& in = insts[h.inst];
	Ray3 r = world;
	if (in.has_transform) r = ray_transform(in.iT, world);   // Instance::hit, instance.h:33
	V3f pos = v3(r.ox, r.oy, r.oz) + v3(r.dx, r.dy, r.dz) * h.t;   // ray.point + dir * t
	V3f nrm;
	if (in.mesh >= 0) {
		const TriAttr& at = sh.attrs[sc.meshes[in.mesh].tri_base + h.prim];
		const float w = 1 - h.u - h.v;   // tri_mesh.cpp:100-101
		nrm = w * v3(at.n0[0], at.n0[1], at.n0[2]) + h.u * v3(at.n1[0], at.n1[1], at.n1[2]) + h.v * v3(at.n2[0], at.n2[1], at.n2[2]);
		s.u = (w * at.uv0[0] + h.u * at.uv1[0]) + h.v * at.uv2[0];
		s.v = (w * at.uv0[1] + h.u * at.uv1[1]) + h.v * at.uv2[1];
	} else {
		nrm = unit(pos);   // shape.cpp:80-81
		sphere_uv(nrm, s.u, s.v);
	}
	if (in.has_transform) {   // Trace::transform(T, iT.T()), trace.h:38-43
		mat_point(in.T, pos.x, pos.y, pos.z, pos.x, pos.y, pos.z);
		// N = iT.T(): its columns are the rows of iT
		const float* m = in.iT;
		const V3f n = nrm;
		nrm.x = ((n.x * m[0] + n.y * m[1]) + n.z * m[2]) + 0.0f * m[3];
		nrm.y = ((n.x * m[4] + n.y * m[5]) + n.z * m[6]) + 0.0f * m[7];
		nrm.z = ((n.x * m[8] + n.y * m[9]) + n.z * m[10]) + 0.0f * m[11];
		nrm = unit(nrm);
	}
	s.position = pos, s.normal = nrm, s.material = in.material;
	return s;
}

#ifdef PTX_PROFILE_SHADE
// Lane statistics of k_shade (a -DPTX_PROFILE_SHADE build): per site, the lanes that were active and the waves that came by.
static __device__ unsigned long long g_shade_prof[32];
#define SHADE_SITE(k)                                                                                  \
	do {                                                                                               \
		const unsigned long long m_ = __ballot(1);                                                     \
		if ((threadIdx.x & 63u) == unsigned(__ffsll((long long)m_) - 1))                               \
			atomicAdd(&g_shade_prof[2 * (k)], (unsigned long long)__popcll(m_)), atomicAdd(&g_shade_prof[2 * (k) + 1], 1ull); \
	} while (0)
#else
#define SHADE_SITE(k) do { } while (0)
#endif

// The body of Pathtracer::trace for one active path, one recursion level per launch.
// DELTA: the scene has point/spot/directional lights.  Their shadow rays are not walked here: the kernel leaves,
// per light, the shadow ray and what the light adds when unoccluded; an occlusion pass over all of them follows
// (any-hit queries, the replenishing two-level kernel on large batches) and k_apply_shadows adds the unoccluded ones
// to the path's radiance in light order -- the same additions in the same order as the reference's loop, since
// nothing else touches the radiance between them and the next level's first addition.
// k0: the first sample (per slot) of the pass in flight.
template<bool DELTA>
__global__ void __launch_bounds__(SHADE_BLOCK) PTX_SHADE_OCCUPANCY k_shade(SceneRec sc, ShadeTables sh, RenderBuffers rb, const uint32_t* __restrict__ list,
                                               const uint32_t* __restrict__ count, uint32_t k0, uint32_t key0, uint32_t key1,
                                               uint32_t* __restrict__ next_list, uint32_t* __restrict__ next_count) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	// A scene of few instances (the Cornell boxes: nine) has its instance table copied into LDS by every block: each
	// scattering vertex reads its instance's two matrices, 112 bytes per lane whichever record it is, and those
	// reads then stay off the vector-memory return path, which is what this kernel is short of.
	__shared__ InstRec s_insts[SHADE_LDS_INSTS > 0 ? SHADE_LDS_INSTS : 1];
	const bool insts_in_lds = SHADE_LDS_INSTS > 0 && sc.n_insts <= uint32_t(SHADE_LDS_INSTS);
	if (insts_in_lds) {
		static_assert(sizeof(InstRec) % sizeof(float4) == 0, "instance records are copied in 16-byte pieces");
		constexpr uint32_t PIECES = sizeof(InstRec) / sizeof(float4);
		for (uint32_t k = threadIdx.x; k < sc.n_insts * PIECES; k += SHADE_BLOCK) reinterpret_cast<float4*>(s_insts)[k] = reinterpret_cast<const float4*>(sc.insts)[k];
		__syncthreads();
	}
	bool alive = false, wants_direct = false;
	uint32_t path = 0;
	bool have_path = i < *count;
	if (have_path) path = list[i];
#if PTX_SHADE_SORT
	// The block's paths regrouped by what their vertex will cost, before any of it is computed: paths whose vertex
	// scatters (surface rebuilt, two BSDF samples, the emitter pre-test) first, one material kind after the other, then
	// the cheap ones (missed, met a lamp, last vertex).  A wave then holds paths of one kind and its instructions run
	// with all lanes; which thread shades which path changes nothing in what is computed for it.
	{
		constexpr int WAVES = SHADE_BLOCK / 64, KEYS = 8;
		__shared__ uint32_t s_path[SHADE_BLOCK];
		__shared__ uint32_t s_slot[WAVES * KEYS];
		const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
		uint32_t key = KEYS - 1;   // no path
		if (have_path) {
			const uint32_t word = __float_as_uint(rb.state[path].w);
			const uint2 hb = rb.hit_b[0][path];
			key = 5;               // cheap: nothing scatters
#if PTX_SHADE_SORT == 2
			// variant: by the instance the main ray hit, in seven runs of the (spatially ordered) instance array, so that
			// the rays leaving one part of the scene are neighbours in the next extend round
			key = hb.x != 0xFFFFFFFFu ? min(6u, uint32_t((uint64_t(hb.y) * 7u) / max(sc.n_insts, 1u))) : 6u;
			if (false)
#endif
			if (((word & PATH_CAMERA) || path_depth(word) != 0) && hb.x != 0xFFFFFFFFu && path_depth(word) != 0) {
				const int32_t mat_id = sc.insts[hb.y].material;
				if (mat_id >= 0) {
					const uint32_t type = sh.materials[mat_id].type;
					if (type != PTX_MAT_EMISSIVE) key = type;   // 0 Lambertian, 1 Mirror, 2 Refract, 3 Glass
				}
			}
		}
		uint32_t rank = 0;
		for (uint32_t k = 0; k < KEYS; k++) {
			const unsigned long long m = __ballot(key == k);
			if (lane == 0) s_slot[k * WAVES + wave] = uint32_t(__popcll(m));
			if (key == k) rank = uint32_t(__popcll(m & ((1ull << lane) - 1ull)));
		}
		__syncthreads();
		if (threadIdx.x < 64) {   // exclusive prefix over (key, wave), key-major
			const uint32_t c = lane < WAVES * KEYS ? s_slot[lane] : 0u;
			uint32_t incl = c;
			for (int d = 1; d < 64; d <<= 1) {
				const uint32_t o = __shfl_up(incl, d);
				if (int(lane) >= d) incl += o;
			}
			if (lane < WAVES * KEYS) s_slot[lane] = incl - c;
		}
		__syncthreads();
		s_path[s_slot[key * WAVES + wave] + rank] = have_path ? path : 0xFFFFFFFFu;
		__syncthreads();
		path = s_path[threadIdx.x];
		have_path = path != 0xFFFFFFFFu;
		if (!have_path) path = 0;
	}
#endif
	if (have_path) {
		SHADE_SITE(0);
		const float4 st = rb.state[path];
#if PTX_SHADE_EARLY_LOADS
		// The main ray and its hit are asked for together with the state, before the direct ray's result is dealt
		// with: nearly every listed path needs them, and the two chains of dependent loads then overlap.
		const float4 ma = rb.ray_a[0][path], mb = rb.ray_b[0][path];
		const float4 ha = rb.hit_a[0][path];
		const uint2 hb = rb.hit_b[0][path];
#endif
#if PTX_SHADE_EARLY_LOADS >= 2
		// likewise the slot's pixel and first sample, which seed the vertex's random numbers
		uint32_t slot, g;
		path_to_slot(rb, path, slot, g);
		const SlotInfo si = load_rec(rb.slots + slot);
#endif
		V3f T = v3(st.x, st.y, st.z);
		const uint32_t word = __float_as_uint(st.w);
		uint32_t depth = path_depth(word);
		const bool camera = word & PATH_CAMERA;
		bool lights_written = false;
		// The radiance gathered so far stays in memory; the few vertices that add to it (a camera ray meeting a lamp
		// or the environment, a direct ray that found light) read it, add in the reference's order, and write it back.
		V3f L = v3(0, 0, 0);
		bool L_loaded = false;
		auto add_radiance = [&](V3f x) {
			if (!L_loaded) {
				const float4 l = rb.rad[path];
				L = v3(l.x, l.y, l.z), L_loaded = true;
			}
			L = L + x;
		};

		// ---- the direct ray traced last round: trace(rng, new_in_ray).first (pathtracer.cpp:71-76)
		// `da`, `db`: its hit; the ray itself (`direct_ray()`) is looked at only where its direction or origin matters.
		auto direct_first = [&](const float4 da, const uint2 db, auto direct_ray) {
			V3f first = v3(0, 0, 0);
			if (db.x == 0xFFFFFFFFu) {
				if (sh.n_env_lights) {   // pathtracer.cpp:129-136
					float4 ra, rbb;
					direct_ray(ra, rbb);
					first = env_radiance(sh, v3(ra.w, rbb.x, rbb.y));
				}
			} else {
				// depth 0: {emissive, {}} (pathtracer.cpp:160-163).  Only an emissive surface with an image texture
				// needs the hit rebuilt (for its uv); everything else is decided by the instance's material.
				const int32_t mat_id = sc.insts[db.y].material;
				if (mat_id >= 0) {
					const MatRec dm = sh.materials[mat_id];
					if (dm.type == PTX_MAT_EMISSIVE) {
						if (sh.textures[dm.tex_a].type == PTX_TEX_CONSTANT) {
							first = tex_evaluate(sh, dm.tex_a, 0.0f, 0.0f);
						} else {
							float4 ra, rbb;
							direct_ray(ra, rbb);
							const Ray3 r{ra.x, ra.y, ra.z, ra.w, rbb.x, rbb.y, rbb.z, rbb.w};
							HitRec dh{};
							dh.distance = da.x, dh.t = da.y, dh.u = da.z, dh.v = da.w, dh.prim = db.x, dh.inst = db.y;
							const Surface s = rebuild_surface(sc.insts, sc, sh, r, dh);
							first = bsdf_emission(sh, dm, s.u, s.v);
						}
					}
				}
			}
			return first;
		};
		if (word & PATH_HAS_DIRECT) {
			SHADE_SITE(1);
			const V3f first = direct_first(rb.hit_a[1][path], rb.hit_b[1][path], [&](float4& ra, float4& rbb) { ra = rb.ray_a[1][path], rbb = rb.ray_b[1][path]; });
			const float4 wd = rb.wd[path];
			add_radiance(first * v3(wd.x, wd.y, wd.z));
		}

		// ---- the main ray
		if (!camera && depth == 0) {
			// the scattered ray of the path's last vertex: trace() would return {emissive, {}} for it and
			// sample_indirect_lighting uses `.second` only (pathtracer.cpp:118,163); it may not even have been traced
		} else {
			SHADE_SITE(2);
#if !PTX_SHADE_EARLY_LOADS
			const float4 ma = rb.ray_a[0][path], mb = rb.ray_b[0][path];
			const float4 ha = rb.hit_a[0][path];
			const uint2 hb = rb.hit_b[0][path];
#endif
			const V3f ray_o = v3(ma.x, ma.y, ma.z), ray_d = v3(ma.w, mb.x, mb.y);
			if (hb.x == 0xFFFFFFFFu || (insts_in_lds ? s_insts[hb.y].material : sc.insts[hb.y].material) < 0) {
				// miss: {env, {}} -- only a camera ray's (or a direct ray's) `.first` is ever used
				if (camera && hb.x == 0xFFFFFFFFu && sh.n_env_lights) add_radiance(env_radiance(sh, ray_d));
			} else {
				SHADE_SITE(3);
				const Ray3 r{ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
				HitRec mh{};
				mh.distance = ha.x, mh.t = ha.y, mh.u = ha.z, mh.v = ha.w, mh.prim = hb.x, mh.inst = hb.y;
				// (two calls, not one call on a selected pointer: each then knows which memory it reads)
				Surface s = insts_in_lds ? rebuild_surface(s_insts, sc, sh, r, mh) : rebuild_surface(sc.insts, sc, sh, r, mh);
				const MatRec m = sh.materials[s.material];
				if (!mat_is_sided(m.type) && dot3(s.normal, ray_d) > 0.0f) s.normal = -s.normal;   // pathtracer.cpp:143-145
				const Frame f = frame_from_normal(s.normal);
				const V3f out_dir = unit(f.to_local(ray_o - s.position));   // pathtracer.cpp:153
				// an indirect ray contributes `.second` only; every material but Emissive emits a zero Spectrum
				if (camera && m.type == PTX_MAT_EMISSIVE) add_radiance(bsdf_emission(sh, m, s.u, s.v));
				if (depth != 0 && m.type != PTX_MAT_EMISSIVE) {   // pathtracer.cpp:163
					SHADE_SITE(4);
					// sum_delta_lights (pathtracer.cpp:645-666): shadow rays and contributions for the occlusion pass
					if (DELTA && !mat_is_specular(m.type)) {
						lights_written = true;
						for (uint32_t li = 0; li < sh.n_delta_lights; li++) {
							const size_t item = size_t(path) * sh.n_delta_lights + li;
							const IncomingRec inc = light_incoming(sh.delta_lights[li], s.position);
							const V3f in_dir = f.to_local(inc.direction);
							const V3f att = bsdf_evaluate(sh, m, in_dir, s.u, s.v);   // evaluate(out_dir, in_dir): cosine of in_dir
							if (0.2126f * att.x + 0.7152f * att.y + 0.0722f * att.z == 0.0f) {
								rb.shadow_add[item] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
								continue;
							}
							const V3f sd = unit(inc.direction);   // Ray's constructor
							store_rec(rb.shadow_rays + item, PtxRay{{s.position.x, s.position.y, s.position.z}, {sd.x, sd.y, sd.z}, SH_EPS, inc.distance - SH_EPS});
							const V3f add = T * (att * inc.radiance);
							rb.shadow_add[item] = make_float4(add.x, add.y, add.z, 1.0f);
						}
					}
#if PTX_SHADE_EARLY_LOADS < 2
					uint32_t slot, g;
					path_to_slot(rb, path, slot, g);
					const SlotInfo si = load_rec(rb.slots + slot);
#endif
					PathRng rng;
					rng.start(key0, key1, si.pixel, si.s_begin + k0 + g, path_draws(word));
					// sample_direct_lighting_task4 (pathtracer.cpp:34-78): weight = attenuation [/ pdf]
					{
						const ScatterRec sc1 = bsdf_scatter(sh, m, rng, out_dir, s.u, s.v);
						const V3f in_world = unit(f.to_world(sc1.direction));   // Ray's constructor normalises
						V3f w = sc1.specular ? sc1.attenuation : bsdf_evaluate(sh, m, out_dir, s.u, s.v);   // evaluate(in_dir, out_dir): `in` = out_dir
						if (!sc1.specular) w = w / bsdf_pdf(m, out_dir);                                      // pdf(in_dir, out_dir): `in` = out_dir
						const V3f Wd = T * w;
						// The ray brings back `first`, the emission of the first surface it meets; the path gains first * Wd.
						// Where first is provably zero and Wd finite that is +-0, and L + (+-0) == L: the ray is left out.
						wants_direct = true;
						if (sh.direct_skip && finite3(Wd)) SHADE_SITE(5);
						if (sh.direct_skip && finite3(Wd))
							wants_direct = direct_ray_may_see_light(sc, sh, Ray3{s.position.x, s.position.y, s.position.z, in_world.x, in_world.y, in_world.z, RAY_MINIMUM_BOUND, INFINITY});
						if (wants_direct) {
							SHADE_SITE(6);
							rb.wd[path] = make_float4(Wd.x, Wd.y, Wd.z, 0.0f);
							rb.ray_a[1][path] = make_float4(s.position.x, s.position.y, s.position.z, in_world.x);
							rb.ray_b[1][path] = make_float4(in_world.y, in_world.z, RAY_MINIMUM_BOUND, INFINITY);
						}
					}
					// sample_indirect_lighting (pathtracer.cpp:81-124)
					{
						const ScatterRec sc2 = bsdf_scatter(sh, m, rng, out_dir, s.u, s.v);
						const V3f in_world = unit(f.to_world(sc2.direction));
						V3f w = sc2.specular ? sc2.attenuation : bsdf_evaluate(sh, m, out_dir, s.u, s.v);
						if (!sc2.specular) w = w / bsdf_pdf(m, out_dir);
						T = T * w;
						depth -= 1;
						if (depth != 0) {   // a ray of depth 0 is not traced (see above)
							rb.ray_a[0][path] = make_float4(s.position.x, s.position.y, s.position.z, in_world.x);
							rb.ray_b[0][path] = make_float4(in_world.y, in_world.z, RAY_MINIMUM_BOUND, INFINITY);
						}
					}
					// A path stays in flight for what the next round can still add to it: its scattered ray's vertex
					// (not when that ray has depth 0: nothing of its hit is used), a traced direct ray's emission, or
					// this vertex's delta lights (added to the path's radiance after their occlusion pass).
					alive = depth != 0 || wants_direct || lights_written;
					if (alive) rb.state[path] = make_float4(T.x, T.y, T.z, __uint_as_float(path_pack(depth, rng.draws, wants_direct ? PATH_HAS_DIRECT : 0u)));
				}
			}
		}
		if (DELTA && !lights_written)   // no light takes part at this vertex: the occlusion pass skips the path's entries
			for (uint32_t li = 0; li < sh.n_delta_lights; li++) rb.shadow_add[size_t(path) * sh.n_delta_lights + li] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (L_loaded) rb.rad[path] = make_float4(L.x, L.y, L.z, 0.0f);   // folded into the slot's sum in sample order by k_sum_pass
	}
	block_append2<SHADE_BLOCK / 64>(alive, next_list, next_count, wants_direct, rb.direct, rb.counts + 2, path);
}

// The shadow rays k_shade left, as a ray source / answer sink of the occlusion kernels: item k is light k / n_paths
// of the path list[k % n_paths]; entries whose light does not take part carry no ray.
struct ShadowIO {
	RenderBuffers rb;
	const uint32_t* list;
	const uint32_t* n_paths;
	__device__ __forceinline__ uint64_t count() const { return uint64_t(*n_paths) * rb.n_lights; }
	__device__ __forceinline__ void load(uint32_t out, Ray3& r) const {
		const float4* rp = reinterpret_cast<const float4*>(rb.shadow_rays + out);
		const float4 a = rp[0], b = rp[1];
		r = Ray3{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
	}
	__device__ __forceinline__ bool fetch(uint64_t item, Ray3& r, uint32_t& out) const {
		// light by light over the active paths: neighbouring lanes then hold rays from neighbouring pixels to one light
		const uint32_t n = *n_paths;
		out = list[item % n] * rb.n_lights + uint32_t(item / n);
		if (rb.shadow_add[out].w == 0.0f) return false;
		load(out, r);
		return true;
	}
	__device__ __forceinline__ void store(uint32_t out, const SceneHit& h) const { rb.shadow_occ[out] = h.hit ? 1 : 0; }
};

// Occlusion queries one ray per lane: small batches, scenes the replenishing kernel does not take, reference order.
template<int MODE>
__global__ void __launch_bounds__(256) k_shadow_plain(SceneRec sc, ShadowIO io) {
	const uint64_t item = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (item >= io.count()) return;
	Ray3 r;
	uint32_t out;
	if (!io.fetch(item, r, out)) return;
	Stack st;
	SceneHit h;
	traverse_scene<MODE, true>(sc, r, st, h);
	io.store(out, h);
}

// `if (!shadow.hit) radiance += attenuation * incoming.radiance` (pathtracer.cpp:660-662) for every light in order.
__global__ void __launch_bounds__(256) k_apply_shadows(RenderBuffers rb, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= *count) return;
	const uint32_t path = list[i];
	float4 L = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	bool any = false;
	for (uint32_t li = 0; li < rb.n_lights; li++) {
		const size_t item = size_t(path) * rb.n_lights + li;
		const float4 add = rb.shadow_add[item];
		if (add.w == 0.0f || rb.shadow_occ[item]) continue;
		if (!any) L = rb.rad[path], any = true;
		L.x += add.x, L.y += add.y, L.z += add.z;
	}
	if (any) rb.rad[path] = L;
}

// The samples of one pass added to their slot's sum in sample order, so that the float sum per pixel is
// the reference's sequence of additions: `if (p.valid()) sample.at(px, py) += p;` (pathtracer.cpp:466-470;
// the pdf of the camera sample is 1).
__global__ void __launch_bounds__(256) k_sum_pass(RenderBuffers rb, uint32_t k0) {
	const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
	if (slot >= rb.n_slots) return;
	const uint32_t s_count = rb.slots[slot].s_count;
	float4 sum = rb.sums[slot];
	for (uint32_t g = 0; g < rb.group && k0 + g < s_count; g++) {
		const float4 L = rb.rad[slot_to_path(rb, slot, g)];
		if (finite3(v3(L.x, L.y, L.z))) sum.x += L.x, sum.y += L.y, sum.z += L.z;
	}
	rb.sums[slot] = sum;
}

// Pathtracer::accumulate (pathtracer.cpp:407-427)
__global__ void __launch_bounds__(256) k_flush(RenderBuffers rb, uint32_t n_slots, unsigned long long* __restrict__ accum, uint32_t* __restrict__ samples) {
	const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
	if (slot >= n_slots) return;
	const SlotInfo si = load_rec(rb.slots + slot);
	const float4 sum = rb.sums[slot];
	const float scale = float(1ll << 24);   // n.r * (1ll << 24ll): the integer converts to float
	atomicAdd(&accum[size_t(si.pixel) * 3 + 0], (unsigned long long)(long long)(sum.x * scale));
	atomicAdd(&accum[size_t(si.pixel) * 3 + 1], (unsigned long long)(long long)(sum.y * scale));
	atomicAdd(&accum[size_t(si.pixel) * 3 + 2], (unsigned long long)(long long)(sum.z * scale));
	atomicAdd(&samples[si.pixel], si.s_count);
}

// Between an extend round and its shade launch: the rays that round traced are added to the scene's running totals
// (ptx_get_counters), the list the shade stage is about to fill and the direct-ray list just consumed are emptied.
__global__ void k_round_tally(uint32_t* __restrict__ counts, uint32_t cur, uint32_t main_traced, unsigned long long* __restrict__ tally) {
	if (blockIdx.x || threadIdx.x) return;
	tally[0] += main_traced ? counts[cur] : 0u;
	tally[1] += counts[2];
	counts[cur ^ 1u] = 0u;
	counts[2] = 0u;
}

// Known-answer access to the random stream: one Philox block per thread.
__global__ void k_philox_blocks(const uint32_t* __restrict__ counters, const uint32_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ out) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t r[4];
	philox4x32_10(counters[4 * i], counters[4 * i + 1], counters[4 * i + 2], counters[4 * i + 3], keys[2 * i], keys[2 * i + 1], r);
	out[4 * i] = r[0], out[4 * i + 1] = r[1], out[4 * i + 2] = r[2], out[4 * i + 3] = r[3];
}

// The first `n` unit draws of the path stream (seed, pixel, sample), as k_raygen / k_shade consume them.
__global__ void k_path_units(uint32_t key0, uint32_t key1, uint32_t pixel, uint32_t sample, uint32_t n, float* __restrict__ out) {
	if (blockIdx.x || threadIdx.x) return;
	PathRng rng;
	rng.start(key0, key1, pixel, sample, 0);
	for (uint32_t i = 0; i < n; i++) out[i] = rng.next_unit();
}

static int device_alloc(PtxScene* s, int slot, size_t bytes, void** out) {
	if (s->scratch_bytes[slot] < bytes) {
		if (s->scratch[slot]) PTX_HIP(hipFree(s->scratch[slot]));
		s->scratch[slot] = nullptr, s->scratch_bytes[slot] = 0;
		PTX_HIP(hipMalloc(&s->scratch[slot], bytes));
		s->scratch_bytes[slot] = bytes;
	}
	*out = s->scratch[slot];
	return PTX_OK;
}

}  // namespace ptx

using namespace ptx;

extern "C" int ptx_render_tiles(ptx_scene_t s, const PtxCamera* cam, const PtxTile* tiles, uint32_t n_tiles, uint64_t seed, int64_t* accum_rgb,
                                uint32_t* samples, uint32_t flags) {
	if (!s || !cam || (n_tiles && !tiles) || !accum_rgb || !samples) return fail(PTX_ERR_INVALID, "ptx_render_tiles: null argument");
	if (s->rec.mesh_only) return fail(PTX_ERR_INVALID, "ptx_render_tiles: the scene was created with ptx_scene_create_mesh and has no materials");
	if (!cam->film_width || !cam->film_height) return fail(PTX_ERR_INVALID, "ptx_render_tiles: empty film");
	if (cam->film_max_ray_depth > PATH_MAX_DEPTH) return fail(PTX_ERR_UNSUPPORTED, "ptx_render_tiles: film_max_ray_depth above 8191");
	for (uint32_t i = 0; i < n_tiles; i++) {
		const PtxTile& t = tiles[i];
		if (t.x_begin > t.x_end || t.y_begin > t.y_end || t.s_begin > t.s_end || t.x_end > cam->film_width || t.y_end > cam->film_height)
			return fail(PTX_ERR_INVALID, "ptx_render_tiles: tile outside the film or with reversed bounds");
	}
	if (n_tiles == 0) return PTX_OK;
	PTX_HIP(hipSetDevice(s->device));
	const size_t n_pixels = size_t(cam->film_width) * cam->film_height;
	const bool host = !(flags & PTX_MEM_DEVICE);
	unsigned long long* d_accum = reinterpret_cast<unsigned long long*>(accum_rgb);
	uint32_t* d_samples = samples;
	if (host) {
		void *a = nullptr, *b = nullptr;
		if (int rc = device_alloc(s, 0, n_pixels * 3 * sizeof(int64_t), &a)) return rc;
		if (int rc = device_alloc(s, 1, n_pixels * sizeof(uint32_t), &b)) return rc;
		d_accum = static_cast<unsigned long long*>(a), d_samples = static_cast<uint32_t*>(b);
		PTX_HIP(hipMemcpyAsync(d_accum, accum_rgb, n_pixels * 3 * sizeof(int64_t), hipMemcpyHostToDevice, s->stream));
		PTX_HIP(hipMemcpyAsync(d_samples, samples, n_pixels * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
	}

	CameraRec cr{};
	cr.h = 2.0f * std::tan((cam->vertical_fov / 2.0f) * (3.14159265358979323846264338327950288f / 180.0f));   // camera.cpp:30
	cr.w = cam->aspect_ratio * cr.h;
	cr.film_w = cam->film_width, cr.film_h = cam->film_height, cr.max_depth = cam->film_max_ray_depth;
	std::copy(cam->camera_to_world, cam->camera_to_world + 16, cr.cam_to_world);
	const uint32_t key0 = uint32_t(seed), key1 = uint32_t(seed >> 32);

	// Tiles are processed in chunks whose slots fit the state buffers.
	uint32_t next_tile = 0;
	while (next_tile < n_tiles) {
		std::vector<TileRec> chunk;
		uint64_t n_slots = 0;
		uint32_t max_samples = 0;
		while (next_tile < n_tiles) {
			const PtxTile& t = tiles[next_tile];
			const uint64_t px = uint64_t(t.x_end - t.x_begin) * (t.y_end - t.y_begin);
			if (!chunk.empty() && n_slots + px > MAX_SLOTS_PER_CHUNK) break;
			if (px > 0xFFFFFFFFull - n_slots) return fail(PTX_ERR_UNSUPPORTED, "ptx_render_tiles: a single tile larger than 2^32 pixels");
			if (px && t.s_end > t.s_begin) {
				chunk.push_back(TileRec{t.x_begin, t.y_begin, t.x_end - t.x_begin, t.y_end - t.y_begin, t.s_begin, t.s_end - t.s_begin, uint32_t(n_slots), 0});
				n_slots += px;
				max_samples = std::max(max_samples, t.s_end - t.s_begin);
			}
			next_tile++;
		}
		if (chunk.empty()) continue;
		const uint32_t ns = uint32_t(n_slots);
		// samples of a slot in flight together: as many as keep the paths within the state buffers' budget
		uint32_t group = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(max_samples, MAX_SLOTS_PER_CHUNK / n_slots)));
		if (s->shade.n_delta_lights) {
			// one shadow-ray record per path in flight and light: keep their number within 2^28 (13 GB) and their index in 32 bits
			const uint64_t per_sample = n_slots * uint64_t(s->shade.n_delta_lights);
			if (per_sample >= (uint64_t(1) << 32)) return fail(PTX_ERR_UNSUPPORTED, "ptx_render_tiles: tiles x delta lights exceed 2^32 shadow rays per sample; pass fewer tiles per call");
			group = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(group, (uint64_t(1) << 28) / per_sample)));
		}
		// passes of equal size: 64 samples in flight at most 63 at a time are two passes of 32, not 63 + 1
		group = (max_samples + (max_samples + group - 1) / group - 1) / ((max_samples + group - 1) / group);
		const uint64_t n_paths = n_slots * group;
		RenderBuffers rb{};
		rb.n_slots = ns, rb.group = group;
		void* p = nullptr;
		if (int rc = device_alloc(s, 2, chunk.size() * sizeof(TileRec), &p)) return rc;
		TileRec* d_tiles = static_cast<TileRec*>(p);
		// one allocation for all per-slot and per-path state
		const uint32_t n_lights = s->shade.n_delta_lights;
		const size_t shadow_bytes = size_t(n_lights) * (sizeof(PtxRay) + sizeof(float4) + 1) + 16;   // per path
		const size_t bytes = n_slots * (sizeof(SlotInfo) + sizeof(float4)) + n_paths * (PATH_STATE_BYTES + (n_lights ? shadow_bytes : 0)) + 256;
		if (int rc = device_alloc(s, 3, bytes, &p)) return rc;
		char* base = static_cast<char*>(p);
		auto take = [&](size_t count, size_t elem) {
			char* at = base;
			base += count * elem;
			return at;
		};
		rb.counts = reinterpret_cast<uint32_t*>(take(1, 256));
		rb.state = reinterpret_cast<float4*>(take(n_paths, sizeof(float4)));
		rb.rad = reinterpret_cast<float4*>(take(n_paths, sizeof(float4)));
		rb.wd = reinterpret_cast<float4*>(take(n_paths, sizeof(float4)));
		for (int which = 0; which < 2; which++) {
			rb.ray_a[which] = reinterpret_cast<float4*>(take(n_paths, sizeof(float4)));
			rb.ray_b[which] = reinterpret_cast<float4*>(take(n_paths, sizeof(float4)));
			rb.hit_a[which] = reinterpret_cast<float4*>(take(n_paths, sizeof(float4)));
		}
		rb.sums = reinterpret_cast<float4*>(take(n_slots, sizeof(float4)));
		rb.slots = reinterpret_cast<SlotInfo*>(take(n_slots, sizeof(SlotInfo)));
		rb.hit_b[0] = reinterpret_cast<uint2*>(take(n_paths, sizeof(uint2)));
		rb.hit_b[1] = reinterpret_cast<uint2*>(take(n_paths, sizeof(uint2)));
		rb.active[0] = reinterpret_cast<uint32_t*>(take(n_paths, sizeof(uint32_t)));
		rb.active[1] = reinterpret_cast<uint32_t*>(take(n_paths, sizeof(uint32_t)));
		rb.direct = reinterpret_cast<uint32_t*>(take(n_paths, sizeof(uint32_t)));
		rb.n_lights = n_lights;
		if (n_lights) {
			base += (16 - (reinterpret_cast<uintptr_t>(base) & 15)) & 15;
			rb.shadow_rays = reinterpret_cast<PtxRay*>(base), base += n_paths * n_lights * sizeof(PtxRay);
			rb.shadow_add = reinterpret_cast<float4*>(base), base += n_paths * n_lights * sizeof(float4);
			rb.shadow_occ = reinterpret_cast<uint8_t*>(base);
		}
		PTX_HIP(hipMemcpyAsync(d_tiles, chunk.data(), chunk.size() * sizeof(TileRec), hipMemcpyHostToDevice, s->stream));
		hipLaunchKernelGGL(k_init_slots, dim3(uint32_t(chunk.size())), dim3(256), 0, s->stream, d_tiles, cam->film_width, rb);

		const uint32_t np = uint32_t(n_paths);
		const dim3 block(256), grid_slots((ns + 255) / 256), grid_items((2 * n_paths + 255) / 256);
		// Small batches (a 256x256 film is 65k paths, 8 waves per CU) do not fill the replenishing kernel's
		// waves for long enough to pay for its set-up launches: they keep one ray per lane.
		if (!s->scratch[9]) {   // running totals of the rays the extend rounds trace: main rays, direct rays
			if (int rc = device_alloc(s, 9, 4 * sizeof(unsigned long long), &p)) return rc;
			PTX_HIP(hipMemsetAsync(s->scratch[9], 0, 4 * sizeof(unsigned long long), s->stream));
		}
		unsigned long long* const ray_tally = static_cast<unsigned long long*>(s->scratch[9]);
		const bool reference_order = (flags & PTX_RENDER_REFERENCE_ORDER) != 0;
		const bool trace_unused = (flags & PTX_RENDER_TRACE_UNUSED_RAYS) != 0;
		ShadeTables shade_tables = s->shade;
		if (trace_unused) shade_tables.direct_skip = 0;
		const bool persistent = scene_persistent_ok(s) && !(flags & PTX_NO_PERSISTENT) && !reference_order && np >= (1u << 18);
		const bool persistent_reference = scene_persistent_ok(s) && !(flags & PTX_NO_PERSISTENT) && reference_order && np >= (1u << 18);
		for (uint32_t k = 0; k < max_samples; k += group) {
			PTX_HIP(hipMemsetAsync(rb.counts, 0, 256, s->stream));
			hipLaunchKernelGGL(k_raygen, dim3((np + RAYGEN_BLOCK - 1) / RAYGEN_BLOCK), dim3(RAYGEN_BLOCK), 0, s->stream, rb, k, cr, key0, key1, rb.active[0], rb.counts + 0);
			int cur = 0;
			// A camera ray of depth D has at most D+1 vertices on its main path (depths D..0): the
			// vertex of depth 0 never scatters, so after D+1 rounds no path is alive and every
			// direct ray (written in round r, used at the start of round r+1) has been consumed.
			for (uint32_t round = 0; round <= cam->film_max_ray_depth; round++) {
				// In the last round every main ray is the scattered ray of a path's last vertex (depth 0, not a camera
				// ray): no hit of theirs is used, only the direct rays are traced.
				const bool last_vertex_round = round == cam->film_max_ray_depth && round > 0;
				const RenderIO io{rb, rb.active[cur], rb.counts + cur, (last_vertex_round && !trace_unused) ? 0u : 1u};
				if (persistent) {
					if (int rc = launch_scene_persistent(s, io, 2 * n_paths)) return rc;
				} else if (persistent_reference) {
					if (int rc = (launch_scene_persistent<RenderIO, false, MODE_REFERENCE>(s, io, 2 * n_paths))) return rc;
				} else if (reference_order) {
					hipLaunchKernelGGL(k_extend<MODE_REFERENCE>, grid_items, block, 0, s->stream, s->rec, io);
				} else {
					hipLaunchKernelGGL(k_extend<MODE_CULLED>, grid_items, block, 0, s->stream, s->rec, io);
				}
				// the direct rays just traced are consumed; k_shade lists the next ones and the next active list
				hipLaunchKernelGGL(k_round_tally, dim3(1), dim3(64), 0, s->stream, rb.counts, uint32_t(cur), io.main_rays, ray_tally);
				const bool delta = n_lights != 0;
				const auto shade = delta ? k_shade<true> : k_shade<false>;
				hipLaunchKernelGGL(shade, dim3((np + SHADE_BLOCK - 1) / SHADE_BLOCK), dim3(SHADE_BLOCK), 0, s->stream, s->rec, shade_tables, rb, rb.active[cur], rb.counts + cur, k, key0, key1,
				                   rb.active[cur ^ 1], rb.counts + (cur ^ 1));
				if (delta && round < cam->film_max_ray_depth) {   // the vertex of depth 0 gathers no light
					// the occlusion pass over the shadow rays of the paths just shaded, then their contributions in light order
					const ShadowIO sio{rb, rb.active[cur], rb.counts + cur};
					const uint64_t max_items = n_paths * n_lights;
					if (persistent) {
						if (int rc = (launch_scene_persistent<ShadowIO, true>(s, sio, max_items))) return rc;
					} else if (persistent_reference) {
						if (int rc = (launch_scene_persistent<ShadowIO, true, MODE_REFERENCE>(s, sio, max_items))) return rc;
					} else {
						const dim3 grid_shadow(uint32_t((max_items + 255) / 256));
						if (reference_order) hipLaunchKernelGGL(k_shadow_plain<MODE_REFERENCE>, grid_shadow, block, 0, s->stream, s->rec, sio);
						else hipLaunchKernelGGL(k_shadow_plain<MODE_CULLED>, grid_shadow, block, 0, s->stream, s->rec, sio);
					}
					hipLaunchKernelGGL(k_apply_shadows, dim3((np + 255) / 256), block, 0, s->stream, rb, rb.active[cur], rb.counts + cur);
				}
				cur ^= 1;
			}
			hipLaunchKernelGGL(k_sum_pass, grid_slots, block, 0, s->stream, rb, k);
		}
		hipLaunchKernelGGL(k_flush, grid_slots, block, 0, s->stream, rb, ns, d_accum, d_samples);
		PTX_HIP(hipGetLastError());
		uint64_t chunk_paths = 0;
		for (const TileRec& t : chunk) chunk_paths += uint64_t(t.w) * t.h * t.s_count;
		s->counters.paths += chunk_paths;
	}
#ifdef PTX_PROFILE_SHADE
	{
		unsigned long long h[32] = {}, zero[32] = {};
		PTX_HIP(hipStreamSynchronize(s->stream));
		PTX_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_shade_prof), sizeof(h)));
		PTX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_shade_prof), zero, sizeof(zero)));
		static const char* const site[] = {"listed path", "direct ray's result", "main ray's vertex", "surface rebuilt", "scatter (both samples)", "emitter pre-test", "direct ray written"};
		for (int k = 0; k < 7; k++)
			std::fprintf(stderr, "[shade lanes] %-24s %12llu wave visits, %5.1f lanes, %6.2f %% of listed paths\n", site[k], h[2 * k + 1], double(h[2 * k]) / double(h[2 * k + 1] ? h[2 * k + 1] : 1),
			             100.0 * double(h[2 * k]) / double(h[0] ? h[0] : 1));
	}
#endif
	if (host) {
		PTX_HIP(hipMemcpyAsync(accum_rgb, d_accum, n_pixels * 3 * sizeof(int64_t), hipMemcpyDeviceToHost, s->stream));
		PTX_HIP(hipMemcpyAsync(samples, d_samples, n_pixels * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
	}
	if (!(flags & PTX_ASYNC) || host) PTX_HIP(hipStreamSynchronize(s->stream));
	return PTX_OK;
}

extern "C" int ptx_selftest_philox(int device, const uint32_t* counters4, const uint32_t* keys2, uint32_t n, uint32_t* out4) {
	if (n && (!counters4 || !keys2 || !out4)) return fail(PTX_ERR_INVALID, "ptx_selftest_philox: null argument");
	if (n == 0) return PTX_OK;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(PTX_ERR_NO_DEVICE, "device index out of range");
	PTX_HIP(hipSetDevice(device));
	uint32_t* d = nullptr;
	PTX_HIP(hipMalloc(&d, size_t(n) * 10 * sizeof(uint32_t)));
	hipError_t e = hipMemcpy(d, counters4, size_t(n) * 16, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(d + 4 * size_t(n), keys2, size_t(n) * 8, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_philox_blocks, dim3((n + 255) / 256), dim3(256), 0, nullptr, d, d + 4 * size_t(n), n, d + 6 * size_t(n));
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpy(out4, d + 6 * size_t(n), size_t(n) * 16, hipMemcpyDeviceToHost);
	(void)hipFree(d);
	if (e != hipSuccess) return hip_fail(e, "ptx_selftest_philox");
	return PTX_OK;
}

extern "C" int ptx_selftest_path_units(int device, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float* out) {
	if (n && !out) return fail(PTX_ERR_INVALID, "ptx_selftest_path_units: null argument");
	if (n == 0) return PTX_OK;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(PTX_ERR_NO_DEVICE, "device index out of range");
	PTX_HIP(hipSetDevice(device));
	float* d = nullptr;
	PTX_HIP(hipMalloc(&d, size_t(n) * sizeof(float)));
	hipLaunchKernelGGL(k_path_units, dim3(1), dim3(64), 0, nullptr, uint32_t(seed), uint32_t(seed >> 32), pixel, sample, n, d);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipMemcpy(out, d, size_t(n) * sizeof(float), hipMemcpyDeviceToHost);
	(void)hipFree(d);
	if (e != hipSuccess) return hip_fail(e, "ptx_selftest_path_units");
	return PTX_OK;
}

// scene_persistent.hip.h -- closest-hit and occlusion queries against a two-level scene (instances over meshes
// and analytic spheres) with wave-level ray replenishment and phase parking.
//
// Aggregate::hit over BVH<Instance> (aggregate.h:41-43) -> Instance::hit (instance.h:32-40) ->
// Tri_Mesh::hit / Shapes::Sphere::hit, in the culled order (nearest child first, subtrees beyond the
// running closest hit skipped, order-free tie rules -- see traverse.hip.h).
//
// One ray per lane and one `while` loop per level leaves most lanes of a wave idle: the lanes
// are at different levels, in different instances, some in leaves and some on interior nodes
// (20 % of the lanes active on the Cornell-box scenes).  Here a wave is again a pool of 64
// traversal slots refilled from a global counter, and a lane is always in exactly one state:
//
//   WALK   on an interior node of the top-level tree or of the current instance's mesh tree
//          (same code for both: the lane's current ray, reciprocals, node array and running
//          hit are whatever level it is on)
//   LEAF   holds a parked triangle leaf of the current mesh (and may go on walking meanwhile)
//   ENTER  stands on a top-level leaf: has to move into an instance (ray to object space,
//          root box test or sphere test)
//   EXIT   finished the current instance's tree: has to carry the hit back to world space
//
// Each loop iteration executes the phase most lanes are ready for, and the lanes in other states wait
// their turn, so every phase's code runs with many lanes active.  ENTER, LEAF and EXIT are also the
// stages a ray goes through, in this order, in a mesh whose tree is a single leaf (a quad): a stage
// runs behind the one before it within the same iteration when that one has made enough lanes ready.
#pragma once
#include <cstdio>
#include <cstdlib>

#include "scene.h"
#include "traverse.hip.h"

namespace ptx {

constexpr int SCENE_LDS_SLOTS = 8;
constexpr unsigned long long SCENE_WAVE_CHUNK = 128;
#ifndef PTX_SCENE_PARTS
#define PTX_SCENE_PARTS 8   // one per XCD; measured against a single counter: renders 0-2.5 % faster, same bits
#endif
constexpr unsigned SCENE_PART_COUNTER0 = 32, SCENE_PART_COUNTER_STRIDE = 8;   // in 8-byte words of the counter block: bytes 256, 320, ... (a 64-byte line each)
constexpr int SCENE_WALK_STEPS = 6;   // box steps per WALK phase
constexpr int SCENE_REFILL_MIN_IDLE = 8;
// Lanes that make a stage of the loop worth running although it is not the iteration's phase (see the loop; 65: never).
#define PTX_SCENE_THRESHOLD(NAME, VALUE) constexpr int SCENE_T_##NAME = VALUE;
#ifndef PTX_SCENE_T_ENTER_COLD
#define PTX_SCENE_T_ENTER_COLD 65
#endif
#ifndef PTX_SCENE_T_WALK_COLD
#define PTX_SCENE_T_WALK_COLD 65
#endif
#ifndef PTX_SCENE_T_ENTER_AFTER_WALK
#define PTX_SCENE_T_ENTER_AFTER_WALK 65
#endif
#ifndef PTX_SCENE_T_TRI_COLD
#define PTX_SCENE_T_TRI_COLD 65
#endif
#ifndef PTX_SCENE_T_TRI_AFTER_ENTER
#define PTX_SCENE_T_TRI_AFTER_ENTER 8
#endif
#ifndef PTX_SCENE_T_TRI_AFTER_WALK
#define PTX_SCENE_T_TRI_AFTER_WALK 65
#endif
#ifndef PTX_SCENE_T_EXIT_COLD
#define PTX_SCENE_T_EXIT_COLD 65
#endif
#ifndef PTX_SCENE_T_EXIT_AFTER_TRI
#define PTX_SCENE_T_EXIT_AFTER_TRI 8        // behind a triangle stage that itself ran behind another stage
#endif
#ifndef PTX_SCENE_T_EXIT_AFTER_OWN_TRI
#define PTX_SCENE_T_EXIT_AFTER_OWN_TRI 24   // behind a triangle stage that was the iteration's phase
#endif
#ifndef PTX_SCENE_T_EXIT_AFTER_WALK
#define PTX_SCENE_T_EXIT_AFTER_WALK 65
#endif
PTX_SCENE_THRESHOLD(ENTER_COLD, PTX_SCENE_T_ENTER_COLD)
PTX_SCENE_THRESHOLD(WALK_COLD, PTX_SCENE_T_WALK_COLD)
PTX_SCENE_THRESHOLD(ENTER_AFTER_WALK, PTX_SCENE_T_ENTER_AFTER_WALK)
PTX_SCENE_THRESHOLD(TRI_COLD, PTX_SCENE_T_TRI_COLD)
PTX_SCENE_THRESHOLD(TRI_AFTER_ENTER, PTX_SCENE_T_TRI_AFTER_ENTER)
PTX_SCENE_THRESHOLD(TRI_AFTER_WALK, PTX_SCENE_T_TRI_AFTER_WALK)
PTX_SCENE_THRESHOLD(EXIT_COLD, PTX_SCENE_T_EXIT_COLD)
PTX_SCENE_THRESHOLD(EXIT_AFTER_TRI, PTX_SCENE_T_EXIT_AFTER_TRI)
PTX_SCENE_THRESHOLD(EXIT_AFTER_OWN_TRI, PTX_SCENE_T_EXIT_AFTER_OWN_TRI)
PTX_SCENE_THRESHOLD(EXIT_AFTER_WALK, PTX_SCENE_T_EXIT_AFTER_WALK)
constexpr uint32_t SCENE_ITERATION_CAP = 1u << 24;   // per wave; ordinary launches stay below 2^20
// Experiment switches (measured 8-14 % slower: the extra kilobyte of LDS costs a block per CU): the first PTX_SCENE_TOP_LDS node records of one tree are kept
// in LDS -- the instance-level tree's (PTX_SCENE_TOP_LDS_LEVEL 0) or those of the mesh most instances refer to (1).
#ifndef PTX_SCENE_TOP_LDS
#define PTX_SCENE_TOP_LDS 0
#endif
#ifndef PTX_SCENE_TOP_LDS_LEVEL
#define PTX_SCENE_TOP_LDS_LEVEL 0
#endif

// Where rays come from and where hits go.  fetch() returns false for an item that carries no ray.
struct BatchIO {   // ptx_trace_closest on a general scene
	const PtxRay* rays;
	uint64_t n;
	PtxHit* hits;
	float* bary;
	uint32_t* inst;
	__device__ __forceinline__ uint64_t count() const { return n; }
	__device__ __forceinline__ bool fetch(uint64_t item, Ray3& r, uint32_t& out) const {
		const float4* rp = reinterpret_cast<const float4*>(rays + item);
		const float4 a = rp[0], b = rp[1];
		r = Ray3{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
		out = uint32_t(item);
		return true;
	}
	__device__ __forceinline__ void load(uint32_t out, Ray3& r) const {
		const float4* rp = reinterpret_cast<const float4*>(rays + out);
		const float4 a = rp[0], b = rp[1];
		r = Ray3{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
	}
	__device__ __forceinline__ void store(uint32_t out, const SceneHit& h) const {
		PtxHit o;
		o.distance = h.distance, o.prim = h.hit ? h.prim : PTX_NO_HIT;
		hits[out] = o;
		if (bary) reinterpret_cast<float2*>(bary)[out] = h.hit ? make_float2(h.u, h.v) : make_float2(0.0f, 0.0f);
		if (inst) inst[out] = h.hit ? h.inst : PTX_NO_HIT;
	}
};

struct BatchAnyIO {   // ptx_trace_any on a general scene: `scene.hit(shadow_ray).hit` (pathtracer.cpp:657-660)
	const PtxRay* rays;
	uint64_t n;
	uint8_t* occluded;
	__device__ __forceinline__ uint64_t count() const { return n; }
	__device__ __forceinline__ void load(uint32_t out, Ray3& r) const {
		const float4* rp = reinterpret_cast<const float4*>(rays + out);
		const float4 a = rp[0], b = rp[1];
		r = Ray3{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
	}
	__device__ __forceinline__ bool fetch(uint64_t item, Ray3& r, uint32_t& out) const {
		out = uint32_t(item);
		load(out, r);
		return true;
	}
	__device__ __forceinline__ void store(uint32_t out, const SceneHit& h) const { occluded[out] = h.hit ? 1 : 0; }
};

// The rays a first pass set aside, as the ray source of the second (zero-aware) pass: item k is ray list[k] of `io`.
template<typename IO>
struct ListIO {
	IO io;
	const uint32_t* list;
	const uint32_t* n;
	__device__ __forceinline__ uint64_t count() const { return *n; }
	__device__ __forceinline__ void load(uint32_t out, Ray3& r) const { io.load(out, r); }
	__device__ __forceinline__ bool fetch(uint64_t item, Ray3& r, uint32_t& out) const {
		out = list[item];
		io.load(out, r);
		return true;
	}
	__device__ __forceinline__ void store(uint32_t out, const SceneHit& h) const { io.store(out, h); }
};

// Registers, not LDS, decide how many waves share a SIMD here.  With the world ray (while the lane is
// inside an instance) and the once-per-ray bookkeeping kept in LDS the kernel fits 96 VGPRs without
// spilling: five waves per SIMD, five 31 KB blocks per CU.  (At the 117 VGPRs of the all-in-registers
// version only four fit; the same render took 11 % longer.)
// ANY: occlusion queries.  A ray is finished by the first primitive it hits inside its distance bounds, subtrees
// are skipped against the ray's far bound instead of a running closest hit, and only the hit flag is stored.
// ZERO: the second pass, over the rays the first one set aside: world or object-space rays with direction components
// that are exactly zero are walked with the zero-aware slab tests of traverse.hip.h (a kernel of its own, so the
// first pass carries none of that code).
// MODE: MODE_CULLED as described above.  MODE_REFERENCE is the reference's own walk -- every node
// whose ancestors pass the box test, left before right, nothing skipped for distance (bvh.cpp:272-375), Trace::min
// folded in array order at both levels (trace.h:28-36) -- in the same lane machine: a lane that reaches a triangle
// leaf does not walk on (no parking: its next step depends on that leaf's fold), it waits for the wave's next
// triangle phase and pops afterwards.
template<typename IO, bool ANY = false, bool ZERO = false, int MODE = MODE_CULLED>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) k_scene_persistent(SceneRec sc, IO io, unsigned long long* __restrict__ next_item, uint32_t* __restrict__ slow_list,
                                                          uint32_t* __restrict__ slow_count, const int refill_min, const unsigned long long wave_chunk, const int walk_steps) {
	const unsigned lane = threadIdx.x & 63u;
	const unsigned long long lanes_below = (1ull << lane) - 1ull;
	const uint64_t n_items = io.count();
	if (ZERO && n_items == 0) return;   // nothing was set aside: the usual case

	constexpr int STOP = (MODE == MODE_CULLED && !ZERO) ? PTX_SCENE_TOP_LDS : 0;
	__shared__ float4 lds_tree_top[STOP > 0 ? STOP * 4 : 1];
	const uint32_t lds_tree_base = PTX_SCENE_TOP_LDS_LEVEL ? sc.meshes[sc.lds_mesh].node_base : sc.top_node_base;
	if (STOP > 0) {
		// the node array is padded by 128 records on the host, so this read stays inside it for any tree
		for (int i = threadIdx.x; i < STOP * 4; i += 256) lds_tree_top[i] = reinterpret_cast<const float4*>(sc.nodes + lds_tree_base)[i];
		__syncthreads();
	}
	__shared__ uint32_t lds_ref[SCENE_LDS_SLOTS * 256];
	__shared__ float lds_tnear[SCENE_LDS_SLOTS * 256];
	Stack spill;
	LdsStack<SCENE_LDS_SLOTS, 256> st{lds_ref + threadIdx.x, lds_tnear + threadIdx.x, &spill};

	bool have_ray = false;
	// Per-lane values touched once or twice per ray wait in LDS, not in registers: where the ray's
	// result goes, the top-level leaf being worked through while the lane is inside one of its
	// instances, and the payload of the best hit so far (only its distance steers the walk).
	__shared__ uint32_t lds_ledger[7 * 256];
	uint32_t* const ledger = lds_ledger + threadIdx.x;
	enum { L_OUT = 0, L_TOP_LEAF = 256, L_T = 2 * 256, L_PRIM = 3 * 256, L_INST = 4 * 256, L_U = 5 * 256, L_V = 6 * 256 };
	// Current ray: the world ray at the top level, the object-space ray inside an instance (the world
	// ray then waits in LDS: registers, not LDS, limit how many waves fit a SIMD here).
	Ray3 cr{};
	RayInv cinv{};
	__shared__ float4 lds_world_ray[2 * 256];
	uint32_t level = 0;         // 0: top-level tree, 1: inside instance `cur_inst`
	uint32_t ref = REF_DONE;
	uint32_t pend = 0u;         // parked triangle leaf (level 1 only)
	int sp = 0, base = 0;       // base: stack height at which the current level's walk is finished
	uint32_t cnode_base = sc.top_node_base, ctri_base = 0u;   // current tree's first node / first triangle in the scene-wide arrays
	bool best_hit = false;      // Aggregate-level running Trace: hit flag and distance (the rest is in the ledger)
	float best_distance = 0.0f;
	MeshHit mh{};               // running Trace of the current instance's mesh
	bool pool_empty = false;
	unsigned long long chunk_next = 0, chunk_end = 0;
	// The item order is cut into PARTS contiguous parts, each with a claim counter of its own.  Blocks b and b + 8
	// run on the same XCD, so with part = b % 8 the rays in flight on one XCD are neighbours in the item order
	// rather than a sample of everything in flight on the chip.  (The second pass has few items: one part.)
	constexpr unsigned PARTS = ZERO ? 1u : unsigned(PTX_SCENE_PARTS);
	const unsigned long long part_size = ((n_items + PARTS - 1) / PARTS + 63ull) & ~63ull;
	unsigned part = blockIdx.x % PARTS, parts_dry = 0;
#ifdef PTX_PROFILE_LANES
	unsigned long long prof[9] = {};
#endif

	auto in_range = [](const Ray3& x) { return ZERO ? ray_in_fast_range_z(x) : ray_in_fast_range(x); };
	auto rcp = [](float d) { return ZERO ? slab_rcp(d) : refined_rcp(d); };
	// At the top level a leaf reference names the instance to visit next (its first) and how many of the
	// leaf's instances are left, that one included.  Empty leaves are skipped.
	auto settle_top = [&]() {
		while ((ref & REF_LEAF) && ref != REF_DONE && (ref & 15u) == 0u) pop_ref<MODE, ANY>(cr, st, 0, sp, ref, best_hit, best_distance);
	};
	// The next instance of the top-level leaf `leaf`, or the next pending top-level subtree.
	auto next_top = [&](uint32_t leaf) {
		if (level == 1u) {   // back from an instance: the world ray returns from LDS
			const float4 a = lds_world_ray[threadIdx.x], b = lds_world_ray[256 + threadIdx.x];
			cr = Ray3{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
			cinv.yx = rcp(cr.dx), cinv.yy = rcp(cr.dy), cinv.yz = rcp(cr.dz);
		}
		level = 0, cnode_base = sc.top_node_base, base = 0;
		if ((leaf & 15u) > 1u) {
			ref = leaf + 15u;   // first + 1, count - 1: ENTER again, for the leaf's next instance
		} else {
			pop_ref<MODE, ANY>(cr, st, 0, sp, ref, best_hit, best_distance);
			settle_top();
		}
	};
	// Trace::min at the Aggregate level: in the reference order as it stands (the candidate replaces the running hit
	// unless that is strictly closer: instances are folded in array order), in the culled order with the order-free
	// tie rule (larger instance index wins a tie).
	auto fold_world = [&](uint32_t inst, bool hit, float dist, float t, uint32_t prim, float u, float v) {
		if (hit) {
			bool keep = MODE == MODE_REFERENCE ? (best_hit && best_distance < dist) : (best_hit && !(dist < best_distance));
			if (MODE != MODE_REFERENCE && keep && dist == best_distance) keep = !(inst > ledger[L_INST]);
			if (!keep) {
				best_hit = true, best_distance = dist;
				ledger[L_T] = __float_as_uint(t), ledger[L_PRIM] = prim, ledger[L_INST] = inst, ledger[L_U] = __float_as_uint(u), ledger[L_V] = __float_as_uint(v);
			}
		} else if (!best_hit) {
			best_distance = 0.0f;   // Trace::min of two misses is a default Trace (trace.h:35)
		}
	};

	// Instance::hit's tail (instance.h:36-39) for a lane whose walk of the current instance's mesh is finished:
	// Trace::transform(T, iT.T()) recomputes the distance from the transformed hit point and origin (trace.h:38-43).
	auto leave_instance = [&]() {
		const uint32_t leaf = ledger[L_TOP_LEAF], cur_inst = (leaf & REF_INDEX_MASK) >> 4;
		const InstRec& in = sc.insts[cur_inst];
		float dist = mh.distance;
		if (!ANY && mh.hit && in.has_transform) {
			const float px = cr.ox + cr.dx * mh.t, py = cr.oy + cr.dy * mh.t, pz = cr.oz + cr.dz * mh.t;
			float wx, wy, wz, qx, qy, qz;
			mat_point(in.T, px, py, pz, wx, wy, wz);
			mat_point(in.T, cr.ox, cr.oy, cr.oz, qx, qy, qz);
			dist = norm3(wx - qx, wy - qy, wz - qz);
		}
		if (!ANY) fold_world(cur_inst, mh.hit, dist, mh.t, mh.prim, mh.u, mh.v);   // ANY: a lane gets here only without a hit
		next_top(leaf);
	};

	// Exit condition every wave reaches: the pool runs dry and its lanes finish.  As a backstop against
	// a scheduling mistake, a wave that has looped far longer than any workload needs hands its rays in
	// flight, and whatever it still claims, to the plain kernel instead of walking them.
	uint32_t iterations = 0;
	bool bailed = false;
	for (;;) {
		if (++iterations == SCENE_ITERATION_CAP) {
			bailed = true;
			if (have_ray) slow_list[atomicAdd(slow_count, 1u)] = ledger[L_OUT];
			have_ray = false, ref = REF_DONE, pend = 0u, level = 0u;
		}
		// ---- replenish dry slots
		const unsigned long long dry = __ballot(!have_ray);
		const int n_dry = __popcll(dry);
		if (n_dry == 64 && pool_empty) break;
		if (!pool_empty && n_dry >= refill_min) {
			if (PARTS == 1) {
				if (chunk_next >= chunk_end) {
					unsigned long long b = 0;
					if (lane == 0) b = atomicAdd(next_item, wave_chunk);
					b = __shfl(b, 0);
					chunk_next = b, chunk_end = b + wave_chunk < n_items ? b + wave_chunk : n_items;
					if (b >= n_items) chunk_next = chunk_end = n_items;
				}
			} else if (chunk_next >= chunk_end) {
				// Claim the next chunk of this wave's part of the item order; when the part is dry, move on to the
				// next one and help there, until all of them are dry.
				for (;;) {
					const unsigned long long lo = part_size * part < n_items ? part_size * part : n_items;
					const unsigned long long hi = part_size * (part + 1) < n_items ? part_size * (part + 1) : n_items;
					unsigned long long b = 0;
					if (lane == 0) b = atomicAdd(next_item + SCENE_PART_COUNTER0 + SCENE_PART_COUNTER_STRIDE * part, wave_chunk);
					b = lo + __shfl(b, 0);
					if (b < hi) {
						chunk_next = b, chunk_end = b + wave_chunk < hi ? b + wave_chunk : hi;
						break;
					}
					if (++parts_dry == PARTS) {
						pool_empty = true, chunk_next = chunk_end = 0;
						break;
					}
					part = (part + 1) % PARTS;
				}
			}
			const unsigned long long b = chunk_next;
			const unsigned long long take = (unsigned long long)n_dry < chunk_end - chunk_next ? (unsigned long long)n_dry : chunk_end - chunk_next;
			chunk_next += take;
			if (PARTS == 1 && chunk_next >= n_items) pool_empty = true;
			if (!have_ray) {
				const uint64_t item = b + __popcll(dry & lanes_below);
				uint32_t out_index;
				if (item < b + take && io.fetch(item, cr, out_index)) {
					if (!bailed && in_range(cr)) {
						have_ray = true;
						ledger[L_OUT] = out_index;
						// Aggregate(BVH<Instance>)::hit starts from hit = false, distance = inf (bvh.cpp:257-259)
						best_hit = false, best_distance = INFINITY;
						ledger[L_T] = 0u, ledger[L_PRIM] = 0xFFFFFFFFu, ledger[L_INST] = 0xFFFFFFFFu, ledger[L_U] = 0u, ledger[L_V] = 0u;
						cinv.yx = rcp(cr.dx), cinv.yy = rcp(cr.dy), cinv.yz = rcp(cr.dz);
						level = 0, cnode_base = sc.top_node_base, sp = 0, base = 0, pend = 0u;
						float tn;
						ref = slab_test<true, ZERO>(sc.top_min[0], sc.top_min[1], sc.top_min[2], sc.top_max[0], sc.top_max[1], sc.top_max[2], cr, cinv, tn) ? sc.top_ref : REF_DONE;
						settle_top();
					} else {
						slow_list[atomicAdd(slow_count, 1u)] = out_index;   // finished by the one-ray-per-lane kernel with plain divisions
					}
				}
			}
		}

		// ---- pick ONE phase for this iteration: the one most lanes are ready for.  Lanes in the other
		// states wait (a waiting lane costs nothing; an instruction issued for a few lanes costs as much
		// as one issued for all 64), so each phase's code runs with as many lanes as the wave can muster.
		// a lane standing on a triangle leaf with its parking slot free parks it and takes its next
		// pending subtree (empty leaves are dropped on the way)
		// (culled order only: in the reference order the lane stays on its leaf until the triangle phase has folded it)
		while (MODE == MODE_CULLED && have_ray && level == 1u && (ref & REF_LEAF) && ref != REF_DONE && pend == 0u) {
			if (ref & 15u) pend = ref;
			pop_ref<MODE, ANY>(cr, st, base, sp, ref, mh.hit, mh.distance);
		}
		const bool can_walk = have_ray && !(ref & REF_LEAF);
		const bool has_parked = MODE == MODE_CULLED ? (have_ray && pend != 0u) : (have_ray && level == 1u && (ref & REF_LEAF) && ref != REF_DONE);
		const bool want_enter = have_ray && level == 0u && (ref & REF_LEAF) && ref != REF_DONE;
		const bool want_exit = have_ray && level == 1u && ref == REF_DONE && pend == 0u;
		const int n_walk = __popcll(__ballot(can_walk)), n_tri = __popcll(__ballot(has_parked));
		const int n_enter = __popcll(__ballot(want_enter)), n_exit = __popcll(__ballot(want_exit));
		const int n_max = max(max(n_walk, n_tri), max(n_enter, n_exit));
		enum { P_NONE, P_WALK, P_TRI, P_EXIT, P_ENTER };
		const int phase = n_max == 0 ? P_NONE : n_walk == n_max ? P_WALK : n_tri == n_max ? P_TRI : n_exit == n_max ? P_EXIT : P_ENTER;
#ifdef PTX_PROFILE_LANES
		prof[0]++;
#endif
		// (P_NONE: nothing to execute -- at most some lanes whose ray finished at once: it missed the scene's box)
		//
		// The four phases are the stages of a cycle a ray goes round -- walk the top-level tree, enter an instance (and
		// walk its tree), test the leaves' triangles, leave the instance -- and the loop body runs them in this order.  The iteration's phase always runs.  Another stage runs as well when enough lanes are ready for it
		// (its instructions cost the same for one lane as for 64): SCENE_T_<stage>_COLD lanes that were ready when the
		// iteration began, or SCENE_T_<stage>_AFTER_<earlier stage> lanes once that earlier stage has run in this
		// iteration and made lanes ready (65: never).  A mesh whose whole tree is one leaf -- a wall of the Cornell
		// box: two triangles -- then takes a ray through enter, triangles and leave in one iteration.  Every lane that
		// is ready for a stage joins it, whichever stage made it ready.
		bool ran_enter = false, ran_walk = false, ran_tri = false;
		{
			const bool run = phase == P_WALK || n_walk >= SCENE_T_WALK_COLD;
			if (run) {
				// ---- WALK: box-test steps on whichever tree the lane is in
				ran_walk = true;
#ifdef PTX_PROFILE_LANES
				prof[1]++, prof[2] += __popcll(__ballot(have_ray && !(ref & REF_LEAF)));
#endif
				// Several steps in a row: choosing a phase costs about a third of a box step, and a wave that
				// walks now mostly walks next time too.  Lanes that reach a leaf on the way sit out the rest.
				// (Measured on two scenes: 6 steps -9 % and -17 % time against 1; parking leaves between the
				// steps or leaving early when half the walkers are done both lose to the plain loop.)
				for (int step = 0; step < walk_steps; step++) {
					if (have_ray && !(ref & REF_LEAF)) {
						const bool run_hit = level ? mh.hit : best_hit;
						const float run_dist = level ? mh.distance : best_distance;
						// (both orders decide with the approximate slab test, exact where it cannot tell: only the decisions matter)
						if (STOP > 0) {
							float4 n0, n1, n2;
							uint2 kids;
							const uint32_t index = ref & REF_INDEX_MASK;
							if (cnode_base == lds_tree_base && index < uint32_t(STOP)) fetch_node_lds(lds_tree_top, index, n0, n1, n2, kids);
							else fetch_node(sc.nodes + cnode_base + index, n0, n1, n2, kids);
							node_decide<MODE, ANY, true, false, ZERO, true>(n0, n1, n2, kids, cr, cinv, st, base, sp, ref, run_hit, run_dist);
						} else {
							node_step<MODE, ANY, true, false, ZERO, true>(sc.nodes + cnode_base, cr, cinv, st, base, sp, ref, run_hit, run_dist, nullptr);
						}
						if (level == 0u) settle_top();
					}
				}
			}
		}
		bool run_enter = phase == P_ENTER || n_enter >= SCENE_T_ENTER_COLD;
		if (!run_enter && ran_walk && SCENE_T_ENTER_AFTER_WALK <= 64)
			run_enter = __popcll(__ballot(have_ray && level == 0u && (ref & REF_LEAF) && ref != REF_DONE)) >= SCENE_T_ENTER_AFTER_WALK;
		if (run_enter) {
			// ---- enter an instance (instance.h:32-35): ray to object space, then the sphere test or the
			// mesh's root box
			ran_enter = true;
#ifdef PTX_PROFILE_LANES
			prof[7]++, prof[8] += n_enter;
#endif
			if (have_ray && level == 0u && (ref & REF_LEAF) && ref != REF_DONE) {
				const uint32_t leaf = ref, cur_inst = (leaf & REF_INDEX_MASK) >> 4;
				const InstRec& in = sc.insts[cur_inst];
				Ray3 r = cr;
				if (in.has_transform) r = ray_transform(in.iT, cr);
				if (in.mesh < 0) {
					float t;
					const bool hit = sphere_test(in.radius, r, t);
					float dist = t;
					if (!ANY && hit && in.has_transform) {
						const float px = r.ox + r.dx * t, py = r.oy + r.dy * t, pz = r.oz + r.dz * t;
						float wx, wy, wz, qx, qy, qz;
						mat_point(in.T, px, py, pz, wx, wy, wz);
						mat_point(in.T, r.ox, r.oy, r.oz, qx, qy, qz);
						dist = norm3(wx - qx, wy - qy, wz - qz);
					}
					if (ANY && hit) {
						best_hit = true, ref = REF_DONE, sp = 0;
					} else {
						fold_world(cur_inst, hit, dist, t, 0u, 0.0f, 0.0f);
						next_top(leaf);
					}
				} else {
					const MeshRec& mesh = sc.meshes[in.mesh];
					if (mesh.is_list || !mesh.tame || !in_range(r)) {
						// outside what this kernel handles exactly: hand the whole ray to the plain kernel
						slow_list[atomicAdd(slow_count, 1u)] = ledger[L_OUT];
						have_ray = false, ref = REF_DONE, pend = 0u, level = 0u;
					} else {
						// BVH<Triangle>::hit (bvh.cpp:246-270): running Trace starts at hit = false, distance = inf
						mh.hit = false, mh.distance = INFINITY, mh.prim = 0xFFFFFFFFu, mh.t = 0.0f, mh.u = mh.v = 0.0f;
						RayInv rinv;
						rinv.yx = rcp(r.dx), rinv.yy = rcp(r.dy), rinv.yz = rcp(r.dz);
						float tn;
						if (slab_test<true, ZERO>(mesh.root_min[0], mesh.root_min[1], mesh.root_min[2], mesh.root_max[0], mesh.root_max[1], mesh.root_max[2], r, rinv, tn)) {
							lds_world_ray[threadIdx.x] = make_float4(cr.ox, cr.oy, cr.oz, cr.dx);
							lds_world_ray[256 + threadIdx.x] = make_float4(cr.dy, cr.dz, cr.tmin, cr.tmax);
							ledger[L_TOP_LEAF] = leaf;
							cr = r, cinv = rinv;
							level = 1u, base = sp, ref = mesh.root_ref;
							cnode_base = mesh.node_base, ctri_base = mesh.tri_base;
						} else {
							fold_world(cur_inst, false, 0.0f, 0.0f, 0u, 0.0f, 0.0f);
							next_top(leaf);
						}
					}
				}
			}
		}
		{
			bool run = phase == P_TRI || n_tri >= SCENE_T_TRI_COLD;
			const int need = min(ran_enter ? SCENE_T_TRI_AFTER_ENTER : 65, ran_walk ? SCENE_T_TRI_AFTER_WALK : 65);
			if (!run && need <= 64) {
				// lanes that have just reached a leaf park it, as at the top of the loop
				while (MODE == MODE_CULLED && have_ray && level == 1u && (ref & REF_LEAF) && ref != REF_DONE && pend == 0u) {
					if (ref & 15u) pend = ref;
					pop_ref<MODE, ANY>(cr, st, base, sp, ref, mh.hit, mh.distance);
				}
				const bool ready = MODE == MODE_CULLED ? (have_ray && pend != 0u) : (have_ray && level == 1u && (ref & REF_LEAF) && ref != REF_DONE);
				run = __popcll(__ballot(ready)) >= need;
			}
			if (run) {
				// ---- triangle stage: every parked leaf (reference order: every lane standing on a leaf of a mesh)
				ran_tri = true;
				const bool at_leaf = MODE == MODE_CULLED ? (have_ray && pend != 0u) : (have_ray && level == 1u && (ref & REF_LEAF) && ref != REF_DONE);
#ifdef PTX_PROFILE_LANES
				prof[3]++, prof[4] += __popcll(__ballot(at_leaf));
#endif
				if (at_leaf) {
					if (MODE == MODE_CULLED) {
						mesh_leaf<MODE, ANY, false>(sc.tris + ctri_base, (pend & REF_INDEX_MASK) >> 4, pend & 15u, cr, mh, nullptr);
						pend = 0u;
						if (ANY && mh.hit) best_hit = true, level = 0u, ref = REF_DONE, sp = 0;   // occluded: the ray is finished
					} else {
						mesh_leaf<MODE, ANY, false>(sc.tris + ctri_base, (ref & REF_INDEX_MASK) >> 4, ref & 15u, cr, mh, nullptr);
						if (ANY && mh.hit) best_hit = true, level = 0u, ref = REF_DONE, sp = 0;   // occluded: the ray is finished
						else pop_ref<MODE, ANY>(cr, st, base, sp, ref, mh.hit, mh.distance);
					}
				}
			}
		}
		{
			bool run = phase == P_EXIT || n_exit >= SCENE_T_EXIT_COLD;
			const int need = min(ran_tri ? (phase == P_TRI ? SCENE_T_EXIT_AFTER_OWN_TRI : SCENE_T_EXIT_AFTER_TRI) : 65, ran_walk ? SCENE_T_EXIT_AFTER_WALK : 65);
			if (!run && need <= 64) run = __popcll(__ballot(have_ray && level == 1u && ref == REF_DONE && pend == 0u)) >= need;
			if (run) {
				// ---- leave an instance: Instance::hit's tail (instance.h:36-39): Trace::transform(T, iT.T())
				// recomputes the distance from the transformed hit point and origin (trace.h:38-43)
				const bool leaving = have_ray && level == 1u && ref == REF_DONE && pend == 0u;
#ifdef PTX_PROFILE_LANES
				prof[5]++, prof[6] += __popcll(__ballot(leaving));
#endif
				if (leaving) leave_instance();
			}
		}

		if (have_ray && level == 0u && ref == REF_DONE) {
			SceneHit best;
			best.hit = best_hit, best.distance = best_distance;
			best.t = __uint_as_float(ledger[L_T]), best.prim = ledger[L_PRIM], best.inst = ledger[L_INST];
			best.u = __uint_as_float(ledger[L_U]), best.v = __uint_as_float(ledger[L_V]);
			io.store(ledger[L_OUT], best);
			have_ray = false;
		}
	}
#ifdef PTX_PROFILE_LANES
	if (lane == 0)
		for (int k = 0; k < 9; k++) atomicAdd(next_item + 16 + k, prof[k]);   // bytes 128.. of the counter block (0: item counter, 64: slow_count)
#endif
}

// The rays k_scene_persistent set aside: one ray per lane, plain divisions, any scene.
template<typename IO, bool ANY = false, int MODE = MODE_CULLED>
__global__ void __launch_bounds__(256) k_scene_deferred(SceneRec sc, IO io, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count) {
	const uint32_t n = *count;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const uint32_t out = list[k];
		Ray3 r;
		io.load(out, r);
		Stack st;
		SceneHit h;
		traverse_scene<MODE, ANY>(sc, r, st, h);
		if (MODE == MODE_CULLED && !ANY && needs_reference_order(r, h.hit, h.distance)) traverse_scene<MODE_REFERENCE, false>(sc, r, st, h);
		io.store(out, h);
	}
}

// ---- host side
// The two-level kernel applies when the instance level is a tame tree; meshes it cannot handle
// exactly (lists, untamed trees) send their rays to k_scene_deferred one by one.
inline bool scene_persistent_ok(const PtxScene* s) { return !s->rec.mesh_only && !s->rec.top_is_list && s->rec.top_tame && s->rec.n_insts > 0; }

template<typename IO, bool ANY = false, int MODE = MODE_CULLED>
int launch_scene_persistent(PtxScene* s, const IO& io, uint64_t max_items) {
	unsigned long long* counters = nullptr;
	if (int rc = scene_counter_block(s, &counters)) return rc;
	if (int rc = scene_scratch(s, 7, size_t(max_items ? max_items : 1) * sizeof(uint32_t))) return rc;   // rays set aside for the plain kernel
	uint32_t* slow_list = static_cast<uint32_t*>(s->scratch[7]);
	if (int rc = scene_scratch(s, 8, size_t(max_items ? max_items : 1) * sizeof(uint32_t))) return rc;   // rays the second pass sets aside
	uint32_t* slow_count = reinterpret_cast<uint32_t*>(counters) + 16;   // same zeroed block, its own 64-byte line
	PTX_HIP(hipMemsetAsync(counters, 0, 1024 + 256, s->stream));   // both passes' counters (the second pass uses the block's second half)
	// Tuning knobs for experiments (not part of the ABI).
	static const int blocks_per_cu = getenv("PTX_SCENE_BLOCKS_PER_CU") ? atoi(getenv("PTX_SCENE_BLOCKS_PER_CU")) : 5;
	static const int refill_min = getenv("PTX_SCENE_REFILL_MIN_IDLE") ? atoi(getenv("PTX_SCENE_REFILL_MIN_IDLE")) : SCENE_REFILL_MIN_IDLE;
	static const unsigned long long wave_chunk = getenv("PTX_SCENE_WAVE_CHUNK") ? std::max(1, atoi(getenv("PTX_SCENE_WAVE_CHUNK"))) : SCENE_WAVE_CHUNK;
	static const int walk_steps = getenv("PTX_SCENE_WALK_STEPS") ? std::max(1, atoi(getenv("PTX_SCENE_WALK_STEPS"))) : SCENE_WALK_STEPS;
	const uint64_t want = (max_items + 255) / 256;
	const dim3 block(256), grid(uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(want, uint64_t(s->cu_count) * uint64_t(std::max(1, blocks_per_cu))))));
	hipLaunchKernelGGL((k_scene_persistent<IO, ANY, false, MODE>), grid, block, 0, s->stream, s->rec, io, counters, slow_list, slow_count, refill_min, wave_chunk, walk_steps);
#ifdef PTX_PROFILE_LANES
	{
		static unsigned long long tot[9] = {};
		static int launches = 0;
		unsigned long long h[32];
		PTX_HIP(hipStreamSynchronize(s->stream));
		PTX_HIP(hipMemcpy(h, counters, sizeof(h), hipMemcpyDeviceToHost));
		for (int k = 0; k < 9; k++) tot[k] += h[16 + k];
		static const int every = getenv("PTX_PROFILE_EVERY") ? std::max(1, atoi(getenv("PTX_PROFILE_EVERY"))) : 200;
		if (++launches % every == 0)
			std::fprintf(stderr, "[scene lanes any=%d] iterations %llu | walk %llu x %.1f | tri %llu x %.1f | exit %llu x %.1f | enter %llu x %.1f\n", int(ANY), tot[0], tot[1],
			             double(tot[2]) / double(tot[1] ? tot[1] : 1), tot[3], double(tot[4]) / double(tot[3] ? tot[3] : 1), tot[5], double(tot[6]) / double(tot[5] ? tot[5] : 1), tot[7],
			             double(tot[8]) / double(tot[7] ? tot[7] : 1));
	}
#endif
	// Second pass: the rays set aside walk the same kernel in its zero-aware form (most of them have a direction
	// component that is exactly zero, in world space or inside an instance); what that pass cannot take either
	// goes to the plain-division kernel.
	uint32_t* slow_list2 = static_cast<uint32_t*>(s->scratch[8]);
	unsigned long long* counters2 = counters + 128;   // the second half of the counter block
	uint32_t* slow_count2 = reinterpret_cast<uint32_t*>(counters2) + 16;
	hipLaunchKernelGGL((k_scene_persistent<ListIO<IO>, ANY, true, MODE>), grid, block, 0, s->stream, s->rec, ListIO<IO>{io, slow_list, slow_count}, counters2, slow_list2, slow_count2, refill_min, wave_chunk, walk_steps);
	hipLaunchKernelGGL((k_scene_deferred<IO, ANY, MODE>), dim3(uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(want, 256)))), block, 0, s->stream, s->rec, io, slow_list2, slow_count2);
	return PTX_OK;
}

}  // namespace ptx

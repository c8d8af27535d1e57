// persistent.hip.h -- ray queries against a single mesh with wave-level ray replenishment.
//
// Incoherent rays take very different numbers of traversal steps (on the 1M-triangle soup the
// slowest lane of a wave needs about 3.5x the wave's mean), so a wave that keeps one ray per
// lane until all 64 are finished runs mostly empty.  Here a wave is a pool of 64 traversal
// slots: whenever enough slots have run dry (a wave-wide __ballot), the wave takes that many
// new rays from a global counter with ONE atomic (lane ranks by prefix popcount of the
// ballot mask) and the dry lanes start over on the new rays while the others carry on.
// A wave claims WAVE_CHUNK consecutive slots of the (grouped) ray order at a time and refills
// from its own chunk, so the rays that share a wave at any moment are neighbours in that
// order -- the coherence the grouping pass created survives the replenishment.
// The arithmetic per ray is the same as in traverse_tree(); only the order in which rays are
// picked up changes, and no result depends on that.
#pragma once
#include "traverse.hip.h"

namespace ptx {

constexpr int REFILL_MIN_IDLE = 20;   // refill when at least this many of the 64 slots are dry
constexpr int TRI_MIN_LANES = 1;      // reference order: run the triangle step when at least this many lanes stand in a leaf
constexpr int PARK_MIN_LANES = 32;    // culled order: run the triangle phase when this many lanes have a leaf parked ...
constexpr int STUCK_MAX_LANES = 16;   // ... or this many lanes cannot go on walking until their parked leaf is tested
#ifndef PTX_WALK_STEPS
#define PTX_WALK_STEPS 1   // box steps per loop iteration of the culled order, before the triangle phase is considered
#endif
#ifndef PTX_WIDE_WAVES
#define PTX_WIDE_WAVES 5   // waves per SIMD the four-wide instance is compiled for (96 VGPRs)
#endif
#ifndef PTX_LDS_SLOTS
#define PTX_LDS_SLOTS 12
#endif
#ifndef PTX_TOP_NODES
#define PTX_TOP_NODES 128
#endif
#ifndef PTX_LDS_SLOTS_TOP
#define PTX_LDS_SLOTS_TOP 9   // stack entries in LDS for the instances that also keep the top of the tree there: 18 + 8 KB, six blocks per CU
#endif
constexpr int PERSISTENT_LDS_SLOTS = PTX_LDS_SLOTS;   // stack entries per lane kept in LDS (8 B each)
// The culled-order walk keeps the first TOP_NODES records of the node array -- the top of the tree, numbered
// breadth-first by the host's flattener -- in LDS: every ray passes through them, and a lane's 64-byte record comes
// out of LDS in four bank-parallel reads where the vector L1 spends four of its one-per-clock line look-ups.
constexpr int TOP_NODES = PTX_TOP_NODES;
constexpr unsigned RAY_PARTS = 8;           // parts of the ray order with their own claim counter (one per XCD)
constexpr unsigned PART_COUNTER0 = 32, PART_COUNTER_STRIDE = 8;   // in 8-byte words of the counter block: bytes 256, 320, ... (a 64-byte line each)
constexpr size_t COUNTER_BLOCK_BYTES = 1024;
constexpr unsigned long long WAVE_CHUNK = 64;     // consecutive ray slots a wave claims at a time (64: 3 % faster than 256, 20 % faster than 1024, measured)

// The culled-order instance needs 82 VGPRs left alone, two above the 80 that let six waves share a
// SIMD; held to 80 it runs 2 % faster on grouped rays (measured), the few spilled values being cold.
// WIDE: walk the four-wide form of the tree (SceneRec::nodes4; culled order only).
// ZERO: the second pass, over the rays the first one set aside (`order` is that list, *n_dev its length): rays with
// direction components that are exactly zero (orthographic views, straight-down shadow rays) are walked here with the
// zero-aware slab tests of traverse.hip.h -- a kernel of its own, so the first pass carries none of that code.
template<int MODE, bool ANY, bool WIDE = false, bool ZERO = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WIDE ? PTX_WIDE_WAVES : (ZERO ? 5 : 6), WIDE ? PTX_WIDE_WAVES : (ZERO ? 5 : 6)))) k_mesh_persistent(SceneRec sc, const PtxRay* __restrict__ rays, uint64_t n_rays, PtxHit* __restrict__ hits,
                                                         float* __restrict__ bary, uint32_t* __restrict__ inst, uint8_t* __restrict__ occluded,
                                                         unsigned long long* __restrict__ next_ray, const int refill_min_idle,
                                                         const uint32_t* __restrict__ order, uint32_t* __restrict__ slow_list, uint32_t* __restrict__ slow_count,
                                                         const int tri_min_lanes, const int stuck_max_lanes, const unsigned long long wave_chunk, const uint32_t* __restrict__ n_dev) {
	static_assert(!(WIDE && ZERO), "the four-wide step has no zero-aware form");
	const uint64_t n = ZERO ? uint64_t(*n_dev) : n_rays;
	if (ZERO && n == 0) return;   // nothing was set aside: the usual case
	const MeshRec mesh = sc.meshes[0];
	const TriGeom* __restrict__ tris = sc.tris + mesh.tri_base;
	const Node2* __restrict__ nodes = sc.nodes + mesh.node_base;
	const Node4* __restrict__ nodes4 = WIDE ? sc.nodes4 + mesh.node4_base : nullptr;
	static_assert(!WIDE || MODE == MODE_CULLED, "the four-wide tree serves the culled order only");
	const unsigned lane = threadIdx.x & 63u;
	const unsigned long long lanes_below = (1ull << lane) - 1ull;

	constexpr int TOP = (MODE == MODE_CULLED && !WIDE && !ZERO) ? TOP_NODES : 0;
	constexpr int LDS_SLOTS = TOP > 0 ? PTX_LDS_SLOTS_TOP : PERSISTENT_LDS_SLOTS;
	__shared__ uint32_t lds_ref[LDS_SLOTS * 256];
	__shared__ float lds_tnear[LDS_SLOTS * 256];
	__shared__ float4 lds_top[TOP > 0 ? TOP * 4 : 1];
	if (TOP > 0) {
		// the node array is padded by TOP_NODES records on the host, so this read stays inside it for any tree
		for (int i = threadIdx.x; i < TOP * 4; i += 256) lds_top[i] = reinterpret_cast<const float4*>(nodes)[i];
		__syncthreads();
	}
	Stack spill;
	LdsStack<LDS_SLOTS, 256> st{lds_ref + threadIdx.x, lds_tnear + threadIdx.x, &spill};
	// Parking leaves reorders triangle tests: fine for the culled mode (its tie rule is order free),
	// not for the reference order.
	constexpr bool POSTPONE = MODE == MODE_CULLED;
	int sp = 0;
	uint32_t pend = 0u;               // a leaf reference parked for the next triangle phase (0: none)
	uint32_t ref = REF_DONE;          // REF_DONE + have_ray == false: the slot is dry
	uint32_t tri_k = 0;               // next triangle of the leaf `ref` stands in
	bool have_ray = false;
	uint64_t ray_index = 0;
	Ray3 r{};
	RayInv inv{};
	MeshHit cur{};
	bool pool_empty = false;
	unsigned long long chunk_next = 0, chunk_end = 0;   // this wave's claimed slots [chunk_next, chunk_end), wave-uniform
	// The ray order is cut into RAY_PARTS contiguous parts, each with a claim counter of its own.  Blocks
	// b and b + 8 run on the same XCD, so with part = b % 8 the rays in flight on one XCD are neighbours
	// in the grouped order rather than a sample of everything in flight on the chip, and the nodes they
	// touch fit that XCD's L2 better.
	const unsigned long long part_size = ((n + RAY_PARTS - 1) / RAY_PARTS + 63ull) & ~63ull;
	unsigned part = blockIdx.x % RAY_PARTS, parts_dry = 0;

	auto store_result = [&]() {
		if (ANY) {
			occluded[ray_index] = cur.hit ? 1 : 0;
		} else {
			PtxHit out;
			out.distance = cur.distance;
			out.prim = cur.hit ? cur.prim : PTX_NO_HIT;
			hits[ray_index] = out;
			if (bary) reinterpret_cast<float2*>(bary)[ray_index] = cur.hit ? make_float2(cur.u, cur.v) : make_float2(0.0f, 0.0f);
			if (inst) inst[ray_index] = cur.hit ? 0u : PTX_NO_HIT;
		}
		have_ray = false;
	};

#ifdef PTX_PROFILE_LANES
	unsigned long long prof_iters = 0, prof_node_lanes = 0, prof_tri_lanes = 0, prof_node_iters = 0, prof_tri_iters = 0, prof_busy = 0;
	// 100 MHz clock: first wave start, first wave to find the pool dry, last wave out (minima kept as maxima of the complement)
	if (lane == 0) atomicMax(next_ray + 22, ~(unsigned long long)wall_clock64());
#endif
	for (;;) {
		// ---- replenish dry slots
		const unsigned long long dry = __ballot(!have_ray);
		const int n_dry = __popcll(dry);
		if (n_dry == 64 && pool_empty) break;
		if (!pool_empty && n_dry >= refill_min_idle) {
			if (chunk_next >= chunk_end) {
				// Claim the next chunk of this wave's part of the ray order (one atomic per chunk); when the
				// part is dry, move on to the next one and help there, until all of them are dry.
				for (;;) {
					const unsigned long long lo = part_size * part < n ? part_size * part : n;
					const unsigned long long hi = part_size * (part + 1) < n ? part_size * (part + 1) : n;
					unsigned long long base = 0;
					if (lane == 0) base = atomicAdd(next_ray + PART_COUNTER0 + PART_COUNTER_STRIDE * part, wave_chunk);
					base = lo + __shfl(base, 0);
					if (base < hi) {
						chunk_next = base, chunk_end = base + wave_chunk < hi ? base + wave_chunk : hi;
						break;
					}
					if (++parts_dry == RAY_PARTS) {
						pool_empty = true, chunk_next = chunk_end = 0;
#ifdef PTX_PROFILE_LANES
						if (lane == 0) atomicMax(next_ray + 23, ~(unsigned long long)wall_clock64());
#endif
						break;
					}
					part = (part + 1) % RAY_PARTS;
				}
			}
			const unsigned long long base = chunk_next;
			const unsigned long long take = (unsigned long long)n_dry < chunk_end - chunk_next ? (unsigned long long)n_dry : chunk_end - chunk_next;
			chunk_next += take;
			if (!have_ray) {
				const uint64_t slot = base + __popcll(dry & lanes_below);
				if (slot < base + take) {
					const uint64_t idx = order ? uint64_t(order[slot]) : slot;   // rays grouped for coherence (raysort.hip.h)
					ray_index = idx;
					have_ray = true;
					const float4* rp = reinterpret_cast<const float4*>(rays + idx);
					const float4 a = rp[0], b = rp[1];
					r = Ray3{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
					if (ZERO ? ray_in_fast_range_z(r) : ray_in_fast_range(r)) {
						// BVH::hit seeds the running Trace with hit = false, distance = inf (bvh.cpp:257-259)
						cur.hit = false, cur.distance = INFINITY, cur.prim = 0xFFFFFFFFu;
						if (ZERO) inv.yx = slab_rcp(r.dx), inv.yy = slab_rcp(r.dy), inv.yz = slab_rcp(r.dz);
						else inv.yx = refined_rcp(r.dx), inv.yy = refined_rcp(r.dy), inv.yz = refined_rcp(r.dz);
						float tn;
						sp = 0;
						tri_k = 0;
						if (slab_test<true, ZERO>(mesh.root_min[0], mesh.root_min[1], mesh.root_min[2], mesh.root_max[0], mesh.root_max[1], mesh.root_max[2], r, inv, tn))
							ref = WIDE ? mesh.root4_ref : mesh.root_ref;
						else
							ref = REF_DONE;
					} else {
						// Outside the range where the short division form is proven exact (a zero direction
						// component, a huge or tiny coordinate): rare.  Hand the ray to k_mesh_deferred, which
						// runs the plain-division traversal, so that code stays out of this kernel.
						slow_list[atomicAdd(slow_count, 1u)] = uint32_t(idx);
						have_ray = false;
						ref = REF_DONE;
					}
				}
			}
		}
		if (POSTPONE) {
			// ---- box-test step for every lane standing on an interior node.  A lane that reaches a
			// leaf does not stop for it: it parks the leaf in `pend` and goes on with its next pending
			// subtree, so the triangle code below runs rarely and with many lanes instead of every
			// iteration with a few.  A parked leaf is tested later than it would have been, so the
			// cull limit the lane uses meanwhile is looser (never tighter) than it could be: the
			// outcome is unchanged because the culled mode's result does not depend on test order.
#pragma unroll
			for (int step = 0; step < PTX_WALK_STEPS; step++) {
				const unsigned long long on_node = __ballot(!(ref & REF_LEAF));
#ifdef PTX_PROFILE_LANES
				prof_iters++, prof_node_lanes += __popcll(on_node), prof_node_iters += on_node != 0ull, prof_busy += __popcll(__ballot(have_ray));
#endif
				if (on_node != 0ull) {
					if (!(ref & REF_LEAF)) {
						if (WIDE) {
							node4_decide<ANY>(nodes4 + (ref & REF_INDEX_MASK), r, inv, st, sp, ref, cur.hit, cur.distance);
						} else {
							const uint32_t index = ref & REF_INDEX_MASK;
							float4 n0, n1, n2;
							uint2 kids;
							if (TOP > 0 && index < uint32_t(TOP)) fetch_node_lds(lds_top, index, n0, n1, n2, kids);
							else fetch_node(nodes + index, n0, n1, n2, kids);
							node_decide<MODE, ANY, true, true, ZERO>(n0, n1, n2, kids, r, inv, st, 0, sp, ref, cur.hit, cur.distance);
						}
					}
				}
				// one pop loop for both reasons to pop: no child taken, or a leaf reached and parked
				if ((ref & REF_LEAF) && ref != REF_DONE && (ref == REF_NONE || pend == 0u)) {
					if (ref != REF_NONE && (ref & 15u)) pend = ref;   // an empty leaf has nothing to test
					pop_ref<MODE, ANY>(r, st, 0, sp, ref, cur.hit, cur.distance);
				}
			}
			// ---- triangle phase: when enough leaves are parked, or too many lanes cannot go on
			// (a second leaf reached, or the walk finished, while one is still parked)
			const unsigned long long parked = __ballot(pend != 0u);
			const unsigned long long stuck = __ballot(pend != 0u && (ref & REF_LEAF));
			const unsigned long long walking = __ballot(!(ref & REF_LEAF));
			if (parked != 0ull && (__popcll(parked) >= tri_min_lanes || __popcll(stuck) >= stuck_max_lanes || walking == 0ull)) {
#ifdef PTX_PROFILE_LANES
				prof_tri_lanes += __popcll(parked), prof_tri_iters++;
#endif
				if (pend != 0u) {
					const bool stop = mesh_leaf<MODE, ANY, false>(tris, (pend & REF_INDEX_MASK) >> 4, pend & 15u, r, cur, nullptr);
					pend = 0u;
					if (stop) ref = REF_DONE, sp = 0;
					// a lane that was held up by a second leaf parks it at the next iteration's pop site
				}
			}
			if (ref == REF_DONE && pend == 0u && have_ray) store_result();
			continue;
		}
		// ---- one step for every busy slot: a box-test step for the lanes standing on an interior
		// node, then one triangle for the lanes standing in a leaf.  The load/return path of the
		// CU's vector memory pipeline is paid per wave instruction, however few lanes are active,
		// so the step is shaped to advance as many lanes as possible per memory instruction.
		const unsigned long long on_node = __ballot(!(ref & REF_LEAF));
#ifdef PTX_PROFILE_LANES
		prof_iters++, prof_node_lanes += __popcll(on_node), prof_node_iters += on_node != 0ull, prof_busy += __popcll(__ballot(have_ray));
		{
			const unsigned long long lf = __ballot(ref != REF_DONE && (ref & REF_LEAF));
			prof_tri_lanes += __popcll(lf), prof_tri_iters += lf != 0ull;
		}
#endif
		if (on_node != 0ull) {
			if (!(ref & REF_LEAF)) {
				node_step<MODE, ANY, true, false, ZERO, true>(nodes, r, inv, st, 0, sp, ref, cur.hit, cur.distance, nullptr);
				tri_k = 0;
			}
		}
		// The triangle step waits until enough lanes want it (or nobody has box work left): it then
		// runs less often and fuller.
		const int in_leaf = __popcll(__ballot(ref != REF_DONE && (ref & REF_LEAF)));
		if (in_leaf >= tri_min_lanes || (in_leaf > 0 && on_node == 0ull)) {
			if (ref != REF_DONE && (ref & REF_LEAF)) {
				const uint32_t first = (ref & REF_INDEX_MASK) >> 4, count = ref & 15u;
				bool stop = false;
				if (tri_k < count) {
					stop = mesh_leaf<MODE, ANY, false>(tris, first + tri_k, 1, r, cur, nullptr);
					tri_k++;
				}
				if (stop) ref = REF_DONE;
				else if (tri_k >= count) {
					pop_ref<MODE, ANY>(r, st, 0, sp, ref, cur.hit, cur.distance);
					tri_k = 0;
				}
			}
		}
		if (ref == REF_DONE && have_ray) store_result();
	}
#ifdef PTX_PROFILE_LANES
	if (lane == 0) {
		atomicAdd(next_ray + 16, prof_iters), atomicAdd(next_ray + 17, prof_node_lanes), atomicAdd(next_ray + 18, prof_tri_lanes);
		atomicAdd(next_ray + 19, prof_node_iters), atomicAdd(next_ray + 20, prof_tri_iters), atomicAdd(next_ray + 21, prof_busy);
		const unsigned long long t_out = wall_clock64();
		atomicMax(next_ray + 24, t_out), atomicAdd(next_ray + 25, t_out & 0xFFFFFFFFFFull), atomicAdd(next_ray + 26, 1ull);
	}
#endif
}

// The rays k_mesh_persistent set aside: one ray per lane, plain `/` divisions.
template<int MODE, bool ANY>
__global__ void __launch_bounds__(256) k_mesh_deferred(SceneRec sc, const PtxRay* __restrict__ rays, const uint32_t* __restrict__ list,
                                                       const uint32_t* __restrict__ count, PtxHit* __restrict__ hits, float* __restrict__ bary,
                                                       uint32_t* __restrict__ inst, uint8_t* __restrict__ occluded) {
	const uint32_t n = *count;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const uint32_t i = list[k];
		const float4* rp = reinterpret_cast<const float4*>(rays + i);
		const float4 a = rp[0], b = rp[1];
		Ray3 r{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
		Stack st;
		MeshHit cur;
		traverse_mesh<MODE, ANY, false>(sc, sc.meshes[0], r, st, 0, cur, nullptr);
		if (MODE == MODE_CULLED && !ANY && needs_reference_order(r, cur.hit, cur.distance)) traverse_mesh<MODE_REFERENCE, ANY, false>(sc, sc.meshes[0], r, st, 0, cur, nullptr);
		if (ANY) {
			occluded[i] = cur.hit ? 1 : 0;
		} else {
			PtxHit out;
			out.distance = cur.distance;
			out.prim = cur.hit ? cur.prim : PTX_NO_HIT;
			hits[i] = out;
			if (bary) reinterpret_cast<float2*>(bary)[i] = cur.hit ? make_float2(cur.u, cur.v) : make_float2(0.0f, 0.0f);
			if (inst) inst[i] = cur.hit ? 0u : PTX_NO_HIT;
		}
	}
}

}  // namespace ptx

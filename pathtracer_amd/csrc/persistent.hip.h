// persistent.hip.h -- ray queries against a single mesh with wave-level ray replenishment.
//
// Incoherent rays take very different numbers of traversal steps (on the 1M-triangle soup the
// slowest lane of a wave needs about 3.5x the wave's mean), so a wave that keeps one ray per
// lane until all 64 are finished runs mostly empty.  Here a wave is a pool of 64 traversal
// slots: whenever enough slots have run dry (a wave-wide __ballot), the wave takes that many
// new rays from a global counter with ONE atomic (lane ranks by prefix popcount of the
// ballot mask) and the dry lanes start over on the new rays while the others carry on.
// The arithmetic per ray is the same as in traverse_tree(); only the order in which rays are
// picked up changes, and no result depends on that.
#pragma once
#include "traverse.hip.h"

namespace ptx {

constexpr int REFILL_MIN_IDLE = 20;   // refill when at least this many of the 64 slots are dry

template<int MODE, bool ANY>
__global__ void __launch_bounds__(256) k_mesh_persistent(SceneRec sc, const PtxRay* __restrict__ rays, uint64_t n, PtxHit* __restrict__ hits,
                                                         float* __restrict__ bary, uint32_t* __restrict__ inst, uint8_t* __restrict__ occluded,
                                                         unsigned long long* __restrict__ next_ray) {
	const MeshRec mesh = sc.meshes[0];
	const TriGeom* __restrict__ tris = sc.tris + mesh.tri_base;
	const Node2* __restrict__ nodes = sc.nodes + mesh.node_base;
	const unsigned lane = threadIdx.x & 63u;
	const unsigned long long lanes_below = (1ull << lane) - 1ull;

	Stack st;
	int sp = 0;
	uint32_t ref = REF_DONE;          // REF_DONE + have_ray == false: the slot is dry
	bool have_ray = false;
	uint64_t ray_index = 0;
	Ray3 r{};
	RayInv inv{};
	MeshHit cur{};
	bool pool_empty = false;

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

	for (;;) {
		// ---- replenish dry slots
		const unsigned long long dry = __ballot(!have_ray);
		const int n_dry = __popcll(dry);
		if (n_dry == 64 && pool_empty) break;
		if (!pool_empty && n_dry >= REFILL_MIN_IDLE) {
			unsigned long long base = 0;
			const unsigned leader = __ffsll((long long)dry) - 1;
			if (lane == leader) base = atomicAdd(next_ray, (unsigned long long)n_dry);
			base = __shfl(base, leader);
			if (base + n_dry >= n) pool_empty = true;
			if (!have_ray) {
				const uint64_t idx = base + __popcll(dry & lanes_below);
				if (idx < n) {
					ray_index = idx;
					have_ray = true;
					const float4* rp = reinterpret_cast<const float4*>(rays + idx);
					const float4 a = rp[0], b = rp[1];
					r = Ray3{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
					if (ray_in_fast_range(r)) {
						// BVH::hit seeds the running Trace with hit = false, distance = inf (bvh.cpp:257-259)
						cur.hit = false, cur.distance = INFINITY, cur.prim = 0xFFFFFFFFu;
						inv.yx = refined_rcp(r.dx), inv.yy = refined_rcp(r.dy), inv.yz = refined_rcp(r.dz);
						float tn;
						sp = 0;
						if (slab_test<true>(mesh.root_min[0], mesh.root_min[1], mesh.root_min[2], mesh.root_max[0], mesh.root_max[1], mesh.root_max[2], r, inv, tn))
							ref = mesh.root_ref;
						else
							ref = REF_DONE;
					} else {
						// Outside the range where the short division form is proven exact (a zero direction
						// component, a huge or tiny coordinate): rare, so run the whole query now with `/`.
						traverse_mesh<MODE, ANY, false>(sc, mesh, r, st, 0, cur, nullptr);
						ref = REF_DONE;
					}
				}
			}
		}
		// ---- traverse until too many slots are dry again (or, once the pool is empty, until all are)
		for (;;) {
			while (!(ref & REF_LEAF)) node_step<MODE, ANY, true, false>(nodes, r, inv, st, 0, sp, ref, cur.hit, cur.distance, nullptr);
			if (ref != REF_DONE) {
				if (mesh_leaf<MODE, ANY, false>(tris, (ref & REF_INDEX_MASK) >> 4, ref & 15u, r, cur, nullptr)) ref = REF_DONE;
				else pop_ref<MODE, ANY>(r, st, 0, sp, ref, cur.hit, cur.distance);
			}
			if (ref == REF_DONE && have_ray) store_result();
			const int busy = __popcll(__ballot(have_ray));
			if (busy == 0 || (!pool_empty && 64 - busy >= REFILL_MIN_IDLE)) break;
		}
	}
}

}  // namespace ptx

// bvh_build.cpp -- host-side binned-SAH builder.
//
// Produces, node for node and primitive for primitive, the tree PT::BVH<P>::build produces
// (reference: src/pathtracer/bvh.cpp:31-243), in O(n) work per tree level instead of the
// reference's per-candidate vector copies, and without its out-of-range bin write:
//   * 8 equal bins per axis over the NODE's box; a primitive goes to the bin of its box
//     centre (bvh.cpp:104-115).  The bin index is clamped to [0,7]; the reference casts an
//     unclamped float to uint32_t and indexes with it (bvh.cpp:111-112).
//   * 7 candidate planes per axis, cost SA(L)*nL + SA(R)*nR (bvh.cpp:139), strictly lowest
//     cost with both sides non-empty wins, first found wins ties (x, y, z; plane 1..7).
//   * The range is then partitioned with the predicate `centre < min + len*plane`
//     (bvh.cpp:191-199) using the exchange order of libstdc++'s std::partition, but the
//     children's sizes and boxes come from the bins (bvh.cpp:213-221).  Where the predicate
//     and the bins disagree the child range holds a primitive its box was not built from;
//     that is what the reference does, so it is kept (and counted).
//   * Children are appended as consecutive nodes, left first, then each is refined
//     depth-first if it holds more than max_leaf_size primitives (bvh.cpp:223-241).
#include "bvh_build.h"

#include <algorithm>
#include <cfloat>
#include <cmath>

namespace ptx {

namespace {

struct Box {
	float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
	void grow(const float* b) {
		for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], b[k]), hi[k] = std::max(hi[k], b[3 + k]);
	}
	void grow(const Box& b) {
		for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], b.lo[k]), hi[k] = std::max(hi[k], b.hi[k]);
	}
	float area() const {  // BBox::surface_area, lib/bbox.h:55-59
		if (lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]) return 0.0f;
		float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
		return 2.0f * (ex * ez + ex * ey + ey * ez);
	}
};

constexpr int NBINS = 8;

void store_box(PtxBvhNode& n, const Box& b) {
	for (int k = 0; k < 3; k++) n.bmin[k] = b.lo[k], n.bmax[k] = b.hi[k];
}

}  // namespace

void triangle_boxes(const float* verts, const uint32_t* indices, uint32_t n_tris, std::vector<float>& boxes) {
	boxes.resize(size_t(n_tris) * 6);
	const float eps = 0.0001f;
	for (uint32_t t = 0; t < n_tris; t++) {
		const float* a = verts + size_t(indices[t * 3 + 0]) * 8;
		const float* b = verts + size_t(indices[t * 3 + 1]) * 8;
		const float* c = verts + size_t(indices[t * 3 + 2]) * 8;
		float* out = &boxes[size_t(t) * 6];
		for (int k = 0; k < 3; k++) {
			float lo = std::min(a[k], std::min(b[k], c[k])), hi = std::max(a[k], std::max(b[k], c[k]));
			if (lo == hi) lo -= eps, hi += eps;   // flat boxes are padded (tri_mesh.cpp:28-39)
			out[k] = lo, out[3 + k] = hi;
		}
	}
}

void build_bvh(const float* boxes, uint32_t n, uint32_t max_leaf_size, HostBvh& out) {
	out = HostBvh();
	out.order.resize(n);
	for (uint32_t i = 0; i < n; i++) out.order[i] = i;

	// Box centres, (min + max) * 0.5f per axis (BBox::center, lib/bbox.h:45-47).
	std::vector<float> centre(size_t(n) * 3);
	Box scene;
	for (uint32_t i = 0; i < n; i++) {
		const float* b = boxes + size_t(i) * 6;
		for (int k = 0; k < 3; k++) centre[size_t(i) * 3 + k] = (b[k] + b[3 + k]) * 0.5f;
		scene.grow(b);
	}

	PtxBvhNode root{};
	store_box(root, scene);
	root.start = 0, root.size = n, root.l = root.r = 0;
	out.nodes.push_back(root);
	if (n <= max_leaf_size) return;

	struct Todo {
		uint32_t node, depth;
	};
	std::vector<Todo> todo{{0u, 0u}};
	uint32_t* order = out.order.data();

	while (!todo.empty()) {
		const Todo job = todo.back();
		todo.pop_back();
		const PtxBvhNode parent = out.nodes[job.node];
		const uint32_t begin = parent.start, count = parent.size;

		float best_cost = INFINITY;
		int best_axis = 0, best_plane = 0;
		Box best_l, best_r;
		uint32_t best_nl = 0, best_nr = 0;
		uint64_t clamped_here = 0;

		for (int axis = 0; axis < 3; axis++) {
			const float span_lo = parent.bmin[axis], span_hi = parent.bmax[axis];
			const float bin_len = (span_hi - span_lo) / NBINS;
			Box bin_box[NBINS];
			uint32_t bin_n[NBINS] = {};
			for (uint32_t i = begin; i < begin + count; i++) {
				const uint32_t p = order[i];
				const float f = std::floor((centre[size_t(p) * 3 + axis] - span_lo) / bin_len);
				int b;
				if (!(f >= 0.0f)) b = 0, clamped_here++;
				else if (f > float(NBINS - 1)) b = NBINS - 1, clamped_here++;
				else b = int(f);
				bin_n[b]++;
				bin_box[b].grow(boxes + size_t(p) * 6);
			}
			// Suffix unions once, prefix union on the fly: min/max are exact, so any order of
			// union gives the boxes the reference gets by re-enclosing bins per candidate.
			Box suffix[NBINS + 1];
			uint32_t suffix_n[NBINS + 1] = {};
			for (int b = NBINS - 1; b >= 0; b--) {
				suffix[b] = suffix[b + 1];
				suffix[b].grow(bin_box[b]);
				suffix_n[b] = suffix_n[b + 1] + bin_n[b];
			}
			Box prefix;
			uint32_t prefix_n = 0;
			for (int plane = 1; plane < NBINS; plane++) {
				prefix.grow(bin_box[plane - 1]);
				prefix_n += bin_n[plane - 1];
				const uint32_t nl = prefix_n, nr = suffix_n[plane];
				const float cost = prefix.area() * float(nl) + suffix[plane].area() * float(nr);
				if (cost < best_cost && nl > 0 && nr > 0) {
					best_cost = cost, best_axis = axis, best_plane = plane;
					best_l = prefix, best_r = suffix[plane], best_nl = nl, best_nr = nr;
				}
			}
		}
		out.clamped += clamped_here;

		// Partition [begin, begin+count) by the reference's predicate, exchanging elements in
		// the order libstdc++'s bidirectional std::partition does (first misplaced element
		// from the front with the last misplaced element from the back).
		const float span_lo = parent.bmin[best_axis], span_hi = parent.bmax[best_axis];
		const float bin_len = (span_hi - span_lo) / NBINS;
		const float threshold = span_lo + bin_len * float(uint32_t(best_plane));
		auto goes_left = [&](uint32_t p) { return centre[size_t(p) * 3 + best_axis] < threshold; };
		{
			uint32_t front = begin, back = begin + count;
			for (;;) {
				while (front != back && goes_left(order[front])) ++front;
				if (front == back) break;
				--back;
				while (front != back && !goes_left(order[back])) --back;
				if (front == back) break;
				std::swap(order[front], order[back]);
				++front;
			}
			if (best_nl + best_nr == 0) out.dropped += count;
			else if (front - begin != best_nl) out.predicate_disagreements += (front - begin > best_nl) ? (front - begin - best_nl) : (best_nl - (front - begin));
		}

		PtxBvhNode left{}, right{};
		store_box(left, best_l);
		left.start = begin, left.size = best_nl;
		store_box(right, best_r);
		right.start = begin + best_nl, right.size = best_nr;
		const uint32_t li = uint32_t(out.nodes.size()), ri = li + 1;
		left.l = left.r = left.start;
		right.l = right.r = right.start;
		out.nodes.push_back(left);
		out.nodes.push_back(right);
		out.nodes[job.node].l = li, out.nodes[job.node].r = ri;
		out.max_depth = std::max(out.max_depth, job.depth + 1);
		// Depth-first, left subtree first: push right, then left.
		if (best_nr > max_leaf_size) todo.push_back({ri, job.depth + 1});
		if (best_nl > max_leaf_size) todo.push_back({li, job.depth + 1});
	}
}

bool validate_tree(const PtxBvhNode* nodes, uint32_t n_nodes, uint32_t n_prims, uint32_t& max_depth) {
	max_depth = 0;
	if (n_nodes == 0) return false;
	struct Item {
		uint32_t node, depth;
	};
	std::vector<Item> stack{{0u, 0u}};
	uint64_t visited = 0;
	while (!stack.empty()) {
		Item it = stack.back();
		stack.pop_back();
		if (++visited > n_nodes) return false;  // more visits than nodes: not a tree
		const PtxBvhNode& n = nodes[it.node];
		if (uint64_t(n.start) + n.size > n_prims) return false;
		max_depth = std::max(max_depth, it.depth);
		if (n.l == n.r) continue;
		if (n.l >= n_nodes || n.r >= n_nodes || n.l == it.node || n.r == it.node) return false;
		stack.push_back({n.r, it.depth + 1});
		stack.push_back({n.l, it.depth + 1});
	}
	return true;
}

}  // namespace ptx

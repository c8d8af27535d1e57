// ptx_api.hip -- implementation of the C ABI declared in include/ptx.h: scene construction
// (host BVH build, flattening, upload) and the ray-query entry points.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "host_math.h"
#include "scene.h"
#include "persistent.hip.h"
#include "scene_persistent.hip.h"
#include "raysort.hip.h"
#include "traverse.hip.h"

namespace ptx {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
	g_last_error = msg;
	return code;
}
int hip_fail(hipError_t e, const char* what) {
	g_last_error = std::string(what) + ": " + hipGetErrorString(e);
	return e == hipErrorOutOfMemory ? PTX_ERR_OOM : PTX_ERR_DEVICE;
}

// ------------------------------------------------------------------ kernels
template<int MODE, bool MESH_ONLY>
__global__ void __launch_bounds__(256) k_trace_closest(SceneRec sc, const PtxRay* __restrict__ rays, uint64_t n, PtxHit* __restrict__ hits,
                                                       float* __restrict__ bary, uint32_t* __restrict__ inst) {
	const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float4* rp = reinterpret_cast<const float4*>(rays + i);
	const float4 a = rp[0], b = rp[1];
	Ray3 r{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
	Stack st;
	SceneHit h;
	traverse_scene<MODE, false, false, MESH_ONLY>(sc, r, st, h);
	if (MODE == MODE_CULLED && needs_reference_order(r, h.hit, h.distance)) traverse_scene<MODE_REFERENCE, false, false, MESH_ONLY>(sc, r, st, h);
	PtxHit out;
	out.distance = h.distance;
	out.prim = h.hit ? h.prim : PTX_NO_HIT;
	hits[i] = out;
	if (bary) reinterpret_cast<float2*>(bary)[i] = h.hit ? make_float2(h.u, h.v) : make_float2(0.0f, 0.0f);
	if (inst) inst[i] = h.hit ? h.inst : PTX_NO_HIT;
}

template<int MODE, bool MESH_ONLY>
__global__ void __launch_bounds__(256) k_trace_any(SceneRec sc, const PtxRay* __restrict__ rays, uint64_t n, uint8_t* __restrict__ occluded) {
	const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float4* rp = reinterpret_cast<const float4*>(rays + i);
	const float4 a = rp[0], b = rp[1];
	Ray3 r{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
	Stack st;
	SceneHit h;
	traverse_scene<MODE, true, false, MESH_ONLY>(sc, r, st, h);
	occluded[i] = h.hit ? 1 : 0;
}

// Diagnostic twin of k_trace_closest: per ray, the interior nodes fetched, leaves entered and
// triangles tested.
template<int MODE>
__global__ void __launch_bounds__(256) k_trace_stats(SceneRec sc, const PtxRay* __restrict__ rays, uint64_t n, uint32_t* __restrict__ stats3) {
	const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float4* rp = reinterpret_cast<const float4*>(rays + i);
	const float4 a = rp[0], b = rp[1];
	Ray3 r{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
	Stack st;
	SceneHit h;
	TraverseStats ts{0, 0, 0, 0};
	traverse_scene<MODE, false, true>(sc, r, st, h, &ts);
	stats3[i * 3 + 0] = ts.nodes | (ts.max_sp << 24), stats3[i * 3 + 1] = ts.leaves, stats3[i * 3 + 2] = ts.tris;
}

// Self test of the short division form: counts operand pairs for which div_by() differs in
// any bit from the IEEE quotient.  Operands are drawn by a counter-based generator over the
// range the traversal's static checks admit (numerator 0 or 2^-94..2^40, divisor 2^-40..2^10).
__global__ void __launch_bounds__(256) k_selftest_division(uint64_t pairs_per_thread, uint64_t seed, unsigned long long* __restrict__ mismatches) {
	uint64_t x = seed ^ (0x9E3779B97F4A7C15ull * (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x + 1));
	auto next = [&]() {
		x ^= x << 13, x ^= x >> 7, x ^= x << 17;
		return x;
	};
	unsigned long long bad = 0;
	for (uint64_t k = 0; k < pairs_per_thread; k++) {
		const uint64_t u = next(), w = next();
		// divisor: sign, exponent in [-40, 10], random mantissa (every 16th: all-ones or all-zero mantissa)
		uint32_t mb = uint32_t(u) & 0x7FFFFFu;
		if ((u >> 60) == 0) mb = (u >> 59) & 1 ? 0x7FFFFFu : 0u;
		const uint32_t eb = 127 - 40 + uint32_t((u >> 23) % 51);
		const float b = __uint_as_float((uint32_t(u >> 32) & 0x80000000u) | (eb << 23) | mb);
		// numerator: sign, exponent in [-94, 39], random mantissa; every 64th is an exact zero
		uint32_t ma = uint32_t(w) & 0x7FFFFFu;
		if ((w >> 60) == 1) ma = (w >> 59) & 1 ? 0x7FFFFFu : 0u;
		const uint32_t ea = 127 - 94 + uint32_t((w >> 23) % 134);
		float a = __uint_as_float((uint32_t(w >> 32) & 0x80000000u) | (ea << 23) | ma);
		if ((w >> 58) == 3) a = 0.0f;
		// every 8th pair: numerator near a multiple of the divisor (quotients close to rounding boundaries)
		if (((u >> 55) & 7) == 0) a = b * float(int((w >> 40) & 0xFFFF) - 32768) * (1.0f + float(int(w & 7) - 3) * 5.9604645e-8f);
		if (!(a == 0.0f || exp_within(a, -94, 39))) continue;
		const float want = a / b;
		const float got = div_by(a, b, refined_rcp(b));
		// the traversal only compares quotients, so +0 and -0 are the same answer
		if (__float_as_uint(want) != __float_as_uint(got) && !(want == 0.0f && got == 0.0f)) bad++;
	}
	if (bad) atomicAdd(mismatches, bad);
}

// ------------------------------------------------------------------ flattening
// Arrays of hundreds of megabytes whose every element is written by a parallel pass right after allocation: a
// vector that value-initialises them first would spend as long again clearing them on one thread.
template<typename T>
struct NoInitAlloc : std::allocator<T> {
	template<typename U>
	struct rebind {
		using other = NoInitAlloc<U>;
	};
	template<typename U, typename... Args>
	void construct(U* p, Args&&... args) {
		if constexpr (sizeof...(Args) == 0) ::new (static_cast<void*>(p)) U;
		else ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
	}
};
template<typename T>
using BigVector = std::vector<T, NoInitAlloc<T>>;

struct FlatTree {
	BigVector<Node2> nodes;
	uint32_t root_ref = make_leaf_ref(0, 0);
	uint32_t depth = 0;
	uint32_t n_nocull = 0;   // child references marked REF_NOCULL
	bool tame = true;        // every stored box coordinate is 0 or in [2^-70, 2^39)
};

// Host twin of tame_coord() in traverse.hip.h.
static bool tame_coord_host(float c) {
	if (c == 0.0f) return true;
	uint32_t bits;
	std::memcpy(&bits, &c, 4);
	const uint32_t e = (bits >> 23) & 0xFFu;
	return e >= uint32_t(127 - 70) && e <= uint32_t(127 + 38);
}

// Runs fn(task) for task = 0 .. n_tasks-1 on the builder's worker threads, tasks handed out one at a time (they are
// subtrees of very different sizes).
static void parallel_tasks(size_t n_tasks, uint64_t work, const std::function<void(size_t)>& fn) {
	const unsigned threads = work >= (1u << 17) ? std::min<unsigned>(build_threads(), unsigned(std::max<size_t>(1, n_tasks))) : 1u;
	if (threads <= 1) {
		for (size_t t = 0; t < n_tasks; t++) fn(t);
		return;
	}
	std::atomic<size_t> next{0};
	auto run = [&] {
		for (size_t t = next.fetch_add(1); t < n_tasks; t = next.fetch_add(1)) fn(t);
	};
	std::vector<std::thread> pool;
	for (unsigned k = 1; k < threads; k++) pool.emplace_back(run);
	run();
	for (auto& t : pool) t.join();
}

// Canonical node array -> interior-only Node2 array (children's boxes pulled up into the
// parent) numbered in depth-first order, leaves folded into child references.
// prim_boxes (6 floats per primitive, in BVH order) lets the flattener find subtrees whose
// box does not contain all their primitives and mark the references to them REF_NOCULL.
//
// Every pass is a walk of the tree, and a tree of millions of nodes is walked as a small top part (the first nodes
// in breadth-first order, sequentially) plus a few thousand disjoint subtrees hanging off it (in parallel): the
// subtrees are validated, measured and boxed on their own, the top part stitches the results together, and the
// depth-first numbering gives every subtree a contiguous run of indices that it then fills in on its own.
static int flatten_tree(const PtxBvhNode* nodes, uint32_t n_nodes, uint32_t n_prims, const float* prim_boxes, FlatTree& out) {
	const char* not_a_tree = "node array is not a tree rooted at 0 over the given primitives";
	if (n_nodes == 0) return fail(PTX_ERR_INVALID, not_a_tree);
	auto is_leaf = [&](uint32_t i) { return nodes[i].l == nodes[i].r; };
	auto node_ok = [&](uint32_t i) {
		const PtxBvhNode& n = nodes[i];
		if (uint64_t(n.start) + n.size > n_prims) return false;
		return n.l == n.r || (n.l < n_nodes && n.r < n_nodes && n.l != i && n.r != i);
	};
	// ---- the top part: nodes expanded in breadth-first order until enough subtrees hang off it
	const size_t want_subtrees = n_nodes >= (1u << 17) ? size_t(build_threads()) * 128 : 0;
	struct Top {
		uint32_t node, depth;
	};
	std::vector<Top> top;            // expanded (interior) nodes, parents before children
	std::vector<Top> roots;          // subtree roots (interior or leaf), not expanded
	{
		std::vector<Top> queue{{0u, 0u}};
		size_t head = 0;
		while (head < queue.size()) {
			const Top t = queue[head++];
			if (!node_ok(t.node)) return fail(PTX_ERR_INVALID, not_a_tree);
			const bool expand = !is_leaf(t.node) && (queue.size() - head) + roots.size() < want_subtrees + 1 && top.size() < size_t(n_nodes);
			// (the first TOP_NODES interior nodes in breadth-first order must all lie in the top part)
			if (expand || (!is_leaf(t.node) && top.size() < size_t(TOP_NODES) + 1)) {
				top.push_back(t);
				queue.push_back({nodes[t.node].l, t.depth + 1}), queue.push_back({nodes[t.node].r, t.depth + 1});
				if (queue.size() > size_t(n_nodes) + 2) return fail(PTX_ERR_INVALID, not_a_tree);
			} else {
				roots.push_back(t);
			}
		}
	}
	// ---- per subtree: validation, depth, interior nodes, and the box its primitives really occupy per node
	std::vector<uint8_t> sticks_out(n_nodes, 0);
	BigVector<float> real(size_t(n_nodes) * 6);
	auto real_box = [&](uint32_t i) {   // children (or primitives) first: called in an order where they are done
		const PtxBvhNode& n = nodes[i];
		float* b = &real[size_t(i) * 6];
		for (int a = 0; a < 3; a++) b[a] = 3.402823466e+38f, b[3 + a] = -3.402823466e+38f;
		auto grow = [&](const float* o) {
			for (int a = 0; a < 3; a++) b[a] = std::min(b[a], o[a]), b[3 + a] = std::max(b[3 + a], o[3 + a]);
		};
		if (n.l == n.r) {
			for (uint32_t p = n.start; p < n.start + n.size; p++) grow(prim_boxes + size_t(p) * 6);
		} else {
			grow(&real[size_t(n.l) * 6]);
			grow(&real[size_t(n.r) * 6]);
		}
		bool out_of_box = false;
		for (int a = 0; a < 3; a++) out_of_box |= (b[a] < n.bmin[a]) || (b[3 + a] > n.bmax[a]);
		const bool holds_any = n.l != n.r || n.size > 0;
		sticks_out[i] = (holds_any && out_of_box) ? 1 : 0;
	};
	struct Measure {
		uint64_t visited = 0;
		uint32_t interior = 0, depth = 0;
		bool ok = true;
	};
	std::vector<Measure> measure(roots.size());
	parallel_tasks(roots.size(), n_nodes, [&](size_t k) {
		Measure m;
		std::vector<uint32_t> order;   // the subtree's nodes, parents before children
		std::vector<Top> stack{roots[k]};
		while (!stack.empty() && m.ok) {
			const Top t = stack.back();
			stack.pop_back();
			if (++m.visited > n_nodes || !node_ok(t.node)) {
				m.ok = false;
				break;
			}
			order.push_back(t.node);
			m.depth = std::max(m.depth, t.depth);
			if (is_leaf(t.node)) continue;
			m.interior++;
			stack.push_back({nodes[t.node].r, t.depth + 1}), stack.push_back({nodes[t.node].l, t.depth + 1});
		}
		if (m.ok)
			for (size_t j = order.size(); j-- > 0;) real_box(order[j]);
		measure[k] = m;
	});
	uint64_t visited = top.size();
	uint32_t depth = 0;
	for (const Top& t : top) depth = std::max(depth, t.depth);
	for (const Measure& m : measure) {
		if (!m.ok) return fail(PTX_ERR_INVALID, not_a_tree);
		visited += m.visited;
		depth = std::max(depth, m.depth);
	}
	if (visited > n_nodes) return fail(PTX_ERR_INVALID, not_a_tree);   // more visits than nodes: not a tree
	for (size_t j = top.size(); j-- > 0;) real_box(top[j].node);        // breadth-first order reversed: children first
	out.depth = depth;

	auto leaf_ref = [&](uint32_t i, uint32_t& ref) {
		const PtxBvhNode& n = nodes[i];
		if (n.size > MAX_LEAF_PRIMS || n.start > MAX_LEAF_START) return false;
		ref = make_leaf_ref(n.start, n.size) | (sticks_out[i] ? REF_NOCULL : 0u);
		return true;
	};
	const char* too_big = "a leaf holds more than 15 primitives (or starts beyond 2^26): not representable in the GPU layout";
	for (int k = 0; k < 3; k++) out.tame = out.tame && tame_coord_host(nodes[0].bmin[k]) && tame_coord_host(nodes[0].bmax[k]);
	if (is_leaf(0)) {
		if (!leaf_ref(0, out.root_ref)) return fail(PTX_ERR_UNSUPPORTED, too_big);
		return PTX_OK;
	}
	// ---- numbering of the interior nodes
	std::vector<uint32_t> interior_of(n_nodes, 0xFFFFFFFFu);
	uint32_t count = 0;
	// The first TOP_NODES interior nodes breadth-first: the top of the tree gets the lowest indices, which is
	// the part k_mesh_persistent keeps in LDS (persistent.hip.h).  `top` holds them in that very order.
	for (size_t j = 0; j < top.size() && count < uint32_t(TOP_NODES); j++) interior_of[top[j].node] = count++;
	// the others depth-first, left first: sequentially through the top part, a whole run of indices per subtree
	std::vector<uint32_t> root_index(n_nodes >= (1u << 17) ? n_nodes : 0, 0xFFFFFFFFu);   // node -> position in `roots` (large trees only)
	std::vector<uint32_t> first_index(roots.size(), 0u);
	{
		auto root_of = [&](uint32_t node) -> size_t {
			if (!root_index.empty()) return root_index[node];
			for (size_t k = 0; k < roots.size(); k++)
				if (roots[k].node == node) return k;
			return roots.size();
		};
		for (size_t k = 0; k < roots.size() && !root_index.empty(); k++) root_index[roots[k].node] = uint32_t(k);
		std::vector<uint8_t> in_top(n_nodes, 0);
		for (const Top& t : top) in_top[t.node] = 1;
		std::vector<uint32_t> stack{0u};
		while (!stack.empty()) {
			const uint32_t i = stack.back();
			stack.pop_back();
			if (is_leaf(i)) continue;
			if (in_top[i]) {
				if (interior_of[i] == 0xFFFFFFFFu) interior_of[i] = count++;
				stack.push_back(nodes[i].r), stack.push_back(nodes[i].l);
			} else {
				const size_t k = root_of(i);
				first_index[k] = count;
				count += measure[k].interior;
			}
		}
	}
	if (count > REF_INDEX_MASK) return fail(PTX_ERR_UNSUPPORTED, "more than 2^30 interior nodes");
	parallel_tasks(roots.size(), n_nodes, [&](size_t k) {
		if (is_leaf(roots[k].node)) return;
		uint32_t next = first_index[k];
		std::vector<uint32_t> stack{roots[k].node};
		while (!stack.empty()) {
			const uint32_t i = stack.back();
			stack.pop_back();
			interior_of[i] = next++;
			const PtxBvhNode& n = nodes[i];
			if (!is_leaf(n.r)) stack.push_back(n.r);
			if (!is_leaf(n.l)) stack.push_back(n.l);
		}
	});
	// ---- the records
	out.nodes.resize(count);
	std::atomic<bool> tame{out.tame}, leaf_too_big{false};
	std::atomic<uint32_t> n_nocull{0};
	parallel_slices(n_nodes, [&](uint64_t lo, uint64_t hi) {
		bool slice_tame = true;
		uint32_t slice_nocull = 0;
		for (uint64_t i = lo; i < hi; i++) {
			if (interior_of[i] == 0xFFFFFFFFu) continue;
			const PtxBvhNode &n = nodes[i], &l = nodes[n.l], &r = nodes[n.r];
			Node2 d{};
			// An empty leaf (the reference's "no split separates the primitives" case, bvh.cpp:183-221)
			// carries an inverted FLT_MAX box; whether a ray "hits" it changes nothing, so store zeros.
			const float zeros[3] = {0.0f, 0.0f, 0.0f};
			const bool l_empty = l.l == l.r && l.size == 0, r_empty = r.l == r.r && r.size == 0;
			d.set_box(0, l_empty ? zeros : l.bmin, l_empty ? zeros : l.bmax);
			d.set_box(1, r_empty ? zeros : r.bmin, r_empty ? zeros : r.bmax);
			for (int k = 0; k < 4; k++) slice_tame = slice_tame && tame_coord_host(d.lxy[k]) && tame_coord_host(d.z[k]) && tame_coord_host(d.rxy[k]);
			if (l.l == l.r) {
				if (!leaf_ref(n.l, d.lref)) leaf_too_big = true;
			} else d.lref = interior_of[n.l] | (sticks_out[n.l] ? REF_NOCULL : 0u);
			if (r.l == r.r) {
				if (!leaf_ref(n.r, d.rref)) leaf_too_big = true;
			} else d.rref = interior_of[n.r] | (sticks_out[n.r] ? REF_NOCULL : 0u);
			slice_nocull += (d.lref & REF_NOCULL ? 1 : 0) + (d.rref & REF_NOCULL ? 1 : 0);
			out.nodes[interior_of[i]] = d;
		}
		if (!slice_tame) tame = false;
		n_nocull += slice_nocull;
	});
	if (leaf_too_big) return fail(PTX_ERR_UNSUPPORTED, too_big);
	out.tame = tame, out.n_nocull = n_nocull;
	out.root_ref = 0;
	return PTX_OK;
}

// Four-wide form of a flattened binary tree: every interior child of a binary node is replaced by its own two
// children when both of their boxes lie inside the child's box (coordinate by coordinate, exactly).  The slab
// test is monotone under box inclusion -- subtraction, division and the min/max merges are -- so a ray that passes
// a grandchild's test passes the skipped child's too, and the set of leaves reached is that of the binary tree.
// (The reference builder can produce a child box that pokes out of its parent's, bvh.cpp:191-221: such a child
// stays a child of the wide node, with its own box test.)
static void widen_tree(const FlatTree& in, std::vector<Node4>& out, uint32_t& root_ref, uint32_t& skipped_levels) {
	out.clear();
	skipped_levels = 0;
	root_ref = in.root_ref;
	if (in.nodes.empty() || (in.root_ref & REF_LEAF)) return;
	struct Kid {
		float lo[3], hi[3];
		uint32_t ref;   // binary-tree reference (leaf ref, or interior index | flags)
	};
	auto kids_of = [&](uint32_t idx, Kid k[2]) {
		const Node2& n = in.nodes[idx];
		n.get_box(0, k[0].lo, k[0].hi), n.get_box(1, k[1].lo, k[1].hi);
		k[0].ref = n.lref, k[1].ref = n.rref;
	};
	auto inside = [](const Kid& g, const Kid& c) {
		for (int a = 0; a < 3; a++)
			if (!(g.lo[a] >= c.lo[a] && g.hi[a] <= c.hi[a])) return false;
		return true;
	};
	// breadth of work: wide nodes are created in the order their binary parents are reached, depth first
	std::vector<uint32_t> wide_of(in.nodes.size(), 0xFFFFFFFFu);
	std::vector<uint32_t> stack{0u};
	wide_of[0] = 0;
	out.emplace_back();
	while (!stack.empty()) {
		const uint32_t b = stack.back();
		stack.pop_back();
		Kid two[2], four[4];
		int n4 = 0;
		kids_of(b, two);
		for (int c = 0; c < 2; c++) {
			bool opened = false;
			if (!(two[c].ref & REF_LEAF)) {
				Kid g[2];
				kids_of(two[c].ref & REF_INDEX_MASK, g);
				if (inside(g[0], two[c]) && inside(g[1], two[c])) {
					four[n4++] = g[0], four[n4++] = g[1];
					opened = true;
					skipped_levels++;
				}
			}
			if (!opened) four[n4++] = two[c];
		}
		Node4 w{};
		uint32_t wref[4] = {REF_NONE4, REF_NONE4, REF_NONE4, REF_NONE4};
		for (int c = 0; c < n4; c++) {
			Node2& hf = w.half[c >> 1];
			hf.set_box(c & 1, four[c].lo, four[c].hi);
			if (four[c].ref & REF_LEAF) {
				wref[c] = four[c].ref;
			} else {
				const uint32_t bi = four[c].ref & REF_INDEX_MASK;
				if (wide_of[bi] == 0xFFFFFFFFu) {
					wide_of[bi] = uint32_t(out.size());
					out.emplace_back();
					stack.push_back(bi);
				}
				wref[c] = wide_of[bi] | (four[c].ref & REF_NOCULL);
			}
		}
		w.half[0].lref = wref[0], w.half[0].rref = wref[1], w.half[1].lref = wref[2], w.half[1].rref = wref[3];
		out[wide_of[b]] = w;
	}
	root_ref = 0;
}

template<typename T, typename A> static int upload(PtxScene* s, const std::vector<T, A>& v, const T*& dev) {
	dev = nullptr;
	if (v.empty()) return PTX_OK;
	void* p = nullptr;
	PTX_HIP(hipMalloc(&p, v.size() * sizeof(T)));
	s->allocations.push_back(p);
	PTX_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
	dev = static_cast<const T*>(p);
	return PTX_OK;
}

static int check_mesh_desc(const PtxMeshDesc& m) {
	if (m.n_tris && (!m.verts || !m.indices)) return fail(PTX_ERR_INVALID, "mesh with triangles but null verts/indices");
	std::atomic<bool> bad_index{false};
	parallel_slices(uint64_t(m.n_tris) * 3, [&](uint64_t lo, uint64_t hi) {
		bool bad = false;
		for (uint64_t i = lo; i < hi; i++) bad |= m.indices[i] >= m.n_verts;
		if (bad) bad_index = true;
	});
	if (bad_index) return fail(PTX_ERR_INVALID, "triangle index out of range");
	const uint32_t max_leaf = m.max_leaf_size ? m.max_leaf_size : 4;
	if (m.use_bvh && !m.nodes && max_leaf > MAX_LEAF_PRIMS) return fail(PTX_ERR_UNSUPPORTED, "max_leaf_size above 15 is not representable in the GPU layout");
	return PTX_OK;
}

struct SceneBuild {
	BigVector<Node2> nodes;
	std::vector<Node4> nodes4;
	bool want_wide = false;   // single-mesh scenes: also build the four-wide form of the tree
	BigVector<TriGeom> tris;
	BigVector<TriAttr> attrs;
	std::vector<MeshRec> meshes;
	std::vector<std::vector<float>> mesh_box;  // Tri_Mesh::bbox per mesh (6 floats)
};

// PT::Tri_Mesh(const Indexed_Mesh&, bool) -- tri_mesh.cpp:160-177
static int add_mesh(PtxScene* s, SceneBuild& sb, const PtxMeshDesc& m) {
	if (int rc = check_mesh_desc(m)) return rc;
	MeshRec rec{};
	rec.node_base = uint32_t(sb.nodes.size());
	rec.tri_base = uint32_t(sb.tris.size());
	rec.n_tris = m.n_tris;
	rec.is_list = m.use_bvh ? 0 : 1;
	HostBvh bvh;
	bool on_device = false;   // the tree came from the GPU builder
	std::vector<float> boxes;
	BuildLaps laps;
	triangle_boxes(m.verts, m.indices, m.n_tris, boxes);
	laps.lap("mesh: triangle boxes");
	float mesh_box[6];
	if (m.nodes) {
		if (!m.n_nodes) return fail(PTX_ERR_INVALID, "injected tree with zero nodes");
		bvh.nodes.assign(m.nodes, m.nodes + m.n_nodes);
		bvh.order.resize(m.n_tris);
		for (uint32_t i = 0; i < m.n_tris; i++) bvh.order[i] = i;
	} else if (m.use_bvh) {
		// Large meshes are built on the GPU the scene lives on (the same tree, byte for byte; a few times faster);
		// PTX_GPU_BUILD=0 keeps the host builder, =1 sends every mesh to the GPU.
		const int gpu_build = getenv("PTX_GPU_BUILD") ? atoi(getenv("PTX_GPU_BUILD")) : -1;
		const uint32_t leaf = m.max_leaf_size ? m.max_leaf_size : 4;
		bool built = false;
		if (gpu_build == 1 || (gpu_build != 0 && m.n_tris >= (1u << 17))) {
			// test switch: make the GPU builder's attempt fail as an out-of-memory device would
			const bool force_fail = getenv("PTX_GPU_BUILD_FORCE_FAIL") && atoi(getenv("PTX_GPU_BUILD_FORCE_FAIL")) != 0;
			const int rc = force_fail ? fail(PTX_ERR_OOM, "GPU builder: forced failure (PTX_GPU_BUILD_FORCE_FAIL)") : build_bvh_gpu(s->device, boxes.data(), m.n_tris, leaf, bvh, on_device);
			if (rc == PTX_OK) {
				built = true;
			} else if (gpu_build == 1) {
				return rc;   // the caller asked for the GPU builder: its failure is the call's failure
			} else {
				// The GPU builder needs about 240 B of transient device memory per primitive; on a GPU that is already
				// full (render state, other scenes) it can fail where the host builder, which gives the same tree byte
				// for byte, cannot.  It was this library's choice, not the caller's: fall back.
				(void)hipGetLastError();
				set_error("");
				bvh = HostBvh{};
				on_device = false;
			}
		}
		if (!built) build_bvh(boxes.data(), m.n_tris, leaf, bvh);
	} else {
		bvh.order.resize(m.n_tris);
		for (uint32_t i = 0; i < m.n_tris; i++) bvh.order[i] = i;
	}
	laps.lap("mesh: tree");
	if (!rec.is_list) {
		FlatTree ft;
		BigVector<float> ordered(boxes.size());
		parallel_slices(m.n_tris, [&](uint64_t lo, uint64_t hi) {
			for (uint64_t i = lo; i < hi; i++) std::copy(&boxes[size_t(bvh.order[i]) * 6], &boxes[size_t(bvh.order[i]) * 6] + 6, &ordered[size_t(i) * 6]);
		});
		if (int rc = flatten_tree(bvh.nodes.data(), uint32_t(bvh.nodes.size()), m.n_tris, ordered.data(), ft)) return rc;
		laps.lap("mesh: GPU node layout");
		bvh.max_depth = ft.depth;
		rec.root_ref = ft.root_ref;
		rec.tame = ft.tame ? 1 : 0;
		for (int k = 0; k < 3; k++) rec.root_min[k] = bvh.nodes[0].bmin[k], rec.root_max[k] = bvh.nodes[0].bmax[k];
		const bool wide = sb.want_wide;
		if (sb.nodes.empty() && !wide) sb.nodes = std::move(ft.nodes);   // the usual case for a big mesh: no copy
		else sb.nodes.insert(sb.nodes.end(), ft.nodes.begin(), ft.nodes.end());
		if (sb.want_wide) {
			std::vector<Node4> wide;
			uint32_t skipped = 0;
			widen_tree(ft, wide, rec.root4_ref, skipped);
			rec.node4_base = uint32_t(sb.nodes4.size());
			sb.nodes4.insert(sb.nodes4.end(), wide.begin(), wide.end());
			laps.lap("mesh: four-wide node layout");
		}
		// BVH::bbox (bvh.cpp:433-436)
		for (int k = 0; k < 3; k++) mesh_box[k] = bvh.nodes[0].bmin[k], mesh_box[3 + k] = bvh.nodes[0].bmax[k];
	} else {
		// List::bbox (list.h:17-23): union of the primitive boxes, FLT_MAX box when empty
		for (int k = 0; k < 3; k++) mesh_box[k] = 3.402823466e+38f, mesh_box[3 + k] = -3.402823466e+38f;
		for (uint32_t t = 0; t < m.n_tris; t++)
			for (int k = 0; k < 3; k++) {
				mesh_box[k] = std::min(mesh_box[k], boxes[size_t(t) * 6 + k]);
				mesh_box[3 + k] = std::max(mesh_box[3 + k], boxes[size_t(t) * 6 + 3 + k]);
			}
	}
	sb.mesh_box.emplace_back(mesh_box, mesh_box + 6);
	const size_t tri0 = sb.tris.size();
	sb.tris.resize(tri0 + m.n_tris);
	sb.attrs.resize(tri0 + m.n_tris);
	parallel_slices(m.n_tris, [&](uint64_t lo, uint64_t hi) {
	for (uint64_t i = lo; i < hi; i++) {
		const size_t t = bvh.order[i];
		const float* a = m.verts + size_t(m.indices[t * 3 + 0]) * 8;
		const float* b = m.verts + size_t(m.indices[t * 3 + 1]) * 8;
		const float* c = m.verts + size_t(m.indices[t * 3 + 2]) * 8;
		TriGeom g{};
		TriAttr at{};
		for (int k = 0; k < 3; k++) {
			g.v0[k] = a[k];
			g.e1[k] = b[k] - a[k];   // tri_mesh.cpp:58-59
			g.e2[k] = c[k] - a[k];
			at.n0[k] = a[3 + k], at.n1[k] = b[3 + k], at.n2[k] = c[3 + k];
		}
		for (int k = 0; k < 2; k++) at.uv0[k] = a[6 + k], at.uv1[k] = b[6 + k], at.uv2[k] = c[6 + k];
		sb.tris[tri0 + i] = g;
		sb.attrs[tri0 + i] = at;
	}
	});
	laps.lap("mesh: triangle records");
	sb.meshes.push_back(rec);
	s->mesh_bvhs.push_back(std::move(bvh));
	s->mesh_built_on_device.push_back(on_device ? 1 : 0);
	return PTX_OK;
}

static int finish_geometry(PtxScene* s, SceneBuild& sb) {
	BuildLaps laps;
	s->bytes_nodes = sb.nodes.size() * sizeof(Node2);
	// k_mesh_persistent copies the first TOP_NODES records of a mesh's node array into LDS whatever the tree's size:
	// zero records behind the last tree keep that read inside the allocation.
	if (TOP_NODES > 0) sb.nodes.insert(sb.nodes.end(), size_t(TOP_NODES), Node2{});
	if (int rc = upload(s, sb.nodes, s->rec.nodes)) return rc;
	if (int rc = upload(s, sb.nodes4, s->rec.nodes4)) return rc;
	if (int rc = upload(s, sb.tris, s->rec.tris)) return rc;
	if (int rc = upload(s, sb.attrs, s->rec.attrs)) return rc;
	if (int rc = upload(s, sb.meshes, s->rec.meshes)) return rc;
	s->shade.attrs = s->rec.attrs;
	s->bytes_tris = sb.tris.size() * sizeof(TriGeom);
	laps.lap("upload");
	return PTX_OK;
}

// ------------------------------------------------------------------ fixed device scratch (ptx_trace_closest_noalloc)
// Callers that may not allocate (HIP graph capture, tight serving loops) get the replenishing kernel's
// working buffers from this file-scope device array instead of hipMalloc: every GPU the library runs on
// holds one copy until the process ends.  Layout for up to NOALLOC_MAX_RAYS rays:
//   [counters 1 KB][slow list n][keys n, final order n][(key, ray index) pairs 2n][table + chunk totals]     (n in uint32)
constexpr uint64_t NOALLOC_MAX_RAYS = PTX_NOALLOC_MAX_RAYS;
constexpr size_t noalloc_table_words() {
	return size_t((SORT_DIGITS * uint32_t((NOALLOC_MAX_RAYS + SORT_TILE - 1) / SORT_TILE) + SCAN_CHUNK - 1) / SCAN_CHUNK) * (SCAN_CHUNK + 1);
}
constexpr size_t NOALLOC_OFF_SLOW = COUNTER_BLOCK_BYTES;
constexpr size_t NOALLOC_OFF_KEYS = NOALLOC_OFF_SLOW + NOALLOC_MAX_RAYS * 4;
constexpr size_t NOALLOC_OFF_IDX = NOALLOC_OFF_KEYS + NOALLOC_MAX_RAYS * 8;
constexpr size_t NOALLOC_OFF_TABLE = NOALLOC_OFF_IDX + NOALLOC_MAX_RAYS * 8;
constexpr size_t NOALLOC_SCRATCH_BYTES = NOALLOC_OFF_TABLE + noalloc_table_words() * 4;
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

// The array's address on each device.  In host code the name `scratch` is only a handle for
// hipGetSymbolAddress -- never an address a kernel may be given -- and every GPU has its own copy at
// its own address, so the lookup runs once per device, with that device current, when the first scene
// is created there (scene creation is this library's per-device set-up step).
constexpr int NOALLOC_MAX_DEVICES = 64;
static std::mutex g_noalloc_mutex;
static unsigned char* g_noalloc_ptr[NOALLOC_MAX_DEVICES] = {};

static int noalloc_scratch_for_device(int device, unsigned char** out) {
	*out = nullptr;
	if (device < 0 || device >= NOALLOC_MAX_DEVICES) return PTX_OK;   // such a device simply has no allocation-free path
	std::lock_guard<std::mutex> lock(g_noalloc_mutex);
	if (!g_noalloc_ptr[device]) {
		void* p = nullptr;
		PTX_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
		g_noalloc_ptr[device] = static_cast<unsigned char*>(p);
	}
	*out = g_noalloc_ptr[device];
	return PTX_OK;
}

static int open_device(int device, PtxScene** out) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(PTX_ERR_NO_DEVICE, "no HIP device visible");
	if (device < 0 || device >= n) return fail(PTX_ERR_NO_DEVICE, "device index out of range");
	PTX_HIP(hipSetDevice(device));
	*out = new PtxScene();
	(*out)->device = device;
	int cus = 0;
	if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
	(*out)->cu_count = uint32_t(cus);
	if (int rc = noalloc_scratch_for_device(device, &(*out)->noalloc_scratch)) {
		delete *out;
		*out = nullptr;
		return rc;
	}
	return PTX_OK;
}

static void free_scene(PtxScene* s) {
	if (!s) return;
	(void)hipSetDevice(s->device);
	for (void* p : s->allocations) (void)hipFree(p);
	for (void* p : s->scratch)
		if (p) (void)hipFree(p);
	for (uint32_t i = 0; i < PtxScene::TIMING_RING; i++) {
		if (s->t_start[i]) (void)hipEventDestroy(s->t_start[i]);
		if (s->t_stop[i]) (void)hipEventDestroy(s->t_stop[i]);
	}
	for (hipEvent_t e : s->chunk_events) (void)hipEventDestroy(e);
	if (s->up_stream) (void)hipStreamDestroy(s->up_stream);
	if (s->down_stream) (void)hipStreamDestroy(s->down_stream);
	delete s;
}

static void mat_copy(float* dst, const float* src) { std::memcpy(dst, src, 16 * sizeof(float)); }

// Single-mesh scenes whose tree is tame run on the persistent kernel (persistent.hip.h).
static bool use_persistent(const PtxScene* s) { return s->rec.mesh_only && s->persistent_ok; }

int scene_scratch(PtxScene* s, int slot, size_t bytes) {
	if (s->scratch_bytes[slot] >= bytes) return PTX_OK;
	if (s->scratch[slot]) PTX_HIP(hipFree(s->scratch[slot]));
	s->scratch[slot] = nullptr, s->scratch_bytes[slot] = 0;
	PTX_HIP(hipMalloc(&s->scratch[slot], bytes));
	s->scratch_bytes[slot] = bytes;
	return PTX_OK;
}
int scene_counter_block(PtxScene* s, unsigned long long** out) {
	if (!s->next_ray) {
		void* p = nullptr;
		PTX_HIP(hipMalloc(&p, 2 * COUNTER_BLOCK_BYTES));   // one block per pass of a replenishing launch
		s->allocations.push_back(p);
		s->next_ray = static_cast<unsigned long long*>(p);
	}
	*out = s->next_ray;
	return PTX_OK;
}
static int grow_scratch(PtxScene* s, int slot, size_t bytes) { return scene_scratch(s, slot, bytes); }

// Table length of the grouping pass for n rays: the digit-major table of per-tile counts, padded with
// zeros to whole scan chunks, followed by the chunk totals.
static void sort_table_shape(uint64_t n, uint32_t& n_tiles, uint32_t& n_chunks, uint32_t& table_len) {
	n_tiles = uint32_t((n + SORT_TILE - 1) / SORT_TILE);
	n_chunks = (SORT_DIGITS * n_tiles + SCAN_CHUNK - 1) / SCAN_CHUNK;
	table_len = n_chunks * SCAN_CHUNK;
}

// Device buffers of one replenishing-kernel launch: the scene's grown-on-demand scratch for the
// ordinary entry points, slices of the fixed device array for the allocation-free one.
struct PersistentBuffers {
	unsigned long long* counters;   // COUNTER_BLOCK_BYTES, zeroed by the launch
	uint32_t* slow_list;            // n entries: rays the first pass sets aside
	uint32_t* slow_list2;           // n entries: rays the second (zero-aware) pass leaves to the plain-division kernel; null: no second pass
	uint32_t* keys;                 // 2n: keys, keys after pass 0          (grouping pass only)
	uint32_t* idx;                  // 2n: indices after pass 0, final order (grouping pass only)
	uint32_t* table;                // table_len + n_chunks of sort_table_shape(n)
	bool timed;                     // record the launch in the scene's ring of HIP events
};

// Groups the batch by origin cell and direction octant (raysort.hip.h); *order receives the
// device array of ray indices in that order.
static int sort_rays(PtxScene* s, const PtxRay* d_rays, uint64_t n, const PersistentBuffers& pb, bool cost_classes, const uint32_t** order) {
	uint32_t n_tiles, n_chunks, table_len;
	sort_table_shape(n, n_tiles, n_chunks, table_len);
	uint32_t* keys0 = pb.keys;                          // pb.keys: 2n words -- the keys, then the final order
	uint32_t* ord = keys0 + n;
	uint2* pairs = reinterpret_cast<uint2*>(pb.idx);    // pb.idx: 2n words, 8-byte aligned -- the first pass's (key, ray index) pairs
	uint32_t* table = pb.table;
	uint32_t* totals = table + table_len;
	auto scan = [&]() {
		hipLaunchKernelGGL(k_scan_local, dim3(n_chunks), dim3(256), 0, s->stream, table, totals);
		hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(256), 0, s->stream, totals, n_chunks);
		hipLaunchKernelGGL(k_scan_add, dim3(n_chunks), dim3(256), 0, s->stream, table, totals);
	};
	const PtxBvhNode& root = s->mesh_bvhs[0].nodes[0];
	const float cells = float(1u << SORT_GRID_BITS);
	float3 lo = make_float3(root.bmin[0], root.bmin[1], root.bmin[2]), scale;
	const float ex = root.bmax[0] - root.bmin[0], ey = root.bmax[1] - root.bmin[1], ez = root.bmax[2] - root.bmin[2];
	scale.x = ex > 0.0f ? cells / ex : 0.0f, scale.y = ey > 0.0f ? cells / ey : 0.0f, scale.z = ez > 0.0f ? cells / ez : 0.0f;
	// Cost classes of the grouping key: chords below 7.5 % and 15 % of the root box's mean extent are queued last
	// (16.7 M rays: 1 % faster, 4 M: 4 %).  Small batches keep one class: there the thinner classes cost more
	// coherence than the shorter tail gains (1 M rays: 4 % slower with classes), and so do occlusion queries,
	// whose rays end at their first hit (16.7 M shadow rays: 3 % slower with classes).
	static const float f1 = getenv("PTX_SORT_CLASS_T1") ? float(atof(getenv("PTX_SORT_CLASS_T1"))) : 0.075f;
	static const float f2 = getenv("PTX_SORT_CLASS_T2") ? float(atof(getenv("PTX_SORT_CLASS_T2"))) : 0.15f;
	const float3 hi = make_float3(root.bmax[0], root.bmax[1], root.bmax[2]);
	const float mean_extent = cost_classes && n >= (uint64_t(1) << 21) ? (ex + ey + ez) / 3.0f : 0.0f;
	const dim3 block(256), grid(n_tiles);
	const size_t pad = size_t(table_len) - size_t(SORT_DIGITS) * n_tiles;
	if (pad) PTX_HIP(hipMemsetAsync(table + size_t(SORT_DIGITS) * n_tiles, 0, pad * sizeof(uint32_t), s->stream));
	hipLaunchKernelGGL(k_sort_count_rays, grid, block, 0, s->stream, d_rays, n, lo, scale, hi, f1 * mean_extent, f2 * mean_extent, keys0, table, n_tiles);
	scan();
	hipLaunchKernelGGL(k_sort_scatter<true>, grid, block, 0, s->stream, keys0, static_cast<const uint2*>(nullptr), n, table, n_tiles, pairs, static_cast<uint32_t*>(nullptr));
	hipLaunchKernelGGL(k_sort_count_keys, grid, block, 0, s->stream, pairs, n, table, n_tiles);
	scan();
	hipLaunchKernelGGL(k_sort_scatter<false>, grid, block, 0, s->stream, static_cast<const uint32_t*>(nullptr), pairs, n, table, n_tiles, static_cast<uint2*>(nullptr), ord);
	*order = ord;
	return PTX_OK;
}

static bool wants_grouping(bool sort, uint64_t n) { return sort && n >= (1u << 16) && n < (uint64_t(1) << 32); }

// The scene's own buffers for a launch over n rays, grown where too small.
static int scene_persistent_buffers(PtxScene* s, bool sort, uint64_t n, PersistentBuffers& pb) {
	pb = PersistentBuffers{};
	pb.timed = true;
	if (wants_grouping(sort, n)) {
		uint32_t n_tiles, n_chunks, table_len;
		sort_table_shape(n, n_tiles, n_chunks, table_len);
		if (int rc = grow_scratch(s, 4, 2 * n * sizeof(uint32_t))) return rc;
		if (int rc = grow_scratch(s, 5, 2 * n * sizeof(uint32_t))) return rc;
		if (int rc = grow_scratch(s, 6, (size_t(table_len) + n_chunks) * sizeof(uint32_t))) return rc;
		pb.keys = static_cast<uint32_t*>(s->scratch[4]);
		pb.idx = static_cast<uint32_t*>(s->scratch[5]);
		pb.table = static_cast<uint32_t*>(s->scratch[6]);
	}
	if (int rc = scene_counter_block(s, &pb.counters)) return rc;
	if (int rc = grow_scratch(s, 7, n * sizeof(uint32_t))) return rc;
	pb.slow_list = static_cast<uint32_t*>(s->scratch[7]);
	if (int rc = grow_scratch(s, 8, n * sizeof(uint32_t))) return rc;
	pb.slow_list2 = static_cast<uint32_t*>(s->scratch[8]);
	return PTX_OK;
}

template<bool ANY>
static int launch_persistent(PtxScene* s, bool culled, bool sort, const PersistentBuffers& pb, const PtxRay* d_rays, uint64_t n, PtxHit* d_hits, float* d_bary,
                             uint32_t* d_inst, uint8_t* d_occ) {
	if (n >= (uint64_t(1) << 32)) return fail(PTX_ERR_INVALID, "ray count too large for one launch");
	const uint32_t* order = nullptr;
	if (wants_grouping(sort, n))
		if (int rc = sort_rays(s, d_rays, n, pb, !ANY, &order)) return rc;
	unsigned long long* const next_ray = pb.counters;
	PTX_HIP(hipMemsetAsync(next_ray, 0, (pb.slow_list2 ? 2 : 1) * COUNTER_BLOCK_BYTES, s->stream));
	uint32_t* slow_list = pb.slow_list;
	uint32_t* slow_count = reinterpret_cast<uint32_t*>(next_ray) + 16;   // same zeroed block, its own 64-byte line
	// enough 256-thread blocks to fill every CU at 8 waves per SIMD; each wave pulls rays until the pool is dry
	const uint64_t want = (n + 255) / 256;
	// Tuning knobs for experiments (not part of the ABI): blocks per CU and the refill threshold.
	static const int blocks_per_cu = getenv("PTX_BLOCKS_PER_CU") ? atoi(getenv("PTX_BLOCKS_PER_CU")) : 8;
	static const int refill = getenv("PTX_REFILL_MIN_IDLE") ? atoi(getenv("PTX_REFILL_MIN_IDLE")) : REFILL_MIN_IDLE;
	static const char* tri_env = getenv("PTX_TRI_MIN_LANES");
	const int tri_min = tri_env ? atoi(tri_env) : (culled ? PARK_MIN_LANES : TRI_MIN_LANES);
	static const int stuck_max = getenv("PTX_STUCK_MAX_LANES") ? atoi(getenv("PTX_STUCK_MAX_LANES")) : STUCK_MAX_LANES;
	static const unsigned long long wave_chunk = getenv("PTX_WAVE_CHUNK") ? std::max(1, atoi(getenv("PTX_WAVE_CHUNK"))) : WAVE_CHUNK;
	const dim3 block(256), grid(uint32_t(std::min<uint64_t>(want, uint64_t(s->cu_count) * uint64_t(std::max(1, blocks_per_cu)))));
	const uint32_t slot = uint32_t(s->timed_launches % PtxScene::TIMING_RING);
	if (pb.timed) {
		if (!s->t_start[slot]) {
			PTX_HIP(hipEventCreate(&s->t_start[slot]));
			PTX_HIP(hipEventCreate(&s->t_stop[slot]));
		}
		PTX_HIP(hipEventRecord(s->t_start[slot], s->stream));
	}
	// Experiment switch: walk the four-wide form of the tree.  Measured 5 % slower than the binary walk on the 1M-triangle
	// soup (half the loop iterations, but the same number of box tests and L1 accesses per ray, 8 more registers and
	// therefore one wave per SIMD fewer), so it is off unless asked for.
	static const bool wide = getenv("PTX_WIDE_TREE") && atoi(getenv("PTX_WIDE_TREE")) != 0;
	if (culled && wide && s->rec.nodes4) hipLaunchKernelGGL((k_mesh_persistent<MODE_CULLED, ANY, true>), grid, block, 0, s->stream, s->rec, d_rays, n, d_hits, d_bary, d_inst, d_occ, next_ray, refill, order, slow_list, slow_count, tri_min, stuck_max, wave_chunk, nullptr);
	else if (culled) hipLaunchKernelGGL((k_mesh_persistent<MODE_CULLED, ANY>), grid, block, 0, s->stream, s->rec, d_rays, n, d_hits, d_bary, d_inst, d_occ, next_ray, refill, order, slow_list, slow_count, tri_min, stuck_max, wave_chunk, nullptr);
	else hipLaunchKernelGGL((k_mesh_persistent<MODE_REFERENCE, ANY>), grid, block, 0, s->stream, s->rec, d_rays, n, d_hits, d_bary, d_inst, d_occ, next_ray, refill, order, slow_list, slow_count, tri_min, stuck_max, wave_chunk, nullptr);
	if (pb.timed) {
		PTX_HIP(hipEventRecord(s->t_stop[slot], s->stream));
		s->timed_launches++;
	}
#ifdef PTX_PROFILE_LANES
	{
		unsigned long long h[32];
		PTX_HIP(hipStreamSynchronize(s->stream));
		PTX_HIP(hipMemcpy(h, next_ray, sizeof(h), hipMemcpyDeviceToHost));
		std::fprintf(stderr, "[lanes] wave-iterations %llu, node phase ran in %llu with %.1f lanes, triangle phase ran in %llu with %.1f lanes, lanes holding a ray %.1f\n", h[16], h[19],
		             double(h[17]) / double(h[19] ? h[19] : 1), h[20], double(h[18]) / double(h[20] ? h[20] : 1), double(h[21]) / double(h[16] ? h[16] : 1));
		const double t0 = double(~h[22]), t_dry = double(~h[23]), t_end = double(h[24]);
		const double mean_out = double(h[25]) / double(h[26] ? h[26] : 1) - double((~h[22]) & 0xFFFFFFFFFFull);
		std::fprintf(stderr, "[tail] %llu rays: kernel %.1f us, pool first dry at %.1f us, waves left on average at %.1f us (%llu waves)\n", (unsigned long long)n, (t_end - t0) / 100.0,
		             (t_dry - t0) / 100.0, mean_out / 100.0, h[26]);
	}
#endif
	// Second pass: the rays set aside walk the same kernel in its zero-aware form (most of them have a direction
	// component that is exactly zero); what that pass cannot take either goes to the plain-division kernel.
	if (pb.slow_list2) {
		unsigned long long* const next_ray2 = next_ray + COUNTER_BLOCK_BYTES / sizeof(unsigned long long);
		uint32_t* slow_count2 = reinterpret_cast<uint32_t*>(next_ray2) + 16;
		if (culled) hipLaunchKernelGGL((k_mesh_persistent<MODE_CULLED, ANY, false, true>), grid, block, 0, s->stream, s->rec, d_rays, n, d_hits, d_bary, d_inst, d_occ, next_ray2, refill, slow_list, pb.slow_list2, slow_count2, tri_min, stuck_max, wave_chunk, slow_count);
		else hipLaunchKernelGGL((k_mesh_persistent<MODE_REFERENCE, ANY, false, true>), grid, block, 0, s->stream, s->rec, d_rays, n, d_hits, d_bary, d_inst, d_occ, next_ray2, refill, slow_list, pb.slow_list2, slow_count2, tri_min, stuck_max, wave_chunk, slow_count);
		slow_list = pb.slow_list2, slow_count = slow_count2;
	}
	const dim3 dgrid(uint32_t(std::min<uint64_t>(want, 1024)));
	if (culled) hipLaunchKernelGGL((k_mesh_deferred<MODE_CULLED, ANY>), dgrid, block, 0, s->stream, s->rec, d_rays, slow_list, slow_count, d_hits, d_bary, d_inst, d_occ);
	else hipLaunchKernelGGL((k_mesh_deferred<MODE_REFERENCE, ANY>), dgrid, block, 0, s->stream, s->rec, d_rays, slow_list, slow_count, d_hits, d_bary, d_inst, d_occ);
	return PTX_OK;
}

}  // namespace ptx

using namespace ptx;

extern "C" {

const char* ptx_last_error(void) { return g_last_error.c_str(); }
const char* ptx_version(void) { return "pathtracer_amd 0.1 (gfx950)"; }

int ptx_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// ------------------------------------------------------------------ host BVH
int ptx_bvh_build(const float* boxes, uint32_t n, uint32_t max_leaf_size, ptx_bvh_t* out) {
	if (!out || (n && !boxes) || max_leaf_size == 0) return fail(PTX_ERR_INVALID, "ptx_bvh_build: null argument or max_leaf_size 0");
	auto* b = new PtxBvh();
	build_bvh(boxes, n, max_leaf_size, b->bvh);
	*out = b;
	return PTX_OK;
}

int ptx_bvh_build_triangles(const float* verts, uint32_t n_verts, const uint32_t* indices, uint32_t n_tris, uint32_t max_leaf_size,
                            ptx_bvh_t* out) {
	if (!out || (n_tris && (!verts || !indices)) || max_leaf_size == 0) return fail(PTX_ERR_INVALID, "ptx_bvh_build_triangles: null argument or max_leaf_size 0");
	for (uint64_t i = 0; i < uint64_t(n_tris) * 3; i++)
		if (indices[i] >= n_verts) return fail(PTX_ERR_INVALID, "triangle index out of range");
	std::vector<float> boxes;
	triangle_boxes(verts, indices, n_tris, boxes);
	return ptx_bvh_build(boxes.data(), n_tris, max_leaf_size, out);
}

int ptx_bvh_build_gpu(int device, const float* boxes, uint32_t n, uint32_t max_leaf_size, ptx_bvh_t* out) {
	if (!out || (n && !boxes) || max_leaf_size == 0) return fail(PTX_ERR_INVALID, "ptx_bvh_build_gpu: null argument or max_leaf_size 0");
	auto* b = new PtxBvh();
	if (int rc = build_bvh_gpu(device, boxes, n, max_leaf_size, b->bvh, b->built_on_device)) {
		delete b;
		return rc;
	}
	*out = b;
	return PTX_OK;
}

int ptx_bvh_build_triangles_gpu(int device, const float* verts, uint32_t n_verts, const uint32_t* indices, uint32_t n_tris, uint32_t max_leaf_size,
                                ptx_bvh_t* out) {
	if (!out || (n_tris && (!verts || !indices)) || max_leaf_size == 0) return fail(PTX_ERR_INVALID, "ptx_bvh_build_triangles_gpu: null argument or max_leaf_size 0");
	for (uint64_t i = 0; i < uint64_t(n_tris) * 3; i++)
		if (indices[i] >= n_verts) return fail(PTX_ERR_INVALID, "triangle index out of range");
	std::vector<float> boxes;
	triangle_boxes(verts, indices, n_tris, boxes);
	return ptx_bvh_build_gpu(device, boxes.data(), n_tris, max_leaf_size, out);
}

int ptx_bvh_built_on_device(ptx_bvh_t b) { return b && b->built_on_device ? 1 : 0; }

uint32_t ptx_bvh_num_nodes(ptx_bvh_t b) { return b ? uint32_t(b->bvh.nodes.size()) : 0; }
uint32_t ptx_bvh_num_prims(ptx_bvh_t b) { return b ? uint32_t(b->bvh.order.size()) : 0; }
uint32_t ptx_bvh_max_depth(ptx_bvh_t b) { return b ? b->bvh.max_depth : 0; }
int ptx_bvh_get_nodes(ptx_bvh_t b, PtxBvhNode* out) {
	if (!b || !out) return fail(PTX_ERR_INVALID, "ptx_bvh_get_nodes: null argument");
	std::copy(b->bvh.nodes.begin(), b->bvh.nodes.end(), out);
	return PTX_OK;
}
int ptx_bvh_get_order(ptx_bvh_t b, uint32_t* out) {
	if (!b || !out) return fail(PTX_ERR_INVALID, "ptx_bvh_get_order: null argument");
	std::copy(b->bvh.order.begin(), b->bvh.order.end(), out);
	return PTX_OK;
}
int ptx_bvh_get_stats(ptx_bvh_t b, uint64_t stats[3]) {
	if (!b || !stats) return fail(PTX_ERR_INVALID, "ptx_bvh_get_stats: null argument");
	stats[0] = b->bvh.clamped, stats[1] = b->bvh.predicate_disagreements, stats[2] = b->bvh.dropped;
	return PTX_OK;
}
void ptx_bvh_destroy(ptx_bvh_t b) { delete b; }

// ------------------------------------------------------------------ scenes
int ptx_scene_create_mesh(int device, const PtxMeshDesc* mesh, ptx_scene_t* out) {
	if (!mesh || !out) return fail(PTX_ERR_INVALID, "ptx_scene_create_mesh: null argument");
	PtxScene* s = nullptr;
	if (int rc = open_device(device, &s)) return rc;
	SceneBuild sb;
	sb.want_wide = getenv("PTX_WIDE_TREE") && atoi(getenv("PTX_WIDE_TREE")) != 0;   // experiment switch, see launch_persistent
	int rc = add_mesh(s, sb, *mesh);
	if (!rc) rc = finish_geometry(s, sb);
	if (!rc) {
		s->rec.mesh_only = 1;
		s->persistent_ok = !sb.meshes[0].is_list && sb.meshes[0].tame;
		s->stack_need = s->mesh_bvhs[0].max_depth + 2;
		if (s->stack_need > uint32_t(STACK_CAP)) rc = fail(PTX_ERR_UNSUPPORTED, "tree deeper than the traversal stack (64 entries)");
	}
	if (rc) {
		free_scene(s);
		return rc;
	}
	*out = s;
	return PTX_OK;
}

int ptx_scene_create(int device, const PtxSceneDesc* d, ptx_scene_t* out) {
	if (!d || !out) return fail(PTX_ERR_INVALID, "ptx_scene_create: null argument");
	if ((d->n_meshes && !d->meshes) || (d->n_instances && !d->instances) || (d->n_textures && !d->textures) ||
	    (d->n_materials && !d->materials) || (d->n_delta_lights && !d->delta_lights) || (d->n_env_lights && !d->env_lights))
		return fail(PTX_ERR_INVALID, "ptx_scene_create: count without array");
	PtxScene* s = nullptr;
	if (int rc = open_device(device, &s)) return rc;
	auto bail = [&](int rc) {
		free_scene(s);
		return rc;
	};
	SceneBuild sb;
	for (uint32_t i = 0; i < d->n_meshes; i++)
		if (int rc = add_mesh(s, sb, d->meshes[i])) return bail(rc);

	// Textures, materials (+ the default Lambertian(0), pathtracer.cpp:240-252), lights.
	std::vector<TexRec> texs;
	std::vector<float> texels;
	for (uint32_t i = 0; i < d->n_textures; i++) {
		const auto& t = d->textures[i];
		TexRec r{};
		r.type = t.type, r.sampler = t.sampler, r.w = t.w, r.h = t.h, r.scale = t.scale;
		for (int k = 0; k < 3; k++) r.color[k] = t.color[k];
		if (t.type == PTX_TEX_IMAGE) {
			if (!t.pixels || !t.w || !t.h) return bail(fail(PTX_ERR_INVALID, "image texture without pixels"));
			r.pixel_base = uint32_t(texels.size() / 3);
			texels.insert(texels.end(), t.pixels, t.pixels + size_t(t.w) * t.h * 3);
		} else if (t.type != PTX_TEX_CONSTANT) return bail(fail(PTX_ERR_INVALID, "unknown texture type"));
		texs.push_back(r);
	}
	const int default_tex = int(texs.size());
	{
		TexRec r{};
		r.type = PTX_TEX_CONSTANT, r.scale = 1.0f;
		texs.push_back(r);
	}
	std::vector<MatRec> mats;
	std::vector<bool> material_ok;
	for (uint32_t i = 0; i < d->n_materials; i++) {
		const auto& m = d->materials[i];
		if (m.type > PTX_MAT_EMISSIVE) return bail(fail(PTX_ERR_INVALID, "unknown material type"));
		const bool needs_a = m.type != PTX_MAT_REFRACT, needs_b = m.type == PTX_MAT_REFRACT || m.type == PTX_MAT_GLASS;
		// A material whose texture reference dangles may sit unused in a scene file (the reference
		// keeps an expired weak_ptr there and only fails if the material is ever evaluated): such a
		// material is accepted, and rejected only when an instance uses it.
		const bool bad_a = needs_a && (m.tex_a < 0 || m.tex_a >= default_tex), bad_b = needs_b && (m.tex_b < 0 || m.tex_b >= default_tex);
		material_ok.push_back(!(bad_a || bad_b));
		mats.push_back(MatRec{m.type, bad_a ? default_tex : m.tex_a, bad_b ? default_tex : m.tex_b, m.ior});
	}
	const int default_mat = int(mats.size());
	mats.push_back(MatRec{PTX_MAT_LAMBERTIAN, default_tex, -1, 1.5f});
	std::vector<DeltaLightRec> dls;
	for (uint32_t i = 0; i < d->n_delta_lights; i++) {
		const auto& l = d->delta_lights[i];
		if (l.type > PTX_LIGHT_SPOT) return bail(fail(PTX_ERR_INVALID, "unknown delta light type"));
		DeltaLightRec r{};
		r.type = l.type, r.intensity = l.intensity, r.inner_angle = l.inner_angle, r.outer_angle = l.outer_angle;
		for (int k = 0; k < 3; k++) r.color[k] = l.color[k];
		Mat4f T;
		mat_copy(T.m, l.T);
		Mat4f iT = mat4_inverse(T);
		mat_copy(r.T, T.m), mat_copy(r.iT, iT.m);
		r.has_transform = mat4_is_identity(T) ? 0 : 1;
		dls.push_back(r);
	}
	std::vector<EnvLightRec> els;
	for (uint32_t i = 0; i < d->n_env_lights; i++) {
		const auto& l = d->env_lights[i];
		if (l.type > PTX_ENV_SPHERE || l.tex < 0 || l.tex >= default_tex) return bail(fail(PTX_ERR_INVALID, "bad environment light"));
		els.push_back(EnvLightRec{l.type, l.tex});
	}

	// Instances (instance.h:17-30) and the top-level tree (pathtracer.cpp:381-385).
	std::vector<InstRec> insts_in;
	std::vector<float> inst_boxes;
	uint32_t deepest_mesh = 0;
	for (uint32_t i = 0; i < d->n_instances; i++) {
		const auto& in = d->instances[i];
		if (in.mesh >= int32_t(d->n_meshes) || in.material >= int32_t(d->n_materials)) return bail(fail(PTX_ERR_INVALID, "instance references a mesh or material that does not exist"));
		InstRec r{};
		Mat4f T;
		mat_copy(T.m, in.T);
		Mat4f iT = mat4_inverse(T);
		mat_copy(r.T, T.m), mat_copy(r.iT, iT.m);
		r.has_transform = mat4_is_identity(T) ? 0 : 1;
		r.mesh = in.mesh < 0 ? -1 : in.mesh;
		r.radius = in.sphere_radius;
		if (in.material >= 0 && !material_ok[size_t(in.material)]) return bail(fail(PTX_ERR_INVALID, "an instance uses a material that references a texture that does not exist"));
		r.material = in.material < 0 ? default_mat : in.material;
		float box[6];
		if (r.mesh >= 0) {
			std::copy(sb.mesh_box[size_t(r.mesh)].begin(), sb.mesh_box[size_t(r.mesh)].end(), box);
			deepest_mesh = std::max(deepest_mesh, s->mesh_bvhs[size_t(r.mesh)].max_depth);
		} else {
			for (int k = 0; k < 3; k++) box[k] = -r.radius, box[3 + k] = r.radius;   // shape.cpp:14-19
		}
		if (r.has_transform) box_transform(T, box);
		inst_boxes.insert(inst_boxes.end(), box, box + 6);
		insts_in.push_back(r);
	}
	std::vector<InstRec> insts(insts_in.size());
	s->rec.top_is_list = d->use_bvh ? 0 : 1;
	s->rec.n_insts = d->n_instances;
	if (d->use_bvh) {
		build_bvh(inst_boxes.data(), d->n_instances, 1, s->top_bvh);
		FlatTree ft;
		std::vector<float> ordered(inst_boxes.size());
		for (uint32_t i = 0; i < d->n_instances; i++)
			std::copy(&inst_boxes[size_t(s->top_bvh.order[i]) * 6], &inst_boxes[size_t(s->top_bvh.order[i]) * 6] + 6, &ordered[size_t(i) * 6]);
		if (int rc = flatten_tree(s->top_bvh.nodes.data(), uint32_t(s->top_bvh.nodes.size()), d->n_instances, ordered.data(), ft)) return bail(rc);
		s->top_bvh.max_depth = ft.depth;
		s->rec.top_ref = ft.root_ref;
		s->rec.top_tame = ft.tame ? 1 : 0;
		s->rec.top_stack_slots = ft.depth + 1;
		s->rec.top_node_base = uint32_t(sb.nodes.size());
		for (int k = 0; k < 3; k++) s->rec.top_min[k] = s->top_bvh.nodes[0].bmin[k], s->rec.top_max[k] = s->top_bvh.nodes[0].bmax[k];
		sb.nodes.insert(sb.nodes.end(), ft.nodes.begin(), ft.nodes.end());
	} else {
		s->top_bvh.order.resize(d->n_instances);
		for (uint32_t i = 0; i < d->n_instances; i++) s->top_bvh.order[i] = i;
	}
	for (uint32_t i = 0; i < d->n_instances; i++) insts[i] = insts_in[s->top_bvh.order[i]];
	{   // the mesh most instances refer to (ties: the larger one)
		std::vector<uint32_t> uses(d->n_meshes, 0u);
		for (const InstRec& r : insts)
			if (r.mesh >= 0) uses[size_t(r.mesh)]++;
		uint32_t best = 0;
		for (uint32_t m = 1; m < d->n_meshes; m++)
			if (uses[m] > uses[best] || (uses[m] == uses[best] && d->meshes[m].n_tris > d->meshes[best].n_tris)) best = m;
		s->rec.lds_mesh = best;
	}
	s->stack_need = s->top_bvh.max_depth + deepest_mesh + 3;
	if (s->stack_need > uint32_t(STACK_CAP)) return bail(fail(PTX_ERR_UNSUPPORTED, "trees deeper than the traversal stack (64 entries)"));

	if (int rc = finish_geometry(s, sb)) return bail(rc);
	if (int rc = upload(s, insts, s->rec.insts)) return bail(rc);
	if (int rc = upload(s, texs, s->shade.textures)) return bail(rc);
	if (int rc = upload(s, texels, s->shade.texels)) return bail(rc);
	if (int rc = upload(s, mats, s->shade.materials)) return bail(rc);
	if (int rc = upload(s, dls, s->shade.delta_lights)) return bail(rc);
	if (int rc = upload(s, els, s->shade.env_lights)) return bail(rc);
	s->shade.n_delta_lights = uint32_t(dls.size());
	s->shade.n_env_lights = uint32_t(els.size());
	s->shade.fresnel_mode = d->fresnel_mode;
	{
		// Instances whose material can emit (bsdf->emission() is zero for every other material, material.cpp).
		uint32_t n = 0, prims = 0;
		bool fits = true;
		for (uint32_t i = 0; i < d->n_instances && fits; i++) {
			if (mats[size_t(insts[i].material)].type != PTX_MAT_EMISSIVE) continue;
			prims += insts[i].mesh >= 0 ? d->meshes[insts[i].mesh].n_tris : 1u;
			if (n == MAX_EMISSIVE_INSTS || prims > MAX_EMISSIVE_PRIMS) fits = false;
			else s->shade.emissive_insts[n++] = i;
		}
		s->shade.n_emissive = fits ? n : 0;
		s->shade.direct_skip = (fits && els.empty()) ? 1 : 0;
	}
	*out = s;
	return PTX_OK;
}

void ptx_scene_destroy(ptx_scene_t s) { free_scene(s); }

int ptx_scene_set_stream(ptx_scene_t s, void* hip_stream) {
	if (!s) return fail(PTX_ERR_INVALID, "ptx_scene_set_stream: null scene");
	s->stream = static_cast<hipStream_t>(hip_stream);
	return PTX_OK;
}

uint32_t ptx_scene_mesh_num_nodes(ptx_scene_t s, uint32_t mesh) {
	return (s && mesh < s->mesh_bvhs.size()) ? uint32_t(s->mesh_bvhs[mesh].nodes.size()) : 0;
}
int ptx_scene_mesh_get_nodes(ptx_scene_t s, uint32_t mesh, PtxBvhNode* out) {
	if (!s || !out || mesh >= s->mesh_bvhs.size()) return fail(PTX_ERR_INVALID, "ptx_scene_mesh_get_nodes: bad argument");
	std::copy(s->mesh_bvhs[mesh].nodes.begin(), s->mesh_bvhs[mesh].nodes.end(), out);
	return PTX_OK;
}
int ptx_scene_mesh_built_on_device(ptx_scene_t s, uint32_t mesh) {
	return (s && mesh < s->mesh_built_on_device.size()) ? int(s->mesh_built_on_device[mesh]) : 0;
}
int ptx_scene_mesh_get_order(ptx_scene_t s, uint32_t mesh, uint32_t* out) {
	if (!s || !out || mesh >= s->mesh_bvhs.size()) return fail(PTX_ERR_INVALID, "ptx_scene_mesh_get_order: bad argument");
	std::copy(s->mesh_bvhs[mesh].order.begin(), s->mesh_bvhs[mesh].order.end(), out);
	return PTX_OK;
}
int ptx_scene_get_instance_order(ptx_scene_t s, uint32_t* out) {
	if (!s || !out) return fail(PTX_ERR_INVALID, "ptx_scene_get_instance_order: bad argument");
	std::copy(s->top_bvh.order.begin(), s->top_bvh.order.end(), out);
	return PTX_OK;
}

int ptx_get_kernel_times(ptx_scene_t s, float* ms, uint32_t cap, uint32_t* n_out) {
	if (!s || !n_out || (cap && !ms)) return fail(PTX_ERR_INVALID, "ptx_get_kernel_times: null argument");
	PTX_HIP(hipSetDevice(s->device));
	PTX_HIP(hipStreamSynchronize(s->stream));
	const uint64_t have = std::min<uint64_t>(s->timed_launches, PtxScene::TIMING_RING);
	const uint32_t n = uint32_t(std::min<uint64_t>(have, cap));
	for (uint32_t k = 0; k < n; k++) {   // oldest of the last n first
		const uint32_t slot = uint32_t((s->timed_launches - n + k) % PtxScene::TIMING_RING);
		PTX_HIP(hipEventElapsedTime(&ms[k], s->t_start[slot], s->t_stop[slot]));
	}
	*n_out = n;
	return PTX_OK;
}

int ptx_get_counters(ptx_scene_t s, PtxCounters* out) {
	if (!s || !out) return fail(PTX_ERR_INVALID, "ptx_get_counters: null argument");
	*out = s->counters;
	if (s->scratch[9]) {   // the render's ray totals live on the device (render.hip, k_round_tally)
		unsigned long long t[2] = {};
		PTX_HIP(hipSetDevice(s->device));
		PTX_HIP(hipMemcpyAsync(t, s->scratch[9], sizeof(t), hipMemcpyDeviceToHost, s->stream));
		PTX_HIP(hipStreamSynchronize(s->stream));
		out->render_main_rays = t[0], out->render_direct_rays = t[1];
	}
	return PTX_OK;
}

// ------------------------------------------------------------------ ray queries
static int ensure_scratch(PtxScene* s, int slot, size_t bytes) {
	if (s->scratch_bytes[slot] >= bytes) return PTX_OK;
	if (s->scratch[slot]) PTX_HIP(hipFree(s->scratch[slot]));
	s->scratch[slot] = nullptr, s->scratch_bytes[slot] = 0;
	PTX_HIP(hipMalloc(&s->scratch[slot], bytes));
	s->scratch_bytes[slot] = bytes;
	return PTX_OK;
}

// One array of a host-buffer ray query: `stride` bytes per ray at `host`, staged at `dev`.
struct HostSpan {
	void* dev;
	void* host;
	size_t stride;
};

static bool host_pinned(const void* p) {
	hipPointerAttribute_t a{};
	if (hipPointerGetAttributes(&a, p) != hipSuccess) {
		(void)hipGetLastError();   // plain malloc'ed memory is reported as an error by some runtimes
		return false;
	}
	return a.type == hipMemoryTypeHost;
}

// Rays per chunk of a host-buffer query: the batch whole below two chunks' worth (the grouping pass works
// better on large batches: 4 M rays trace at 0.8 of the rate of 16 M), else equal chunks of about 4 M rays.
static uint64_t host_chunk_rays(uint64_t n) {
	uint64_t want = uint64_t(1) << 22;
	if (const char* e = std::getenv("PTX_HOST_CHUNK_RAYS")) want = std::strtoull(e, nullptr, 10);
	if (want == 0 || n / want < 2) return n;
	const uint64_t chunks = n / want;
	return ((n + chunks - 1) / chunks + 63) & ~uint64_t(63);
}

// A ray query on host buffers: `launch(first, count)` enqueues the traversal of a range of the staged rays
// on the scene's stream.  Large batches go in chunks, the upload of chunk c+1 and the download of chunk c-1
// on streams of their own beside the traversal of chunk c.  Copies to pageable memory block the calling
// thread, so those downloads are issued from a helper thread.  Returns with everything complete.
static int host_pipeline(PtxScene* s, uint64_t n, const HostSpan& in, const HostSpan* outs, int n_outs, const std::function<int(uint64_t, uint64_t)>& launch) {
	const uint64_t chunk = host_chunk_rays(n);
	if (chunk >= n) {
		PTX_HIP(hipMemcpyAsync(in.dev, in.host, n * in.stride, hipMemcpyHostToDevice, s->stream));
		if (int rc = launch(uint64_t(0), n)) return rc;
		for (int k = 0; k < n_outs; k++) PTX_HIP(hipMemcpyAsync(outs[k].host, outs[k].dev, n * outs[k].stride, hipMemcpyDeviceToHost, s->stream));
		PTX_HIP(hipStreamSynchronize(s->stream));
		return PTX_OK;
	}
	const uint32_t n_chunks = uint32_t((n + chunk - 1) / chunk);
	if (!s->up_stream) PTX_HIP(hipStreamCreateWithFlags(&s->up_stream, hipStreamNonBlocking));
	if (!s->down_stream) PTX_HIP(hipStreamCreateWithFlags(&s->down_stream, hipStreamNonBlocking));
	while (s->chunk_events.size() < size_t(2) * n_chunks + 1) {
		hipEvent_t e = nullptr;
		PTX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		s->chunk_events.push_back(e);
	}
	hipEvent_t* uploaded = s->chunk_events.data();
	hipEvent_t* traced = uploaded + n_chunks;
	hipEvent_t begun = s->chunk_events[size_t(2) * n_chunks];
	bool pinned_out = true;
	for (int k = 0; k < n_outs; k++) pinned_out = pinned_out && host_pinned(outs[k].host);

	auto download = [&](uint32_t c) -> hipError_t {
		const uint64_t first = uint64_t(c) * chunk, count = std::min(chunk, n - first);
		hipError_t e = hipStreamWaitEvent(s->down_stream, traced[c], 0);
		for (int k = 0; k < n_outs && e == hipSuccess; k++)
			e = hipMemcpyAsync(static_cast<char*>(outs[k].host) + first * outs[k].stride, static_cast<char*>(outs[k].dev) + first * outs[k].stride,
			                   count * outs[k].stride, hipMemcpyDeviceToHost, s->down_stream);
		return e;
	};

	std::mutex mu;
	std::condition_variable cv;
	uint32_t recorded = 0;      // chunks whose `traced` event is on the stream
	bool stop = false;
	hipError_t down_err = hipSuccess;
	std::thread helper;
	if (!pinned_out)
		helper = std::thread([&] {
			if ((down_err = hipSetDevice(s->device)) != hipSuccess) return;
			for (uint32_t c = 0; c < n_chunks; c++) {
				{
					std::unique_lock<std::mutex> lk(mu);
					cv.wait(lk, [&] { return recorded > c || stop; });
					if (recorded <= c) return;
				}
				if ((down_err = download(c)) != hipSuccess) return;
			}
		});

	auto enqueue = [&]() -> int {
		// the staging buffers may still be read by earlier work on the scene's stream
		PTX_HIP(hipEventRecord(begun, s->stream));
		PTX_HIP(hipStreamWaitEvent(s->up_stream, begun, 0));
		for (uint32_t c = 0; c < n_chunks; c++) {
			const uint64_t first = uint64_t(c) * chunk, count = std::min(chunk, n - first);
			PTX_HIP(hipMemcpyAsync(static_cast<char*>(in.dev) + first * in.stride, static_cast<const char*>(in.host) + first * in.stride, count * in.stride,
			                       hipMemcpyHostToDevice, s->up_stream));
			PTX_HIP(hipEventRecord(uploaded[c], s->up_stream));
			PTX_HIP(hipStreamWaitEvent(s->stream, uploaded[c], 0));
			if (int rc = launch(first, count)) return rc;
			PTX_HIP(hipEventRecord(traced[c], s->stream));
			if (pinned_out) {
				PTX_HIP(download(c));
			} else {
				std::lock_guard<std::mutex> lk(mu);
				recorded = c + 1;
				cv.notify_one();
			}
		}
		return PTX_OK;
	};
	const int rc = enqueue();
	if (helper.joinable()) {
		{
			std::lock_guard<std::mutex> lk(mu);
			stop = true;
			cv.notify_one();
		}
		helper.join();
	}
	// whatever was enqueued finishes before the caller's buffers are given back
	const hipError_t e_up = hipStreamSynchronize(s->up_stream), e_main = hipStreamSynchronize(s->stream), e_down = hipStreamSynchronize(s->down_stream);
	if (rc) return rc;
	PTX_HIP(down_err);
	PTX_HIP(e_up);
	PTX_HIP(e_main);
	PTX_HIP(e_down);
	return PTX_OK;
}

// Closest-hit traversal of device-resident rays, enqueued on the scene's stream.
static int trace_closest_resident(PtxScene* s, const PtxRay* d_rays, uint64_t n, PtxHit* d_hits, float* d_bary, uint32_t* d_inst, uint32_t flags) {
	const dim3 block(256), grid(uint32_t((n + 255) / 256));
	const bool culled = (flags & PTX_TRAVERSE_MASK) == PTX_TRAVERSE_CULLED;
	if (use_persistent(s) && !(flags & PTX_NO_PERSISTENT)) {
		PersistentBuffers pb;
		if (int rc = scene_persistent_buffers(s, !(flags & PTX_RAYS_COHERENT), n, pb)) return rc;
		if (int rc = launch_persistent<false>(s, culled, !(flags & PTX_RAYS_COHERENT), pb, d_rays, n, d_hits, d_bary, d_inst, nullptr)) return rc;
	} else if (s->rec.mesh_only) {
		if (culled) hipLaunchKernelGGL((k_trace_closest<MODE_CULLED, true>), grid, block, 0, s->stream, s->rec, d_rays, n, d_hits, d_bary, d_inst);
		else hipLaunchKernelGGL((k_trace_closest<MODE_REFERENCE, true>), grid, block, 0, s->stream, s->rec, d_rays, n, d_hits, d_bary, d_inst);
	} else if (scene_persistent_ok(s) && !(flags & PTX_NO_PERSISTENT) && n < (uint64_t(1) << 32)) {
		if (culled) {
			if (int rc = launch_scene_persistent(s, BatchIO{d_rays, n, d_hits, d_bary, d_inst}, n)) return rc;
		} else {
			if (int rc = (launch_scene_persistent<BatchIO, false, MODE_REFERENCE>(s, BatchIO{d_rays, n, d_hits, d_bary, d_inst}, n))) return rc;
		}
	} else {
		if (culled) hipLaunchKernelGGL((k_trace_closest<MODE_CULLED, false>), grid, block, 0, s->stream, s->rec, d_rays, n, d_hits, d_bary, d_inst);
		else hipLaunchKernelGGL((k_trace_closest<MODE_REFERENCE, false>), grid, block, 0, s->stream, s->rec, d_rays, n, d_hits, d_bary, d_inst);
	}
	PTX_HIP(hipGetLastError());
	return PTX_OK;
}

int ptx_trace_closest(ptx_scene_t s, const PtxRay* rays, uint64_t n, PtxHit* hits, float* bary, uint32_t* inst, uint32_t flags) {
	if (!s || (n && (!rays || !hits))) return fail(PTX_ERR_INVALID, "ptx_trace_closest: null argument");
	if ((flags & PTX_ASYNC) && !(flags & PTX_MEM_DEVICE)) return fail(PTX_ERR_INVALID, "PTX_ASYNC needs PTX_MEM_DEVICE");
	if (n == 0) return PTX_OK;
	if (n > (uint64_t(1) << 39)) return fail(PTX_ERR_INVALID, "ray count too large for one launch");
	PTX_HIP(hipSetDevice(s->device));
	if (flags & PTX_MEM_DEVICE) {
		if (int rc = trace_closest_resident(s, rays, n, hits, bary, inst, flags)) return rc;
		s->counters.rays_closest += n;
		if (!(flags & PTX_ASYNC)) PTX_HIP(hipStreamSynchronize(s->stream));
		return PTX_OK;
	}
	if (int rc = ensure_scratch(s, 0, n * sizeof(PtxRay))) return rc;
	if (int rc = ensure_scratch(s, 1, n * sizeof(PtxHit))) return rc;
	if (bary)
		if (int rc = ensure_scratch(s, 2, n * 2 * sizeof(float))) return rc;
	if (inst)
		if (int rc = ensure_scratch(s, 3, n * sizeof(uint32_t))) return rc;
	const PtxRay* d_rays = static_cast<const PtxRay*>(s->scratch[0]);
	PtxHit* d_hits = static_cast<PtxHit*>(s->scratch[1]);
	float* d_bary = bary ? static_cast<float*>(s->scratch[2]) : nullptr;
	uint32_t* d_inst = inst ? static_cast<uint32_t*>(s->scratch[3]) : nullptr;
	HostSpan outs[3] = {{d_hits, hits, sizeof(PtxHit)}};
	int n_outs = 1;
	if (bary) outs[n_outs++] = {d_bary, bary, 2 * sizeof(float)};
	if (inst) outs[n_outs++] = {d_inst, inst, sizeof(uint32_t)};
	const int rc = host_pipeline(s, n, HostSpan{s->scratch[0], const_cast<PtxRay*>(rays), sizeof(PtxRay)}, outs, n_outs, [&](uint64_t first, uint64_t count) {
		return trace_closest_resident(s, d_rays + first, count, d_hits + first, d_bary ? d_bary + 2 * first : nullptr, d_inst ? d_inst + first : nullptr, flags);
	});
	if (rc) return rc;
	s->counters.rays_closest += n;
	return PTX_OK;
}

uint64_t ptx_noalloc_max_rays(void) { return NOALLOC_MAX_RAYS; }

int ptx_trace_closest_noalloc(ptx_scene_t s, const PtxRay* rays, uint64_t n, PtxHit* hits, float* bary, uint32_t* inst, uint32_t flags) {
	if (!s || (n && (!rays || !hits))) return fail(PTX_ERR_INVALID, "ptx_trace_closest_noalloc: null argument");
	if (!(flags & PTX_MEM_DEVICE)) return fail(PTX_ERR_INVALID, "ptx_trace_closest_noalloc: rays and hits must be device pointers (PTX_MEM_DEVICE); staging host buffers would allocate");
	if (n > NOALLOC_MAX_RAYS) return fail(PTX_ERR_INVALID, "ptx_trace_closest_noalloc: " + std::to_string(n) + " rays exceed the fixed scratch array's limit of " + std::to_string(NOALLOC_MAX_RAYS) + " per call (split the batch, or use ptx_trace_closest)");
	if (n == 0) return PTX_OK;
	PTX_HIP(hipSetDevice(s->device));
	const dim3 block(256), grid(uint32_t((n + 255) / 256));
	const bool culled = (flags & PTX_TRAVERSE_MASK) == PTX_TRAVERSE_CULLED;
	if (use_persistent(s) && !(flags & PTX_NO_PERSISTENT)) {
		unsigned char* base = s->noalloc_scratch;   // device address from hipGetSymbolAddress, not the array's name
		if (!base) return fail(PTX_ERR_UNSUPPORTED, "ptx_trace_closest_noalloc: no fixed scratch array on this device");
		PersistentBuffers pb{};   // no second list: this entry point keeps to its fixed array (zero-direction rays go to the plain-division kernel)
		pb.counters = reinterpret_cast<unsigned long long*>(base);
		pb.slow_list = reinterpret_cast<uint32_t*>(base + NOALLOC_OFF_SLOW);
		pb.keys = reinterpret_cast<uint32_t*>(base + NOALLOC_OFF_KEYS);
		pb.idx = reinterpret_cast<uint32_t*>(base + NOALLOC_OFF_IDX);
		pb.table = reinterpret_cast<uint32_t*>(base + NOALLOC_OFF_TABLE);
		pb.timed = false;   // the timing ring creates HIP events on first use
		if (int rc = launch_persistent<false>(s, culled, !(flags & PTX_RAYS_COHERENT), pb, rays, n, hits, bary, inst, nullptr)) return rc;
	} else if (s->rec.mesh_only) {
		// lists and trees outside the exact-division range: one ray per lane, no working buffers at all
		if (culled) hipLaunchKernelGGL((k_trace_closest<MODE_CULLED, true>), grid, block, 0, s->stream, s->rec, rays, n, hits, bary, inst);
		else hipLaunchKernelGGL((k_trace_closest<MODE_REFERENCE, true>), grid, block, 0, s->stream, s->rec, rays, n, hits, bary, inst);
	} else {
		if (culled) hipLaunchKernelGGL((k_trace_closest<MODE_CULLED, false>), grid, block, 0, s->stream, s->rec, rays, n, hits, bary, inst);
		else hipLaunchKernelGGL((k_trace_closest<MODE_REFERENCE, false>), grid, block, 0, s->stream, s->rec, rays, n, hits, bary, inst);
	}
	PTX_HIP(hipGetLastError());
	s->counters.rays_closest += n;
	if (!(flags & PTX_ASYNC)) PTX_HIP(hipStreamSynchronize(s->stream));
	return PTX_OK;
}

// Occlusion traversal of device-resident rays, enqueued on the scene's stream.
static int trace_any_resident(PtxScene* s, const PtxRay* d_rays, uint64_t n, uint8_t* d_occ, uint32_t flags) {
	const dim3 block(256), grid(uint32_t((n + 255) / 256));
	const bool culled = (flags & PTX_TRAVERSE_MASK) == PTX_TRAVERSE_CULLED;
	if (use_persistent(s) && !(flags & PTX_NO_PERSISTENT)) {
		PersistentBuffers pb;
		if (int rc = scene_persistent_buffers(s, !(flags & PTX_RAYS_COHERENT), n, pb)) return rc;
		if (int rc = launch_persistent<true>(s, culled, !(flags & PTX_RAYS_COHERENT), pb, d_rays, n, nullptr, nullptr, nullptr, d_occ)) return rc;
	} else if (s->rec.mesh_only) {
		if (culled) hipLaunchKernelGGL((k_trace_any<MODE_CULLED, true>), grid, block, 0, s->stream, s->rec, d_rays, n, d_occ);
		else hipLaunchKernelGGL((k_trace_any<MODE_REFERENCE, true>), grid, block, 0, s->stream, s->rec, d_rays, n, d_occ);
	} else if (scene_persistent_ok(s) && !(flags & PTX_NO_PERSISTENT) && n < (uint64_t(1) << 32)) {
		if (culled) {
			if (int rc = (launch_scene_persistent<BatchAnyIO, true>(s, BatchAnyIO{d_rays, n, d_occ}, n))) return rc;
		} else {
			if (int rc = (launch_scene_persistent<BatchAnyIO, true, MODE_REFERENCE>(s, BatchAnyIO{d_rays, n, d_occ}, n))) return rc;
		}
	} else {
		if (culled) hipLaunchKernelGGL((k_trace_any<MODE_CULLED, false>), grid, block, 0, s->stream, s->rec, d_rays, n, d_occ);
		else hipLaunchKernelGGL((k_trace_any<MODE_REFERENCE, false>), grid, block, 0, s->stream, s->rec, d_rays, n, d_occ);
	}
	PTX_HIP(hipGetLastError());
	return PTX_OK;
}

int ptx_trace_any(ptx_scene_t s, const PtxRay* rays, uint64_t n, uint8_t* occluded, uint32_t flags) {
	if (!s || (n && (!rays || !occluded))) return fail(PTX_ERR_INVALID, "ptx_trace_any: null argument");
	if ((flags & PTX_ASYNC) && !(flags & PTX_MEM_DEVICE)) return fail(PTX_ERR_INVALID, "PTX_ASYNC needs PTX_MEM_DEVICE");
	if (n == 0) return PTX_OK;
	if (n > (uint64_t(1) << 39)) return fail(PTX_ERR_INVALID, "ray count too large for one launch");
	PTX_HIP(hipSetDevice(s->device));
	if (flags & PTX_MEM_DEVICE) {
		if (int rc = trace_any_resident(s, rays, n, occluded, flags)) return rc;
		s->counters.rays_any += n;
		if (!(flags & PTX_ASYNC)) PTX_HIP(hipStreamSynchronize(s->stream));
		return PTX_OK;
	}
	if (int rc = ensure_scratch(s, 0, n * sizeof(PtxRay))) return rc;
	if (int rc = ensure_scratch(s, 1, n)) return rc;
	const PtxRay* d_rays = static_cast<const PtxRay*>(s->scratch[0]);
	uint8_t* d_occ = static_cast<uint8_t*>(s->scratch[1]);
	const HostSpan out{d_occ, occluded, 1};
	const int rc = host_pipeline(s, n, HostSpan{s->scratch[0], const_cast<PtxRay*>(rays), sizeof(PtxRay)}, &out, 1,
	                             [&](uint64_t first, uint64_t count) { return trace_any_resident(s, d_rays + first, count, d_occ + first, flags); });
	if (rc) return rc;
	s->counters.rays_any += n;
	return PTX_OK;
}

int ptx_trace_closest_stats(ptx_scene_t s, const PtxRay* rays, uint64_t n, uint32_t* stats3, uint32_t flags) {
	if (!s || (n && (!rays || !stats3))) return fail(PTX_ERR_INVALID, "ptx_trace_closest_stats: null argument");
	if (n == 0) return PTX_OK;
	PTX_HIP(hipSetDevice(s->device));
	const PtxRay* d_rays = rays;
	uint32_t* d_stats = stats3;
	const bool host = !(flags & PTX_MEM_DEVICE);
	if (host) {
		if (int rc = ensure_scratch(s, 0, n * sizeof(PtxRay))) return rc;
		if (int rc = ensure_scratch(s, 1, n * 3 * sizeof(uint32_t))) return rc;
		PTX_HIP(hipMemcpyAsync(s->scratch[0], rays, n * sizeof(PtxRay), hipMemcpyHostToDevice, s->stream));
		d_rays = static_cast<const PtxRay*>(s->scratch[0]);
		d_stats = static_cast<uint32_t*>(s->scratch[1]);
	}
	const dim3 block(256), grid(uint32_t((n + 255) / 256));
	if ((flags & PTX_TRAVERSE_MASK) == PTX_TRAVERSE_CULLED) hipLaunchKernelGGL(k_trace_stats<MODE_CULLED>, grid, block, 0, s->stream, s->rec, d_rays, n, d_stats);
	else hipLaunchKernelGGL(k_trace_stats<MODE_REFERENCE>, grid, block, 0, s->stream, s->rec, d_rays, n, d_stats);
	PTX_HIP(hipGetLastError());
	if (host) PTX_HIP(hipMemcpyAsync(stats3, d_stats, n * 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
	PTX_HIP(hipStreamSynchronize(s->stream));
	return PTX_OK;
}

int ptx_selftest_division(int device, uint64_t pairs, uint64_t seed, uint64_t* mismatches) {
	if (!mismatches) return fail(PTX_ERR_INVALID, "ptx_selftest_division: null argument");
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(PTX_ERR_NO_DEVICE, "device index out of range");
	PTX_HIP(hipSetDevice(device));
	unsigned long long* d = nullptr;
	PTX_HIP(hipMalloc(&d, sizeof(*d)));
	hipError_t e = hipMemset(d, 0, sizeof(*d));
	const uint32_t threads = 256 * 2048;
	const uint64_t per_thread = (pairs + threads - 1) / threads;
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_selftest_division, dim3(2048), dim3(256), 0, nullptr, per_thread, seed, d);
		e = hipGetLastError();
	}
	unsigned long long h = 0;
	if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
	(void)hipFree(d);
	if (e != hipSuccess) return hip_fail(e, "ptx_selftest_division");
	*mismatches = h;
	return PTX_OK;
}

// ------------------------------------------------------------------ tiles, accumulator, tonemap (host)
uint32_t ptx_make_tiles(uint32_t W, uint32_t H, uint32_t samples, uint32_t fixed_seed, PtxTile* out, uint32_t cap) {
	// pathtracer.cpp:522-563
	constexpr uint32_t tile_w = 100, tile_h = 100, tile_s = 50;
	std::mt19937 seeds(fixed_seed);
	std::vector<PtxTile> tiles;
	for (uint32_t y = 0; y < H; y += tile_h)
		for (uint32_t x = 0; x < W; x += tile_w)
			for (uint32_t sb = 0; sb < samples; sb += tile_s) {
				PtxTile t;
				t.seed = uint32_t(seeds());
				t.x_begin = x, t.x_end = std::min(x + tile_w, W);
				t.y_begin = y, t.y_end = std::min(y + tile_h, H);
				t.s_begin = sb, t.s_end = std::min(sb + tile_s, samples);
				tiles.push_back(t);
			}
	auto centre_distance = [&](const PtxTile& t) {
		const float dx = 0.5f * (t.x_begin + t.x_end) - W * 0.5f, dy = 0.5f * (t.y_begin + t.y_end) - H * 0.5f;
		return std::sqrt(dx * dx + dy * dy);
	};
	std::stable_sort(tiles.begin(), tiles.end(), [&](const PtxTile& a, const PtxTile& b) {
		const float da = centre_distance(a), db = centre_distance(b);
		if (da != db) return da < db;
		return a.s_begin < b.s_begin;
	});
	if (out)
		for (size_t i = 0; i < tiles.size() && i < cap; i++) out[i] = tiles[i];
	return uint32_t(tiles.size());
}

int ptx_accumulator_to_image(const int64_t* accum_rgb, const uint32_t* samples, uint64_t n_pixels, float* rgb) {
	if (n_pixels && (!accum_rgb || !samples || !rgb)) return fail(PTX_ERR_INVALID, "ptx_accumulator_to_image: null argument");
	for (uint64_t i = 0; i < n_pixels; i++)
		for (int c = 0; c < 3; c++)
			rgb[i * 3 + c] = samples[i] > 0 ? float(accum_rgb[i * 3 + c] / double(1ll << 24) / double(samples[i])) : 0.0f;
	return PTX_OK;
}

int ptx_tonemap(const float* rgb, uint64_t n_pixels, float exposure, uint8_t* rgba) {
	if (n_pixels && (!rgb || !rgba)) return fail(PTX_ERR_INVALID, "ptx_tonemap: null argument");
	auto srgb = [](float f) { return f > 0.0031308f ? 1.055f * (std::pow(f, (1.0f / 2.4f))) - 0.055f : f * 12.92f; };
	for (uint64_t i = 0; i < n_pixels; i++) {
		for (int c = 0; c < 3; c++) {
			const float v = 1.0f - std::exp(-rgb[i * 3 + c] * exposure);
			rgba[i * 4 + c] = static_cast<uint8_t>(std::round(srgb(v) * 255.0f));
		}
		rgba[i * 4 + 3] = 255;
	}
	return PTX_OK;
}

}  // extern "C"

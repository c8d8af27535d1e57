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

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>

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

unsigned build_threads() {
	static const unsigned n = [] {
		if (const char* e = std::getenv("PTX_BUILD_THREADS")) return unsigned(std::max(1, std::atoi(e)));
		unsigned c = std::thread::hardware_concurrency();
		cpu_set_t set;
		if (sched_getaffinity(0, sizeof(set), &set) == 0) c = unsigned(CPU_COUNT(&set));
		return std::max(1u, std::min(16u, c));
	}();
	return n;
}

void parallel_slices(uint64_t n, const std::function<void(uint64_t, uint64_t)>& fn) {
	const unsigned threads = n >= (1u << 17) ? build_threads() : 1u;
	if (threads <= 1) {
		fn(0, n);
		return;
	}
	std::vector<std::thread> pool;
	for (unsigned k = 1; k < threads; k++) pool.emplace_back([&, k]() { fn(n * k / threads, n * (k + 1) / threads); });
	fn(0, n / threads);
	for (auto& t : pool) t.join();
}

BuildLaps::BuildLaps() : on(std::getenv("PTX_BUILD_PROFILE") != nullptr), last(std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count()) {}
void BuildLaps::lap(const char* what) {
	const double now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
	if (on) std::fprintf(stderr, "[build] %-28s %.3f s\n", what, now - last);
	last = now;
}

void triangle_boxes(const float* verts, const uint32_t* indices, uint32_t n_tris, std::vector<float>& boxes) {
	boxes.resize(size_t(n_tris) * 6);
	const float eps = 0.0001f;
	parallel_slices(n_tris, [&](uint64_t t_begin, uint64_t t_end) {
	for (uint64_t t = t_begin; t < t_end; t++) {
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
	});
}

namespace {

// Worker threads kept for the length of one build: run(jobs, fn) calls fn(k) for k in [0, jobs) on the
// workers and the calling thread and returns when all calls have returned.  (Starting threads per call
// cost more than the work of most calls here.)
class Workers {
public:
	explicit Workers(unsigned threads) {
		for (unsigned t = 1; t < threads; t++) pool.emplace_back([this]() { loop(); });
	}
	~Workers() {
		{
			std::lock_guard<std::mutex> lock(m);
			quit = true;
		}
		wake.notify_all();
		for (auto& t : pool) t.join();
	}
	unsigned size() const { return unsigned(pool.size()) + 1u; }
	template<typename F>
	void run(unsigned jobs, F&& fn) {
		if (pool.empty() || jobs <= 1) {
			for (unsigned k = 0; k < jobs; k++) fn(k);
			return;
		}
		std::function<void(unsigned)> f = std::ref(fn);
		{
			std::lock_guard<std::mutex> lock(m);
			job = &f, n_jobs = jobs, next.store(0), busy = unsigned(pool.size()), generation++;
		}
		wake.notify_all();
		for (unsigned k = next++; k < jobs; k = next++) fn(k);
		std::unique_lock<std::mutex> lock(m);
		done.wait(lock, [this]() { return busy == 0; });
		job = nullptr;
	}

private:
	void loop() {
		uint64_t seen = 0;
		for (;;) {
			const std::function<void(unsigned)>* f;
			unsigned jobs;
			{
				std::unique_lock<std::mutex> lock(m);
				wake.wait(lock, [&]() { return quit || generation != seen; });
				if (quit) return;
				seen = generation, f = job, jobs = n_jobs;
			}
			for (unsigned k = next++; k < jobs; k = next++) (*f)(k);
			{
				std::lock_guard<std::mutex> lock(m);
				if (--busy == 0) done.notify_one();
			}
		}
	}
	std::vector<std::thread> pool;
	std::mutex m;
	std::condition_variable wake, done;
	const std::function<void(unsigned)>* job = nullptr;
	unsigned n_jobs = 0, busy = 0;
	std::atomic<unsigned> next{0};
	uint64_t generation = 0;
	bool quit = false;
};

struct Bins {
	Box box[3][NBINS];
	uint32_t n[3][NBINS] = {};
	uint64_t clamped = 0;
	void merge(const Bins& o) {
		for (int a = 0; a < 3; a++)
			for (int b = 0; b < NBINS; b++) box[a][b].grow(o.box[a][b]), n[a][b] += o.n[a][b];
		clamped += o.clamped;
	}
};

// A primitive as the builder moves it around: its box and the index it came with.  The records
// themselves are partitioned (not an index array over them), so every pass over a node's range
// streams through memory.
struct Prim {
	float box[6];
	uint32_t id;
	uint32_t pad;
	float centre(int axis) const { return (box[axis] + box[3 + axis]) * 0.5f; }   // BBox::center, lib/bbox.h:45-47
};

struct BuildInput {
	Prim* prims;
	uint32_t max_leaf_size;
};

struct Counters {
	uint64_t clamped = 0, predicate_disagreements = 0, dropped = 0;
	uint32_t max_depth = 0;
};

void bin_range(const BuildInput& in, const PtxBvhNode& parent, uint32_t begin, uint32_t end, Bins& bins) {
	float span_lo[3], bin_len[3];
	for (int axis = 0; axis < 3; axis++) span_lo[axis] = parent.bmin[axis], bin_len[axis] = (parent.bmax[axis] - parent.bmin[axis]) / NBINS;
	for (uint32_t i = begin; i < end; i++) {
		const Prim& p = in.prims[i];
		for (int axis = 0; axis < 3; axis++) {
			const float f = std::floor((p.centre(axis) - span_lo[axis]) / bin_len[axis]);
			int b;
			if (!(f >= 0.0f)) b = 0, bins.clamped++;
			else if (f > float(NBINS - 1)) b = NBINS - 1, bins.clamped++;
			else b = int(f);
			bins.n[axis][b]++;
			bins.box[axis][b].grow(p.box);
		}
	}
}

// One step of buildRecursive on `parent`: bins, the cheapest plane, the partition of the node's range.
// Bins are filled by `threads` threads on slices of the range when the range is long: box unions and
// counts do not depend on the order they are formed in.
void split_node(const BuildInput& in, const PtxBvhNode& parent, Workers* workers, PtxBvhNode& left, PtxBvhNode& right, Counters& c) {
	const unsigned threads = workers ? workers->size() : 1u;
	const uint32_t begin = parent.start, count = parent.size;
	Bins bins;
	if (threads > 1 && count >= (1u << 16)) {
		std::vector<Bins> part(threads);
		workers->run(threads, [&](unsigned k) {
			const uint64_t lo = begin + uint64_t(count) * k / threads, hi = begin + uint64_t(count) * (k + 1) / threads;
			bin_range(in, parent, uint32_t(lo), uint32_t(hi), part[k]);
		});
		for (const Bins& b : part) bins.merge(b);
	} else {
		bin_range(in, parent, begin, begin + count, bins);
	}
	c.clamped += bins.clamped;

	float best_cost = INFINITY;
	int best_axis = 0, best_plane = 0;
	Box best_l, best_r;
	uint32_t best_nl = 0, best_nr = 0;
	for (int axis = 0; axis < 3; axis++) {
		// Suffix unions once, prefix union on the fly: min/max are exact, so any order of
		// union gives the boxes the reference gets by re-enclosing bins per candidate.
		Box suffix[NBINS + 1];
		uint32_t suffix_n[NBINS + 1] = {};
		for (int b = NBINS - 1; b >= 0; b--) {
			suffix[b] = suffix[b + 1];
			suffix[b].grow(bins.box[axis][b]);
			suffix_n[b] = suffix_n[b + 1] + bins.n[axis][b];
		}
		Box prefix;
		uint32_t prefix_n = 0;
		for (int plane = 1; plane < NBINS; plane++) {
			prefix.grow(bins.box[axis][plane - 1]);
			prefix_n += bins.n[axis][plane - 1];
			const uint32_t nl = prefix_n, nr = suffix_n[plane];
			const float cost = prefix.area() * float(nl) + suffix[plane].area() * float(nr);
			if (cost < best_cost && nl > 0 && nr > 0) {
				best_cost = cost, best_axis = axis, best_plane = plane;
				best_l = prefix, best_r = suffix[plane], best_nl = nl, best_nr = nr;
			}
		}
	}

	// Partition [begin, begin+count) by the reference's predicate, exchanging elements in
	// the order libstdc++'s bidirectional std::partition does (first misplaced element
	// from the front with the last misplaced element from the back).
	const float span_lo = parent.bmin[best_axis], span_hi = parent.bmax[best_axis];
	const float bin_len = (span_hi - span_lo) / NBINS;
	const float threshold = span_lo + bin_len * float(uint32_t(best_plane));
	Prim* order = in.prims;
	auto goes_left = [&](const Prim& p) { return p.centre(best_axis) < threshold; };
	uint32_t front = begin, back = begin + count;
	if (threads > 1 && count >= (1u << 16)) {
		// The same exchanges, found without the two-pointer walk: with K elements satisfying the predicate,
		// that walk swaps the i-th non-satisfying element of the first K positions (counted from the front)
		// with the i-th satisfying element of the rest (counted from the back).
		std::vector<uint8_t> flag(count);
		std::vector<uint32_t> slice_n(threads + 1, 0u);
		auto slice = [&](uint32_t lo, uint32_t hi, unsigned k) { return lo + uint32_t(uint64_t(hi - lo) * k / threads); };
		workers->run(threads, [&](unsigned k) {
			uint32_t c = 0;
			for (uint32_t i = slice(0, count, k); i < slice(0, count, k + 1); i++) c += (flag[i] = goes_left(order[begin + i]) ? 1 : 0);
			slice_n[k + 1] = c;
		});
		uint32_t n_left = 0;
		for (unsigned k = 1; k <= threads; k++) n_left += slice_n[k];
		// misplaced elements of the front part (flag 0) and of the back part (flag 1), positions ascending
		std::vector<uint32_t> f_n(threads + 1, 0u), b_n(threads + 1, 0u);
		workers->run(threads, [&](unsigned k) {
			uint32_t c = 0;
			for (uint32_t i = slice(0, n_left, k); i < slice(0, n_left, k + 1); i++) c += !flag[i];
			f_n[k + 1] = c;
			c = 0;
			for (uint32_t i = slice(n_left, count, k); i < slice(n_left, count, k + 1); i++) c += flag[i];
			b_n[k + 1] = c;
		});
		for (unsigned k = 1; k <= threads; k++) f_n[k] += f_n[k - 1], b_n[k] += b_n[k - 1];
		const uint32_t m = f_n[threads];   // == b_n[threads]
		std::vector<uint32_t> f_pos(m), b_pos(m);
		workers->run(threads, [&](unsigned k) {
			uint32_t w = f_n[k];
			for (uint32_t i = slice(0, n_left, k); i < slice(0, n_left, k + 1); i++)
				if (!flag[i]) f_pos[w++] = begin + i;
			w = b_n[k];
			for (uint32_t i = slice(n_left, count, k); i < slice(n_left, count, k + 1); i++)
				if (flag[i]) b_pos[w++] = begin + i;
		});
		workers->run(threads, [&](unsigned k) {
			for (uint32_t i = slice(0, m, k); i < slice(0, m, k + 1); i++) std::swap(order[f_pos[i]], order[b_pos[m - 1 - i]]);
		});
		front = begin + n_left;
	} else {
		for (;;) {
			while (front != back && goes_left(order[front])) ++front;
			if (front == back) break;
			--back;
			while (front != back && !goes_left(order[back])) --back;
			if (front == back) break;
			std::swap(order[front], order[back]);
			++front;
		}
	}
	if (best_nl + best_nr == 0) c.dropped += count;
	else if (front - begin != best_nl) c.predicate_disagreements += (front - begin > best_nl) ? (front - begin - best_nl) : (best_nl - (front - begin));

	left = PtxBvhNode{}, right = PtxBvhNode{};
	store_box(left, best_l);
	left.start = begin, left.size = best_nl;
	store_box(right, best_r);
	right.start = begin + best_nl, right.size = best_nr;
	left.l = left.r = left.start;   // leaves until refined
	right.l = right.r = right.start;
}

// Refines nodes[0] depth-first, left subtree first; children are appended as consecutive nodes
// (bvh.cpp:223-241), so `nodes` ends up in the reference's creation order relative to its root.
void build_subtree(const BuildInput& in, std::vector<PtxBvhNode>& nodes, uint32_t root_depth, Counters& c) {
	struct Todo {
		uint32_t node, depth;
	};
	std::vector<Todo> todo{{0u, root_depth}};
	while (!todo.empty()) {
		const Todo job = todo.back();
		todo.pop_back();
		PtxBvhNode left, right;
		split_node(in, nodes[job.node], nullptr, left, right, c);
		const uint32_t li = uint32_t(nodes.size()), ri = li + 1;
		nodes.push_back(left);
		nodes.push_back(right);
		nodes[job.node].l = li, nodes[job.node].r = ri;
		c.max_depth = std::max(c.max_depth, job.depth + 1);
		// Depth-first, left subtree first: push right, then left.
		if (right.size > in.max_leaf_size) todo.push_back({ri, job.depth + 1});
		if (left.size > in.max_leaf_size) todo.push_back({li, job.depth + 1});
	}
}

}  // namespace

void build_bvh(const float* boxes, uint32_t n, uint32_t max_leaf_size, HostBvh& out) {
	out = HostBvh();
	out.order.resize(n);
	const unsigned threads = n >= (1u << 17) ? build_threads() : 1u;
	const bool profile = std::getenv("PTX_BUILD_PROFILE") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	auto lap = [&](const char* what) {
		const auto now = std::chrono::steady_clock::now();
		if (profile) std::fprintf(stderr, "[build] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
		t_last = now;
	};

	Workers workers(threads);
	std::vector<Prim> prims(n);
	std::vector<Box> slice_box(threads);
	workers.run(threads, [&](unsigned k) {
		const uint64_t lo = uint64_t(n) * k / threads, hi = uint64_t(n) * (k + 1) / threads;
		for (uint64_t i = lo; i < hi; i++) {
			const float* b = boxes + i * 6;
			for (int a = 0; a < 6; a++) prims[i].box[a] = b[a];
			prims[i].id = uint32_t(i), prims[i].pad = 0;
			slice_box[k].grow(b);
		}
	});
	Box scene;
	for (const Box& b : slice_box) scene.grow(b);
	lap("records");

	PtxBvhNode root{};
	store_box(root, scene);
	root.start = 0, root.size = n, root.l = root.r = 0;
	const BuildInput in{prims.data(), max_leaf_size};
	Counters total;
	if (threads <= 1 || n <= max_leaf_size) {
		out.nodes.push_back(root);
		if (n > max_leaf_size) build_subtree(in, out.nodes, 0, total);
	} else {
		// The top of the tree, node by node with the bins filled in parallel, down to subtrees small
		// enough to hand out whole; then those subtrees in parallel, each into a node array of its own
		// (their primitive ranges are disjoint); then one walk in the reference's creation order that
		// gives every node the index the one-thread build gives it.
		struct Top {
			PtxBvhNode node;
			int32_t left = -1, right = -1, task = -1;
			uint32_t depth = 0;
		};
		std::vector<Top> top(1);
		top[0].node = root;
		const uint32_t hand_out = std::max<uint32_t>(4096u, n / (threads * 8u));
		std::vector<uint32_t> tasks;
		std::vector<uint32_t> pending{0u};
		while (!pending.empty()) {
			const uint32_t x = pending.back();
			pending.pop_back();
			if (top[x].node.size <= max_leaf_size) continue;
			if (top[x].node.size <= hand_out) {
				top[x].task = int32_t(tasks.size());
				tasks.push_back(x);
				continue;
			}
			Top l, r;
			split_node(in, top[x].node, &workers, l.node, r.node, total);
			l.depth = r.depth = top[x].depth + 1;
			total.max_depth = std::max(total.max_depth, l.depth);
			top[x].left = int32_t(top.size()), top[x].right = int32_t(top.size() + 1);
			pending.push_back(uint32_t(top.size()));
			pending.push_back(uint32_t(top.size() + 1));
			top.push_back(l);
			top.push_back(r);
		}
		lap("top of the tree");
		std::vector<std::vector<PtxBvhNode>> sub(tasks.size());
		std::vector<Counters> sub_counters(tasks.size());
		std::vector<uint32_t> by_size(tasks.size());
		for (uint32_t k = 0; k < tasks.size(); k++) by_size[k] = k;
		std::sort(by_size.begin(), by_size.end(), [&](uint32_t a, uint32_t b) { return top[tasks[a]].node.size > top[tasks[b]].node.size; });
		workers.run(unsigned(tasks.size()), [&](unsigned j) {
			const uint32_t k = by_size[j];
			sub[k].push_back(top[tasks[k]].node);
			build_subtree(in, sub[k], top[tasks[k]].depth, sub_counters[k]);
		});
		for (const Counters& c : sub_counters) {
			total.clamped += c.clamped, total.predicate_disagreements += c.predicate_disagreements, total.dropped += c.dropped;
			total.max_depth = std::max(total.max_depth, c.max_depth);
		}
		lap("subtrees");
		size_t n_nodes = top.size();
		for (const auto& v : sub) n_nodes += v.size() - 1;
		out.nodes.resize(n_nodes);
		out.nodes[0] = root;
		size_t used = 1;
		struct Paste {
			uint32_t task, node, first;   // subtree, index of its root, index its first node below the root gets
		};
		std::vector<Paste> pastes;
		struct Visit {
			uint32_t top_index, node;
		};
		std::vector<Visit> walk{{0u, 0u}};
		while (!walk.empty()) {
			const Visit v = walk.back();
			walk.pop_back();
			const Top& t = top[v.top_index];
			if (t.task >= 0) {
				// the subtree's nodes below its root, in their own creation order, right here
				pastes.push_back({uint32_t(t.task), v.node, uint32_t(used)});
				used += sub[size_t(t.task)].size() - 1;
			} else if (t.left >= 0) {
				const uint32_t li = uint32_t(used);
				out.nodes[li] = top[size_t(t.left)].node;
				out.nodes[li + 1] = top[size_t(t.right)].node;
				used += 2;
				out.nodes[v.node].l = li, out.nodes[v.node].r = li + 1;
				walk.push_back({uint32_t(t.right), li + 1});
				walk.push_back({uint32_t(t.left), li});
			}
		}
		workers.run(unsigned(pastes.size()), [&](unsigned j) {
			const Paste& p = pastes[j];
			const std::vector<PtxBvhNode>& local = sub[p.task];
			const uint32_t shift = p.first - 1u;   // local index j >= 1 becomes shift + j
			out.nodes[p.node].l = local[0].l + shift, out.nodes[p.node].r = local[0].r + shift;
			for (size_t k = 1; k < local.size(); k++) {
				PtxBvhNode m = local[k];
				if (m.l != m.r) m.l += shift, m.r += shift;
				out.nodes[shift + k] = m;
			}
		});
	}
	lap("numbering");
	workers.run(threads, [&](unsigned k) {
		for (uint64_t i = uint64_t(n) * k / threads; i < uint64_t(n) * (k + 1) / threads; i++) out.order[i] = prims[i].id;
	});
	out.clamped = total.clamped, out.predicate_disagreements = total.predicate_disagreements, out.dropped = total.dropped;
	out.max_depth = total.max_depth;
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

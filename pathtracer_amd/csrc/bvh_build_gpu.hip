// bvh_build_gpu.hip -- the binned-SAH builder of bvh_build.cpp on the GPU: the same tree, node for node and
// primitive for primitive, as PT::BVH<P>::build (reference: src/pathtracer/bvh.cpp:31-243).
//
// The reference refines one node at a time, depth first.  Every step of that refinement depends only on the node's
// own primitive range, so here all nodes of one tree level are refined together:
//
//   bin      every primitive of a node that is to be split goes into one of 8 bins per axis (bvh.cpp:104-115, index
//            clamped to [0,7]); a bin is a box union and a count, formed with integer atomics on order-preserving
//            keys of the floats (min and max are exact, so any order of union gives the reference's boxes);
//            blocks whose 256 primitives all belong to one node gather their bins in LDS first
//   split    one thread per node: 7 candidate planes per axis, cost SA(L)*nL + SA(R)*nR, strictly lowest with both
//            sides non-empty wins, first found wins ties (bvh.cpp:117-160); the children's boxes and sizes come
//            from the bins (bvh.cpp:213-221)
//   move     the node's range is rearranged as libstdc++'s bidirectional std::partition rearranges it under the
//            predicate `centre < min + len * plane` (bvh.cpp:191-199): with K elements satisfying the predicate,
//            that walk exchanges the i-th non-satisfying element of the first K positions (from the front) with
//            the i-th satisfying element of the rest (from the back).  Flags, one prefix sum over the whole
//            array, two position tables, one exchange pass.
//
// The reference numbers nodes in creation order (children consecutive, the left subtree's nodes before the right
// child's, bvh.cpp:223-241).  A split node that is the k-th to be split in that depth-first order gets children
// 1+2k and 2+2k; k is the node's pre-order rank among split nodes, found from subtree counts (one pass up the
// levels, one pass down).
//
// Floating-point expressions are the host builder's, compiled with -ffp-contract=off.  Inputs whose boxes hold a
// negative zero or a NaN are handed to the host builder: there `std::min(+0, -0)` depends on which came first.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

#include "bvh_build.h"
#include "scene.h"

namespace ptx {
namespace {

constexpr int GB_NBINS = 8;
constexpr uint32_t GB_NONE = 0xFFFFFFFFu;

struct alignas(16) GPrim {   // a primitive as the builder moves it around: its box and the index it came with
	float box[6];
	uint32_t id, pad;
};

struct alignas(16) GNode {
	float bmin[3], bmax[3];
	uint32_t start, size;
	uint32_t child;        // left child (the right one follows it); GB_NONE: not split
	uint32_t bins;         // slot in the level's bin table while the node is being split
	uint32_t axis, nl;     // the split: axis, primitives the bins put on the left
	float threshold;
	uint32_t n_left;       // primitives the predicate puts on the left
	uint32_t internal;     // split nodes in this node's subtree, itself included
	uint32_t rank;         // pre-order rank among split nodes (valid for split nodes)
	uint32_t final_index;  // index in the reference's creation order
	uint32_t pad[3];
};

struct GBins {   // order-preserving keys of the box bounds, and counts
	uint32_t lo[3][GB_NBINS][3], hi[3][GB_NBINS][3], n[3][GB_NBINS];
};
constexpr int GB_BIN_WORDS = sizeof(GBins) / 4;

struct GCounters {
	unsigned long long clamped, disagreements, dropped;
	uint32_t n_nodes, n_active, bad_input, pad;
	uint32_t root_lo[3], root_hi[3];
};

// float <-> unsigned key with the same order (negative zero sorts just below positive zero)
__host__ __device__ inline uint32_t key_of(float f) {
	union {
		float f;
		uint32_t u;
	} v{f};
	return (v.u & 0x80000000u) ? ~v.u : (v.u | 0x80000000u);
}
__host__ __device__ inline float float_of(uint32_t k) {
	union {
		uint32_t u;
		float f;
	} v{(k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k};
	return v.f;
}

__device__ __forceinline__ float box_area(const float* lo, const float* hi) {   // BBox::surface_area, lib/bbox.h:55-59
	if (lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]) return 0.0f;
	const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
	return 2.0f * (ex * ez + ex * ey + ey * ez);
}

// ---- records, the root's box, input screening
__global__ void __launch_bounds__(256) k_gb_init(const float* __restrict__ boxes, uint32_t n, GPrim* __restrict__ prims, uint32_t* __restrict__ node_of, GCounters* c) {
	__shared__ uint32_t s_lo[3], s_hi[3];
	if (threadIdx.x < 3) s_lo[threadIdx.x] = 0xFFFFFFFFu, s_hi[threadIdx.x] = 0u;
	__syncthreads();
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) {
		GPrim p;
		bool bad = false;
		for (int a = 0; a < 6; a++) {
			p.box[a] = boxes[size_t(i) * 6 + a];
			bad |= (__float_as_uint(p.box[a]) == 0x80000000u) || (p.box[a] != p.box[a]);
		}
		p.id = i, p.pad = 0;
		prims[i] = p;
		node_of[i] = 0;
		for (int a = 0; a < 3; a++) atomicMin(&s_lo[a], key_of(p.box[a])), atomicMax(&s_hi[a], key_of(p.box[3 + a]));
		if (bad) atomicOr(&c->bad_input, 1u);
	}
	__syncthreads();
	if (threadIdx.x < 3) atomicMin(&c->root_lo[threadIdx.x], s_lo[threadIdx.x]), atomicMax(&c->root_hi[threadIdx.x], s_hi[threadIdx.x]);
}

__global__ void k_gb_root(GNode* nodes, uint32_t n, GCounters* c) {
	GNode r{};
	for (int a = 0; a < 3; a++) r.bmin[a] = float_of(c->root_lo[a]), r.bmax[a] = float_of(c->root_hi[a]);
	r.start = 0, r.size = n, r.child = GB_NONE, r.bins = GB_NONE;
	nodes[0] = r;
	c->n_nodes = 1;
}

// ---- per level: nodes of [level_begin, level_end) holding more than max_leaf primitives get a bin slot
__global__ void __launch_bounds__(256) k_gb_prepare(GNode* __restrict__ nodes, uint32_t level_begin, uint32_t level_end, uint32_t max_leaf, GBins* __restrict__ bins, GCounters* c) {
	const uint32_t x = level_begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= level_end) return;
	if (nodes[x].size <= max_leaf) return;
	const uint32_t slot = atomicAdd(&c->n_active, 1u);
	nodes[x].bins = slot;
	GBins& b = bins[slot];
	const uint32_t empty_lo = key_of(FLT_MAX), empty_hi = key_of(-FLT_MAX);
	for (int a = 0; a < 3; a++)
		for (int k = 0; k < GB_NBINS; k++) {
			for (int d = 0; d < 3; d++) b.lo[a][k][d] = empty_lo, b.hi[a][k][d] = empty_hi;
			b.n[a][k] = 0;
		}
}

__global__ void __launch_bounds__(256) k_gb_bin(const GPrim* __restrict__ prims, const uint32_t* __restrict__ node_of, uint32_t n, const GNode* __restrict__ nodes, uint32_t level_begin,
                                                GBins* __restrict__ bins, GCounters* c) {
	__shared__ GBins s_bins;
	__shared__ uint32_t s_slot;
	const uint32_t first = blockIdx.x * blockDim.x, last = min(first + blockDim.x, n) - 1u;
	const uint32_t i = first + threadIdx.x;
	// one node for the whole block (a node's primitives are contiguous): gather in LDS, flush once
	const uint32_t x_first = node_of[first], x_last = node_of[last];
	const bool block_local = x_first == x_last;
	if (block_local) {
		if (x_first < level_begin || nodes[x_first].bins == GB_NONE) return;   // nothing to split here
		uint32_t* w = reinterpret_cast<uint32_t*>(&s_bins);
		const uint32_t empty_lo = key_of(FLT_MAX), empty_hi = key_of(-FLT_MAX);
		for (int k = threadIdx.x; k < GB_BIN_WORDS; k += blockDim.x) w[k] = k < 72 ? empty_lo : (k < 144 ? empty_hi : 0u);
		if (threadIdx.x == 0) s_slot = nodes[x_first].bins;
		__syncthreads();
	}
	uint32_t clamped = 0;
	if (i < n) {
		const uint32_t x = node_of[i];
		const GNode& node = nodes[x];
		if (x >= level_begin && node.bins != GB_NONE) {
			GBins& b = block_local ? s_bins : bins[node.bins];
			const GPrim p = prims[i];
			for (int axis = 0; axis < 3; axis++) {
				const float span_lo = node.bmin[axis], bin_len = (node.bmax[axis] - node.bmin[axis]) / GB_NBINS;
				const float centre = (p.box[axis] + p.box[3 + axis]) * 0.5f;   // BBox::center, lib/bbox.h:45-47
				const float f = floorf((centre - span_lo) / bin_len);
				int k;
				if (!(f >= 0.0f)) k = 0, clamped++;
				else if (f > float(GB_NBINS - 1)) k = GB_NBINS - 1, clamped++;
				else k = int(f);
				atomicAdd(&b.n[axis][k], 1u);
				for (int d = 0; d < 3; d++) atomicMin(&b.lo[axis][k][d], key_of(p.box[d])), atomicMax(&b.hi[axis][k][d], key_of(p.box[3 + d]));
			}
		}
	}
	if (clamped) atomicAdd(&c->clamped, (unsigned long long)clamped);
	if (block_local) {
		__syncthreads();
		GBins& g = bins[s_slot];
		const uint32_t* w = reinterpret_cast<const uint32_t*>(&s_bins);
		uint32_t* gw = reinterpret_cast<uint32_t*>(&g);
		for (int k = threadIdx.x; k < GB_BIN_WORDS; k += blockDim.x) {
			if (k < 72) atomicMin(&gw[k], w[k]);
			else if (k < 144) atomicMax(&gw[k], w[k]);
			else if (w[k]) atomicAdd(&gw[k], w[k]);
		}
	}
}

// One step of buildRecursive per node that has a bin slot: the cheapest plane and the two children.
__global__ void __launch_bounds__(64) k_gb_split(GNode* __restrict__ nodes, uint32_t level_begin, uint32_t level_end, const GBins* __restrict__ bins, GCounters* c) {
	const uint32_t x = level_begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= level_end) return;
	GNode node = nodes[x];
	if (node.bins == GB_NONE) return;
	const GBins& b = bins[node.bins];
	float best_cost = INFINITY;
	int best_axis = 0, best_plane = 0;
	float best_l[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX}, best_r[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
	uint32_t best_nl = 0, best_nr = 0;
	for (int axis = 0; axis < 3; axis++) {
		// suffix unions once, prefix union on the fly
		float suffix[GB_NBINS + 1][6];
		uint32_t suffix_n[GB_NBINS + 1];
		for (int d = 0; d < 3; d++) suffix[GB_NBINS][d] = FLT_MAX, suffix[GB_NBINS][3 + d] = -FLT_MAX;
		suffix_n[GB_NBINS] = 0;
		for (int k = GB_NBINS - 1; k >= 0; k--) {
			for (int d = 0; d < 3; d++) {
				suffix[k][d] = fminf(suffix[k + 1][d], float_of(b.lo[axis][k][d]));
				suffix[k][3 + d] = fmaxf(suffix[k + 1][3 + d], float_of(b.hi[axis][k][d]));
			}
			suffix_n[k] = suffix_n[k + 1] + b.n[axis][k];
		}
		float prefix[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
		uint32_t prefix_n = 0;
		for (int plane = 1; plane < GB_NBINS; plane++) {
			for (int d = 0; d < 3; d++) {
				prefix[d] = fminf(prefix[d], float_of(b.lo[axis][plane - 1][d]));
				prefix[3 + d] = fmaxf(prefix[3 + d], float_of(b.hi[axis][plane - 1][d]));
			}
			prefix_n += b.n[axis][plane - 1];
			const uint32_t nl = prefix_n, nr = suffix_n[plane];
			const float cost = box_area(prefix, prefix + 3) * float(nl) + box_area(suffix[plane], suffix[plane] + 3) * float(nr);
			if (cost < best_cost && nl > 0 && nr > 0) {
				best_cost = cost, best_axis = axis, best_plane = plane, best_nl = nl, best_nr = nr;
				for (int d = 0; d < 6; d++) best_l[d] = prefix[d], best_r[d] = suffix[plane][d];
			}
		}
	}
	const float span_lo = node.bmin[best_axis], span_hi = node.bmax[best_axis];
	const float bin_len = (span_hi - span_lo) / GB_NBINS;
	node.axis = uint32_t(best_axis), node.nl = best_nl;
	node.threshold = span_lo + bin_len * float(uint32_t(best_plane));
	node.child = atomicAdd(&c->n_nodes, 2u);
	nodes[x] = node;
	GNode l{}, r{};
	for (int d = 0; d < 3; d++) l.bmin[d] = best_l[d], l.bmax[d] = best_l[3 + d], r.bmin[d] = best_r[d], r.bmax[d] = best_r[3 + d];
	l.start = node.start, l.size = best_nl, r.start = node.start + best_nl, r.size = best_nr;
	l.child = r.child = GB_NONE, l.bins = r.bins = GB_NONE;
	nodes[node.child] = l, nodes[node.child + 1] = r;
}

// flag[i] = the predicate of the partition, for primitives of nodes split at this level (0 elsewhere)
__global__ void __launch_bounds__(256) k_gb_flags(const GPrim* __restrict__ prims, const uint32_t* __restrict__ node_of, uint32_t n, const GNode* __restrict__ nodes, uint32_t level_begin,
                                                  uint32_t* __restrict__ flag) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t x = node_of[i];
	uint32_t f = 0;
	if (x >= level_begin && nodes[x].child != GB_NONE) {
		const GNode& node = nodes[x];
		const float* box = prims[i].box;
		f = ((box[node.axis] + box[3 + node.axis]) * 0.5f < node.threshold) ? 1u : 0u;
	}
	flag[i] = f;
}

// ---- exclusive prefix sum of `in` (n entries) into `out` (n + 1 entries: out[n] is the total)
constexpr uint32_t SCAN_ITEMS = 8, SCAN_TILE = 256 * SCAN_ITEMS;
__global__ void __launch_bounds__(256) k_gb_scan_local(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out, uint32_t* __restrict__ totals) {
	__shared__ uint32_t s[256];
	const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
	uint32_t v[SCAN_ITEMS], sum = 0;
	for (uint32_t k = 0; k < SCAN_ITEMS; k++) v[k] = base + k < n ? in[base + k] : 0u, sum += v[k];
	s[threadIdx.x] = sum;
	__syncthreads();
	for (uint32_t d = 1; d < 256; d <<= 1) {
		const uint32_t t = threadIdx.x >= d ? s[threadIdx.x - d] : 0u;
		__syncthreads();
		s[threadIdx.x] += t;
		__syncthreads();
	}
	uint32_t run = s[threadIdx.x] - sum;
	for (uint32_t k = 0; k < SCAN_ITEMS; k++) {
		if (base + k < n) out[base + k] = run;
		run += v[k];
	}
	if (threadIdx.x == 255) totals[blockIdx.x] = s[255];
}
__global__ void __launch_bounds__(256) k_gb_scan_totals(uint32_t* __restrict__ totals, uint32_t count, uint32_t* __restrict__ grand_total) {
	__shared__ uint32_t s[256];
	uint32_t carry = 0;
	for (uint32_t base = 0; base < count; base += 256) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t v = i < count ? totals[i] : 0u;
		s[threadIdx.x] = v;
		__syncthreads();
		for (uint32_t d = 1; d < 256; d <<= 1) {
			const uint32_t t = threadIdx.x >= d ? s[threadIdx.x - d] : 0u;
			__syncthreads();
			s[threadIdx.x] += t;
			__syncthreads();
		}
		if (i < count) totals[i] = carry + s[threadIdx.x] - v;
		carry += s[255];
		__syncthreads();
	}
	if (threadIdx.x == 0) *grand_total = carry;
}
__global__ void __launch_bounds__(256) k_gb_scan_add(uint32_t* __restrict__ out, uint32_t n, const uint32_t* __restrict__ totals) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] += totals[i / SCAN_TILE];
}

// per split node: how many of its primitives satisfy the predicate; statistics
__global__ void __launch_bounds__(256) k_gb_count(GNode* __restrict__ nodes, uint32_t level_begin, uint32_t level_end, const uint32_t* __restrict__ excl, GCounters* c) {
	const uint32_t x = level_begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= level_end) return;
	GNode& node = nodes[x];
	if (node.child == GB_NONE) return;
	const uint32_t n_left = excl[node.start + node.size] - excl[node.start];
	node.n_left = n_left;
	const uint32_t nl = node.nl, nr = nodes[node.child + 1].size;
	if (nl + nr == 0) atomicAdd(&c->dropped, (unsigned long long)node.size);
	else if (n_left != nl) atomicAdd(&c->disagreements, (unsigned long long)(n_left > nl ? n_left - nl : nl - n_left));
}

// Positions of the satisfying elements behind the first n_left positions of the node's range, in ascending position
// (their partners, the non-satisfying ones among the first n_left, find their rank from the prefix sums themselves).
__global__ void __launch_bounds__(256) k_gb_positions(const uint32_t* __restrict__ node_of, uint32_t n, const GNode* __restrict__ nodes, uint32_t level_begin, const uint32_t* __restrict__ flag,
                                                      const uint32_t* __restrict__ excl, uint32_t* __restrict__ bpos) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t x = node_of[i];
	if (x < level_begin || nodes[x].child == GB_NONE) return;
	const GNode& node = nodes[x];
	const uint32_t begin = node.start, j = i - begin, f = flag[i];
	if (j >= node.n_left && f) bpos[begin + (excl[i] - excl[begin + node.n_left])] = i;
}

// The exchanges, and which child every position of the range now belongs to.
__global__ void __launch_bounds__(256) k_gb_exchange(GPrim* __restrict__ prims, uint32_t* __restrict__ node_of, uint32_t n, const GNode* __restrict__ nodes, uint32_t level_begin,
                                                     const uint32_t* __restrict__ flag, const uint32_t* __restrict__ excl, const uint32_t* __restrict__ bpos) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t x = node_of[i];
	if (x < level_begin || nodes[x].child == GB_NONE) return;
	const GNode& node = nodes[x];
	const uint32_t begin = node.start, j = i - begin;
	if (j < node.n_left && !flag[i]) {
		const uint32_t r = j - (excl[i] - excl[begin]);
		const uint32_t m = node.n_left - (excl[begin + node.n_left] - excl[begin]);
		const uint32_t partner = bpos[begin + (m - 1u - r)];
		const GPrim a = prims[i], b = prims[partner];
		prims[i] = b, prims[partner] = a;
	}
	node_of[i] = j < node.nl ? node.child : node.child + 1u;
}

// ---- numbering in the reference's creation order
__global__ void __launch_bounds__(256) k_gb_internal(GNode* __restrict__ nodes, uint32_t level_begin, uint32_t level_end) {
	const uint32_t x = level_begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= level_end) return;
	GNode& node = nodes[x];
	node.internal = node.child == GB_NONE ? 0u : 1u + nodes[node.child].internal + nodes[node.child + 1].internal;
}
__global__ void __launch_bounds__(256) k_gb_number(GNode* __restrict__ nodes, uint32_t level_begin, uint32_t level_end) {
	const uint32_t x = level_begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= level_end) return;
	const GNode& node = nodes[x];
	if (node.child == GB_NONE) return;
	const uint32_t k = node.rank;
	GNode &l = nodes[node.child], &r = nodes[node.child + 1];
	l.final_index = 1u + 2u * k, r.final_index = 2u + 2u * k;
	l.rank = k + 1u, r.rank = k + 1u + l.internal;
}
__global__ void __launch_bounds__(256) k_gb_emit(const GNode* __restrict__ nodes, uint32_t n_nodes, PtxBvhNode* __restrict__ out) {
	const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= n_nodes) return;
	const GNode& node = nodes[x];
	PtxBvhNode o{};
	for (int d = 0; d < 3; d++) o.bmin[d] = node.bmin[d], o.bmax[d] = node.bmax[d];
	o.start = node.start, o.size = node.size;
	if (node.child == GB_NONE) o.l = o.r = node.start;   // leaves carry l == r (bvh.cpp:63)
	else o.l = nodes[node.child].final_index, o.r = nodes[node.child + 1].final_index;
	out[node.final_index] = o;
}
__global__ void __launch_bounds__(256) k_gb_order(const GPrim* __restrict__ prims, uint32_t n, uint32_t* __restrict__ order) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) order[i] = prims[i].id;
}

struct DeviceBuffers {
	std::vector<void*> all;
	~DeviceBuffers() {
		for (void* p : all) (void)hipFree(p);
	}
	void release(void* p) {
		if (!p) return;
		all.erase(std::remove(all.begin(), all.end(), p), all.end());
		(void)hipFree(p);
	}
	template<typename T>
	hipError_t get(T** out, size_t count) {
		void* p = nullptr;
		const hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
		if (e == hipSuccess) all.push_back(p);
		*out = static_cast<T*>(p);
		return e;
	}
};

inline uint32_t blocks_for(uint64_t n, uint32_t block = 256) { return uint32_t(std::max<uint64_t>(1, (n + block - 1) / block)); }

}  // namespace

// PT::BVH<P>::build over n boxes (6 floats each) on `device`; `used_gpu` tells whether the device built the tree
// (false: the input went to the host builder -- empty, or with a negative zero / NaN among the box bounds).
int build_bvh_gpu(int device, const float* boxes, uint32_t n, uint32_t max_leaf_size, HostBvh& out, bool& used_gpu) {
	used_gpu = false;
	if (n == 0) {
		build_bvh(boxes, n, max_leaf_size, out);
		return PTX_OK;
	}
	if (n >= (1u << 31)) return fail(PTX_ERR_UNSUPPORTED, "build_bvh_gpu: more than 2^31 primitives");
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(PTX_ERR_NO_DEVICE, "device index out of range");
	PTX_HIP(hipSetDevice(device));
	DeviceBuffers mem;
	float* d_boxes;
	GPrim* prims;
	uint32_t *node_of, *flag, *excl, *bpos, *totals;
	GNode* nodes;
	GCounters* counters;
	const uint64_t max_nodes = 2ull * n + 2;
	PTX_HIP(mem.get(&d_boxes, size_t(n) * 6));
	PTX_HIP(mem.get(&prims, n));
	PTX_HIP(mem.get(&node_of, n));
	PTX_HIP(mem.get(&flag, n));
	PTX_HIP(mem.get(&excl, size_t(n) + 1));
	PTX_HIP(mem.get(&bpos, n));
	PTX_HIP(mem.get(&totals, blocks_for(n, SCAN_TILE)));
	PTX_HIP(mem.get(&nodes, max_nodes));
	PTX_HIP(mem.get(&counters, 1));
	PTX_HIP(hipMemcpy(d_boxes, boxes, size_t(n) * 6 * sizeof(float), hipMemcpyHostToDevice));
	GCounters init{};
	for (int a = 0; a < 3; a++) init.root_lo[a] = 0xFFFFFFFFu, init.root_hi[a] = 0u;
	PTX_HIP(hipMemcpy(counters, &init, sizeof(init), hipMemcpyHostToDevice));
	hipStream_t st = nullptr;
	const dim3 block(256), grid_n(blocks_for(n));
	hipLaunchKernelGGL(k_gb_init, grid_n, block, 0, st, d_boxes, n, prims, node_of, counters);
	hipLaunchKernelGGL(k_gb_root, dim3(1), dim3(1), 0, st, nodes, n, counters);
	GCounters h{};
	PTX_HIP(hipMemcpy(&h, counters, sizeof(h), hipMemcpyDeviceToHost));
	if (h.bad_input) {
		build_bvh(boxes, n, max_leaf_size, out);
		return PTX_OK;
	}

	std::vector<uint32_t> level_begin{0u};
	uint32_t begin = 0, end = 1;
	GBins* bins = nullptr;
	size_t bins_capacity = 0;
	for (;;) {
		const uint32_t count = end - begin;
		// One bin table (672 B) per node that will be split.  Such a node holds more than max_leaf primitives and the
		// nodes of a level hold disjoint ranges, so a level never needs more than n / (max_leaf + 1) tables however many
		// nodes it has; an outgrown array is freed before its successor is allocated.
		const size_t need = std::min<size_t>(count, size_t(n) / (size_t(max_leaf_size) + 1) + 1);
		if (bins_capacity < need) {
			mem.release(bins);
			bins = nullptr;
			bins_capacity = std::min<size_t>(std::max<size_t>(need, bins_capacity * 2), size_t(n) / (size_t(max_leaf_size) + 1) + 1);
			PTX_HIP(mem.get(&bins, bins_capacity));
		}
		PTX_HIP(hipMemsetAsync(&counters->n_active, 0, sizeof(uint32_t), st));
		const dim3 grid_level(blocks_for(count));
		hipLaunchKernelGGL(k_gb_prepare, grid_level, block, 0, st, nodes, begin, end, max_leaf_size, bins, counters);
		hipLaunchKernelGGL(k_gb_bin, grid_n, block, 0, st, prims, node_of, n, nodes, begin, bins, counters);
		hipLaunchKernelGGL(k_gb_split, dim3(blocks_for(count, 64)), dim3(64), 0, st, nodes, begin, end, bins, counters);
		PTX_HIP(hipMemcpy(&h, counters, sizeof(h), hipMemcpyDeviceToHost));
		if (h.n_nodes == end) break;   // nothing was split: the tree is complete
		if (h.n_nodes > max_nodes) return fail(PTX_ERR_DEVICE, "build_bvh_gpu: node count out of bounds");
		hipLaunchKernelGGL(k_gb_flags, grid_n, block, 0, st, prims, node_of, n, nodes, begin, flag);
		const uint32_t tiles = blocks_for(n, SCAN_TILE);
		hipLaunchKernelGGL(k_gb_scan_local, dim3(tiles), block, 0, st, flag, n, excl, totals);
		hipLaunchKernelGGL(k_gb_scan_totals, dim3(1), block, 0, st, totals, tiles, excl + n);
		hipLaunchKernelGGL(k_gb_scan_add, grid_n, block, 0, st, excl, n, totals);
		hipLaunchKernelGGL(k_gb_count, grid_level, block, 0, st, nodes, begin, end, excl, counters);
		hipLaunchKernelGGL(k_gb_positions, grid_n, block, 0, st, node_of, n, nodes, begin, flag, excl, bpos);
		hipLaunchKernelGGL(k_gb_exchange, grid_n, block, 0, st, prims, node_of, n, nodes, begin, flag, excl, bpos);
		begin = end, end = h.n_nodes;
		level_begin.push_back(begin);
	}
	const uint32_t n_nodes = end;
	level_begin.push_back(end);
	const size_t n_levels = level_begin.size() - 1;
	for (size_t l = n_levels; l-- > 0;)
		hipLaunchKernelGGL(k_gb_internal, dim3(blocks_for(level_begin[l + 1] - level_begin[l])), block, 0, st, nodes, level_begin[l], level_begin[l + 1]);
	for (size_t l = 0; l < n_levels; l++)   // the root has rank 0 and index 0 (its record was zero-initialised)
		hipLaunchKernelGGL(k_gb_number, dim3(blocks_for(level_begin[l + 1] - level_begin[l])), block, 0, st, nodes, level_begin[l], level_begin[l + 1]);
	PtxBvhNode* d_out;
	uint32_t* d_order;
	PTX_HIP(mem.get(&d_out, n_nodes));
	PTX_HIP(mem.get(&d_order, n));
	hipLaunchKernelGGL(k_gb_emit, dim3(blocks_for(n_nodes)), block, 0, st, nodes, n_nodes, d_out);
	hipLaunchKernelGGL(k_gb_order, grid_n, block, 0, st, prims, n, d_order);
	PTX_HIP(hipGetLastError());
	out = HostBvh();
	out.nodes.resize(n_nodes);
	out.order.resize(n);
	PTX_HIP(hipMemcpy(out.nodes.data(), d_out, size_t(n_nodes) * sizeof(PtxBvhNode), hipMemcpyDeviceToHost));
	PTX_HIP(hipMemcpy(out.order.data(), d_order, size_t(n) * sizeof(uint32_t), hipMemcpyDeviceToHost));
	PTX_HIP(hipMemcpy(&h, counters, sizeof(h), hipMemcpyDeviceToHost));
	out.clamped = h.clamped, out.predicate_disagreements = h.disagreements, out.dropped = h.dropped;
	out.max_depth = uint32_t(n_levels - 1);
	used_gpu = true;
	return PTX_OK;
}

}  // namespace ptx

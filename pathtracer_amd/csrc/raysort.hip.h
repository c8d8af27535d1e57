// raysort.hip.h -- bucket sort of a ray batch for coherence.
//
// A wave advances only when the slowest of its 64 node fetches has come back, so with
// incoherent rays every traversal step pays the latency of the farthest memory tier any
// lane touched.  Rays that start in the same part of the scene and point into the same
// octant walk the same upper part of the tree: their node fetches collapse into a few
// cache lines and stay in L1/L2.  This pass groups the batch that way before tracing:
//   key = Morton code of the origin's cell on a 32^3 grid over the tree's root box (15 bits)
//         followed by the direction's sign octant (3 bits)
// and a counting sort (histogram, exclusive scan, scatter) produces `order`, the ray indices
// grouped by key.  Only the order in which rays are picked up changes; every ray's result is
// computed from that ray alone and written to its own slot, so results are unaffected.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/ptx.h"

namespace ptx {

#ifndef PTX_SORT_GRID_BITS
#define PTX_SORT_GRID_BITS 5
#endif
constexpr uint32_t SORT_GRID_BITS = PTX_SORT_GRID_BITS;                   // 32 cells per axis
constexpr uint32_t SORT_BUCKETS = 1u << (3 * SORT_GRID_BITS + 3);         // 262144
constexpr uint32_t SCAN_THREADS = 1024;

__device__ __forceinline__ uint32_t spread3(uint32_t v) {   // 5 bits -> every third bit
	v &= 0x1Fu;
	v = (v | (v << 8)) & 0x100Fu;
	v = (v | (v << 4)) & 0x10C3u;
	v = (v | (v << 2)) & 0x1249u;
	return v;
}

__device__ __forceinline__ uint32_t ray_sort_key(const PtxRay& r, float3 lo, float3 scale) {
	const float cells = float(1u << SORT_GRID_BITS) - 1.0f;
	// fminf/fmaxf drop NaNs, so any origin lands in some cell
	const uint32_t cx = uint32_t(fminf(fmaxf((r.o[0] - lo.x) * scale.x, 0.0f), cells));
	const uint32_t cy = uint32_t(fminf(fmaxf((r.o[1] - lo.y) * scale.y, 0.0f), cells));
	const uint32_t cz = uint32_t(fminf(fmaxf((r.o[2] - lo.z) * scale.z, 0.0f), cells));
	const uint32_t octant = (r.d[0] < 0.0f ? 1u : 0u) | (r.d[1] < 0.0f ? 2u : 0u) | (r.d[2] < 0.0f ? 4u : 0u);
	return ((spread3(cx) | (spread3(cy) << 1) | (spread3(cz) << 2)) << 3) | octant;
}

__global__ void __launch_bounds__(256) k_sort_histogram(const PtxRay* __restrict__ rays, uint64_t n, float3 lo, float3 scale,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ hist) {
	const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float4* rp = reinterpret_cast<const float4*>(rays + i);
	const float4 a = rp[0], b = rp[1];
	PtxRay r{{a.x, a.y, a.z}, {a.w, b.x, b.y}, b.z, b.w};
	const uint32_t k = ray_sort_key(r, lo, scale);
	keys[i] = k;
	atomicAdd(&hist[k], 1u);
}

// Exclusive scan of SORT_BUCKETS counters in place, one block.
__global__ void __launch_bounds__(SCAN_THREADS) k_sort_scan(uint32_t* __restrict__ hist) {
	__shared__ uint32_t partial[SCAN_THREADS];
	constexpr uint32_t PER = SORT_BUCKETS / SCAN_THREADS;
	const uint32_t t = threadIdx.x;
	uint32_t sum = 0;
	for (uint32_t k = 0; k < PER; k++) sum += hist[t * PER + k];
	partial[t] = sum;
	__syncthreads();
	for (uint32_t step = 1; step < SCAN_THREADS; step <<= 1) {   // Hillis-Steele inclusive scan
		const uint32_t add = t >= step ? partial[t - step] : 0u;
		__syncthreads();
		partial[t] += add;
		__syncthreads();
	}
	uint32_t running = partial[t] - sum;
	for (uint32_t k = 0; k < PER; k++) {
		const uint32_t c = hist[t * PER + k];
		hist[t * PER + k] = running;
		running += c;
	}
}

__global__ void __launch_bounds__(256) k_sort_scatter(const uint32_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ offsets,
                                                      uint32_t* __restrict__ order) {
	const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t pos = atomicAdd(&offsets[keys[i]], 1u);
	order[pos] = uint32_t(i);
}

}  // namespace ptx

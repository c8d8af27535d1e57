// raysort.hip.h -- grouping of a ray batch for coherence.
//
// A wave advances only when the slowest of its 64 node fetches has come back, so with
// incoherent rays every traversal step pays the latency of the farthest memory tier any
// lane touched.  Rays that start in the same part of the scene and point into the same
// octant walk the same upper part of the tree: their node fetches collapse into a few
// cache lines and stay in L1/L2.  This pass groups the batch that way before tracing:
//   key = Morton code of the origin's cell on a 32^3 grid over the tree's root box (15 bits)
//         followed by the direction's sign octant (3 bits), with a 2-bit cost class slipped in
//         below the Morton code's top level (see ray_cost_class)
// and two radix passes of 10 bits each (low digit, then high digit) produce `order`, the ray
// indices grouped by key.  Each pass is count -> scan -> scatter over tiles of SORT_TILE rays:
// a block counts its tile's digits in an LDS histogram, a three-kernel scan turns the digit-major
// table of all tiles' counts into offsets, and the scatter ranks every ray with an LDS atomic on its digit's
// running offset.  The LDS rank is not stable inside a tile, so rays of one tile that share the
// high digit can leave the second pass in another order than they entered it: the result is a
// permutation grouped by the high digit and, within it, by the low digit up to a shuffle among
// a tile's few dozen neighbours -- as good for coherence, and no atomics leave the CU.
// Only the order in which rays are picked up changes; every ray's result is computed from
// that ray alone and written to its own slot, so results are unaffected.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/ptx.h"

namespace ptx {

#ifndef PTX_SORT_GRID_BITS
#define PTX_SORT_GRID_BITS 5
#endif
constexpr uint32_t SORT_GRID_BITS = PTX_SORT_GRID_BITS;    // 32 cells per axis
#ifndef PTX_SORT_CLASS_BITS
#define PTX_SORT_CLASS_BITS 2
#endif
constexpr uint32_t SORT_CLASS_BITS = PTX_SORT_CLASS_BITS;
constexpr uint32_t SORT_KEY_BITS = 3 * SORT_GRID_BITS + 3 + SORT_CLASS_BITS; // 20
constexpr uint32_t SORT_DIGIT_BITS = (SORT_KEY_BITS + 1) / 2, SORT_DIGITS = 1u << SORT_DIGIT_BITS;
constexpr uint32_t SORT_TILE = 16384;                      // rays per block and pass
static_assert(SORT_KEY_BITS <= 2 * SORT_DIGIT_BITS, "two passes cover the key");

__device__ __forceinline__ uint32_t spread3(uint32_t v) {   // up to 8 bits -> every third bit
	v &= 0xFFu;
	v = (v | (v << 8)) & 0x00F00Fu;
	v = (v | (v << 4)) & 0x0C30C3u;
	v = (v | (v << 2)) & 0x249249u;
	return v;
}

// What a ray costs is close to proportional to the distance it travels inside the root box before its
// first hit (correlation 0.97 on a triangle soup), and the chord [entry, exit] clipped to [tmin, tmax]
// bounds that distance before anything is traced.  The replenishing kernel's last rays decide how long
// it runs half empty after its queue has run dry, so rays with a short chord are queued last within
// each eighth of the order: class 0 (chord >= t2) first, then [t1, t2), then (0, t1), then the rays
// that miss the box.  Only the pick-up order changes.
__device__ __forceinline__ uint32_t ray_cost_class(const PtxRay& r, float3 lo, float3 hi, float t1, float t2) {
	// x/0 is +-inf and fminf/fmaxf drop the NaN of 0/0, so a ray that does not move along an axis is
	// either unconstrained by that slab or misses the box
	const float ax = (lo.x - r.o[0]) / r.d[0], bx = (hi.x - r.o[0]) / r.d[0];
	const float ay = (lo.y - r.o[1]) / r.d[1], by = (hi.y - r.o[1]) / r.d[1];
	const float az = (lo.z - r.o[2]) / r.d[2], bz = (hi.z - r.o[2]) / r.d[2];
	const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), r.tmin));
	const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), r.tmax));
	const float chord = t_out - t_in;
	return chord >= t2 ? 0u : chord >= t1 ? 1u : chord > 0.0f ? 2u : 3u;   // a NaN chord fails every comparison
}

__device__ __forceinline__ uint32_t ray_sort_key(const PtxRay& r, float3 lo, float3 scale, float3 hi, float t1, float t2) {
	const float cells = float(1u << SORT_GRID_BITS) - 1.0f;
	// fminf/fmaxf drop NaNs, so any origin lands in some cell
	const uint32_t cx = uint32_t(fminf(fmaxf((r.o[0] - lo.x) * scale.x, 0.0f), cells));
	const uint32_t cy = uint32_t(fminf(fmaxf((r.o[1] - lo.y) * scale.y, 0.0f), cells));
	const uint32_t cz = uint32_t(fminf(fmaxf((r.o[2] - lo.z) * scale.z, 0.0f), cells));
	const uint32_t octant = (r.d[0] < 0.0f ? 1u : 0u) | (r.d[1] < 0.0f ? 2u : 0u) | (r.d[2] < 0.0f ? 4u : 0u);
	const uint32_t morton = spread3(cx) | (spread3(cy) << 1) | (spread3(cz) << 2);
	if (SORT_CLASS_BITS == 0) return (morton << 3) | octant;
	constexpr uint32_t LOW_BITS = 3 * SORT_GRID_BITS - 3;   // Morton bits below the top level
	const uint32_t cls = ray_cost_class(r, lo, hi, t1, t2) >> (2 - SORT_CLASS_BITS);
	return (((((morton >> LOW_BITS) << SORT_CLASS_BITS) | cls) << LOW_BITS | (morton & ((1u << LOW_BITS) - 1u))) << 3) | octant;
}

// Pass 0 count: keys from the rays (stored for the later kernels) and the tile's low-digit histogram.
__global__ void __launch_bounds__(256) k_sort_count_rays(const PtxRay* __restrict__ rays, uint64_t n, float3 lo, float3 scale, float3 hi, float t1, float t2,
                                                         uint32_t* __restrict__ keys, uint32_t* __restrict__ table, uint32_t n_tiles) {
	__shared__ uint32_t hist[SORT_DIGITS];
	for (uint32_t d = threadIdx.x; d < SORT_DIGITS; d += blockDim.x) hist[d] = 0;
	__syncthreads();
	const uint64_t begin = uint64_t(blockIdx.x) * SORT_TILE, end = begin + SORT_TILE < n ? begin + SORT_TILE : n;
	for (uint64_t i = begin + threadIdx.x; i < end; i += blockDim.x) {
		const float4* rp = reinterpret_cast<const float4*>(rays + i);
		const float4 a = rp[0], b = rp[1];
		const PtxRay r{{a.x, a.y, a.z}, {a.w, b.x, b.y}, b.z, b.w};
		const uint32_t k = ray_sort_key(r, lo, scale, hi, t1, t2);
		keys[i] = k;
		atomicAdd(&hist[k & (SORT_DIGITS - 1u)], 1u);
	}
	__syncthreads();
	for (uint32_t d = threadIdx.x; d < SORT_DIGITS; d += blockDim.x) table[size_t(d) * n_tiles + blockIdx.x] = hist[d];
}

// Pass 1 count: high-digit histogram of a tile of the first pass's output ((key, ray index) pairs).
__global__ void __launch_bounds__(256) k_sort_count_keys(const uint2* __restrict__ pairs, uint64_t n, uint32_t* __restrict__ table, uint32_t n_tiles) {
	__shared__ uint32_t hist[SORT_DIGITS];
	for (uint32_t d = threadIdx.x; d < SORT_DIGITS; d += blockDim.x) hist[d] = 0;
	__syncthreads();
	const uint64_t begin = uint64_t(blockIdx.x) * SORT_TILE, end = begin + SORT_TILE < n ? begin + SORT_TILE : n;
	for (uint64_t i = begin + threadIdx.x; i < end; i += blockDim.x) atomicAdd(&hist[pairs[i].x >> SORT_DIGIT_BITS], 1u);
	__syncthreads();
	for (uint32_t d = threadIdx.x; d < SORT_DIGITS; d += blockDim.x) table[size_t(d) * n_tiles + blockIdx.x] = hist[d];
}

// Exclusive scan of the table in place, three small kernels:
//   k_scan_local   every block scans its own SCAN_CHUNK counters and publishes their total
//   k_scan_totals  one block scans the chunk totals (a few hundred values for 16M rays)
//   k_scan_add     every block adds its chunk's offset
// The table's length is a multiple of SCAN_CHUNK (the caller pads it with zeros).
constexpr uint32_t SCAN_CHUNK = 2048;   // 256 threads x 8 counters

__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t value, uint32_t* lds /* 256 entries */, uint32_t& total) {
	const uint32_t t = threadIdx.x;
	lds[t] = value;
	__syncthreads();
	for (uint32_t step = 1; step < 256; step <<= 1) {   // Hillis-Steele inclusive scan
		const uint32_t add = t >= step ? lds[t - step] : 0u;
		__syncthreads();
		lds[t] += add;
		__syncthreads();
	}
	total = lds[255];
	return lds[t] - value;
}

__global__ void __launch_bounds__(256) k_scan_local(uint32_t* __restrict__ table, uint32_t* __restrict__ totals) {
	__shared__ uint32_t lds[256];
	uint4* p = reinterpret_cast<uint4*>(table + size_t(blockIdx.x) * SCAN_CHUNK) + 2 * threadIdx.x;
	uint4 a = p[0], b = p[1];
	const uint32_t sum = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
	uint32_t total;
	uint32_t run = block_exclusive_scan_256(sum, lds, total);
	uint4 oa, ob;
	oa.x = run, run += a.x, oa.y = run, run += a.y, oa.z = run, run += a.z, oa.w = run, run += a.w;
	ob.x = run, run += b.x, ob.y = run, run += b.y, ob.z = run, run += b.z, ob.w = run;
	p[0] = oa, p[1] = ob;
	if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) k_scan_totals(uint32_t* __restrict__ totals, uint32_t count) {
	__shared__ uint32_t lds[256];
	const uint32_t per = (count + 255) / 256, begin = threadIdx.x * per, end = begin + per < count ? begin + per : count;
	uint32_t sum = 0;
	for (uint32_t k = begin; k < end; k++) sum += totals[k];
	uint32_t total;
	uint32_t run = block_exclusive_scan_256(sum, lds, total);
	for (uint32_t k = begin; k < end; k++) {
		const uint32_t c = totals[k];
		totals[k] = run;
		run += c;
	}
}

__global__ void __launch_bounds__(256) k_scan_add(uint32_t* __restrict__ table, const uint32_t* __restrict__ totals) {
	const uint32_t add = totals[blockIdx.x];
	uint4* p = reinterpret_cast<uint4*>(table + size_t(blockIdx.x) * SCAN_CHUNK) + 2 * threadIdx.x;
	uint4 a = p[0], b = p[1];
	a.x += add, a.y += add, a.z += add, a.w += add, b.x += add, b.y += add, b.z += add, b.w += add;
	p[0] = a, p[1] = b;
}

// Scatter of one pass.  FIRST: by the key's low digit; the key and the ray's own index travel together as one
// 8-byte pair (one scattered store per ray, and the ~16 rays a tile sends to a bin fill a 128-byte run).  Else: by the
// high digit of the first pass's pairs, and only the ray index is written (the final order).
template<bool FIRST>
__global__ void __launch_bounds__(256) k_sort_scatter(const uint32_t* __restrict__ keys_in, const uint2* __restrict__ pairs_in, uint64_t n,
                                                      const uint32_t* __restrict__ table, uint32_t n_tiles, uint2* __restrict__ pairs_out,
                                                      uint32_t* __restrict__ order_out) {
	__shared__ uint32_t offset[SORT_DIGITS];
	for (uint32_t d = threadIdx.x; d < SORT_DIGITS; d += blockDim.x) offset[d] = table[size_t(d) * n_tiles + blockIdx.x];
	__syncthreads();
	const uint64_t begin = uint64_t(blockIdx.x) * SORT_TILE, end = begin + SORT_TILE < n ? begin + SORT_TILE : n;
	// Four rays per thread and step: the four loads are in flight together, then the four LDS atomics, then the
	// four stores (a tile is walked by one block of four waves, so the latencies of a one-ray chain would show).
	constexpr uint32_t UNROLL = 4;
	for (uint64_t base = begin + threadIdx.x; base < end; base += uint64_t(UNROLL) * blockDim.x) {
		uint2 item[UNROLL];
		uint32_t pos[UNROLL];
#pragma unroll
		for (uint32_t u = 0; u < UNROLL; u++) {
			const uint64_t i = base + uint64_t(u) * blockDim.x;
			if (i < end) item[u] = FIRST ? make_uint2(keys_in[i], uint32_t(i)) : pairs_in[i];
		}
#pragma unroll
		for (uint32_t u = 0; u < UNROLL; u++) {
			const uint64_t i = base + uint64_t(u) * blockDim.x;
			if (i < end) pos[u] = atomicAdd(&offset[FIRST ? (item[u].x & (SORT_DIGITS - 1u)) : (item[u].x >> SORT_DIGIT_BITS)], 1u);
		}
#pragma unroll
		for (uint32_t u = 0; u < UNROLL; u++) {
			const uint64_t i = base + uint64_t(u) * blockDim.x;
			if (i < end) {
				if (FIRST) pairs_out[pos[u]] = item[u];
				else order_out[pos[u]] = item[u].y;
			}
		}
	}
}

}  // namespace ptx

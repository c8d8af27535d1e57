// Microbenchmark: how fast can a wave fetch one random 64-byte record per lane?
//   A  each lane loads its own record with 4 x 16-byte loads (what node_step does today)
//   B  the 4 lanes of a quad load one record together (one 16-byte quarter each), 4 instructions
//      cover the quad's 4 records; quarters are exchanged through LDS
//   C  like B but the exchange is skipped (load path only; result is wrong on purpose) -- upper bound
// Dependent chain: the next index comes from the loaded data, like a traversal.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                   \
	do {                                                                           \
		hipError_t e = (x);                                                        \
		if (e != hipSuccess) {                                                     \
			std::printf("%s failed: %s\n", #x, hipGetErrorString(e));              \
			std::exit(1);                                                          \
		}                                                                          \
	} while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
	x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
	return x;
}

template<int MODE>
__global__ void __launch_bounds__(256) k_gather(const float4* __restrict__ table, uint32_t n, int iters, uint32_t* __restrict__ out) {
	__shared__ float4 lds[256 * 4];
	const uint32_t lane = threadIdx.x & 63u, wave_base = threadIdx.x & ~63u;
	uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) % n;
	uint32_t acc = 0;
	for (int it = 0; it < iters; it++) {
		float4 q0, q1, q2, q3;
		if (MODE == 0) {
			const float4* p = table + size_t(idx) * 4;
			q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
		} else if (MODE == 3) {
			// quad-cooperative LDS-DMA: instruction j brings the records of lanes 16j..16j+15, one quad per
			// record, one quarter per lane; the quarter order is XOR-swizzled on the source side so that
			// the readers (one record per lane, stride 64 B) hit distinct banks
			__attribute__((address_space(3))) char* wave_lds = (__attribute__((address_space(3))) char*)lds + wave_base * 64;
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const uint32_t S = 16u * j + (lane >> 2);
				const uint32_t iS = __shfl(idx, S);
				const char* src = reinterpret_cast<const char*>(table + size_t(iS) * 4) + 16u * ((lane & 3u) ^ ((S >> 2) & 3u));
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), wave_lds + j * 1024, 16, 0, 0);
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			const float4* rd = lds + (wave_base + lane) * 4;
			const uint32_t sw = (lane >> 2) & 3u;
			q0 = rd[0 ^ sw], q1 = rd[1 ^ sw], q2 = rd[2 ^ sw], q3 = rd[3 ^ sw];
			__builtin_amdgcn_wave_barrier();
		} else {
			const uint32_t quad = lane & ~3u, k = lane & 3u;
			const uint32_t i0 = __shfl(idx, quad | 0), i1 = __shfl(idx, quad | 1), i2 = __shfl(idx, quad | 2), i3 = __shfl(idx, quad | 3);
			const float4 r0 = table[size_t(i0) * 4 + k], r1 = table[size_t(i1) * 4 + k], r2 = table[size_t(i2) * 4 + k], r3 = table[size_t(i3) * 4 + k];
			if (MODE == 1) {
				// record of lane (quad|j), quarter k  ->  lds[(wave_base + quad + j) * 4 + k]
				float4* w = lds + (wave_base + quad) * 4 + k;
				w[0] = r0, w[4] = r1, w[8] = r2, w[12] = r3;
				__builtin_amdgcn_wave_barrier();
				const float4* rd = lds + (wave_base + lane) * 4;
				q0 = rd[0], q1 = rd[1], q2 = rd[2], q3 = rd[3];
				__builtin_amdgcn_wave_barrier();
			} else {
				q0 = r0, q1 = r1, q2 = r2, q3 = r3;
			}
		}
		const uint32_t h = __float_as_uint(q0.x) ^ __float_as_uint(q1.y) ^ __float_as_uint(q2.z) ^ __float_as_uint(q3.w);
		acc += h;
		idx = mix(h + it + (blockIdx.x * 256u + threadIdx.x) * 0x9E3779B9u) % n;   // lane-dependent: chains of different lanes never merge
	}
	out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
	const uint32_t n = argc > 1 ? std::atoi(argv[1]) : 332255;   // 64-byte records (21 MB)
	const int iters = 400, blocks = 256 * 8;
	std::vector<float4> h(size_t(n) * 4);
	uint32_t s = 12345;
	for (auto& v : h) {
		s = s * 1664525u + 1013904223u;
		v = make_float4(float(s & 0xffffu), float((s >> 4) & 0xffffu), float((s >> 8) & 0xffffu), float((s >> 12) & 0xffffu));
	}
	float4* d;
	uint32_t* out;
	CHECK(hipMalloc(&d, h.size() * sizeof(float4)));
	CHECK(hipMalloc(&out, blocks * 256 * 4));
	CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	std::vector<uint32_t> r0(blocks * 256), r1(blocks * 256);
	for (int mode = 0; mode < 4; mode++) {
		float best = 1e9f;
		for (int rep = 0; rep < 3; rep++) {
			CHECK(hipEventRecord(e0));
			if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(256), 0, 0, d, n, iters, out);
			if (mode == 1) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, d, n, iters, out);
			if (mode == 3) hipLaunchKernelGGL(k_gather<3>, dim3(blocks), dim3(256), 0, 0, d, n, iters, out);
			if (mode == 2) hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(256), 0, 0, d, n, iters, out);
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			best = ms < best ? ms : best;
		}
		CHECK(hipMemcpy((mode == 0 ? r0 : r1).data(), out, blocks * 256 * 4, hipMemcpyDeviceToHost));
		const double records = double(blocks) * 256 * iters;
		std::printf("mode %d: %.3f ms, %.1f G records/s, %.1f GB/s%s\n", mode, best, records / best / 1e6, records * 64 / best / 1e6,
		            (mode == 1 || mode == 3) ? (r0 == r1 ? "  (results equal mode 0)" : "  (RESULTS DIFFER)") : "");
	}
	return 0;
}

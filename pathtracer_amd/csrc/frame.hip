// frame.hip -- the film kept on the GPU between calls, and its sum across GPUs.
//
// Pathtracer::accumulator / accumulator_samples (pathtracer.h:80-83) are one 40.24 fixed-point sum per channel and one
// sample count per pixel; every finished tile is added under a mutex (pathtracer.cpp:566-584, 407-427).  With several
// GPUs every GPU accumulates the tiles it was dealt in a film of its own and the films are added once somebody looks:
// integer sums, so the frame is the one a single accumulator would hold, whatever the order.
//
// Two exchanges:
//   ptx_frame_reduce       dst += sum of parts.  Parts on GPUs the destination can address (its own, and peers over
//                          xGMI once peer access is enabled) are read where they lie by ONE kernel that adds up to
//                          eight films at a time -- every source GPU has its own link to the destination, so the reads
//                          run side by side and nothing is staged.  Parts it cannot address are copied into staging
//                          films first (hipMemcpyPeerAsync, one copy stream per part).  Staging memory, streams and
//                          events stay with the destination film: after the first call nothing is allocated.
//   ptx_frame_reduce_rccl / ptx_frame_allreduce
//                          the same sums as RCCL collectives (ncclReduce / ncclAllReduce of int64 and uint32) over a
//                          communicator with one rank per GPU of this process.  RCCL is loaded on first use.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "scene.h"

using namespace ptx;

struct PtxFrame {
	PtxScene* scene;
	size_t n_pixels;
	int64_t* accum;
	uint32_t* samples;
	// ptx_frame_reduce's working state as a destination, kept between calls
	void* staging = nullptr;
	size_t staging_films = 0;
	std::vector<hipStream_t> copy_streams;
	std::vector<hipEvent_t> copied;
	PtxFrameReduceInfo last{};
};

extern "C" int ptx_frame_create(ptx_scene_t s, uint32_t width, uint32_t height, ptx_frame_t* out) {
	if (!s || !out || !width || !height) return fail(PTX_ERR_INVALID, "ptx_frame_create: null argument or empty film");
	PTX_HIP(hipSetDevice(s->device));
	PtxFrame* f = new PtxFrame{s, size_t(width) * height, nullptr, nullptr};
	const size_t a_bytes = f->n_pixels * 3 * sizeof(int64_t), s_bytes = f->n_pixels * sizeof(uint32_t);
	void* p = nullptr;
	if (hipMalloc(&p, a_bytes + s_bytes) != hipSuccess) {
		delete f;
		return fail(PTX_ERR_OOM, "ptx_frame_create: out of device memory");
	}
	f->accum = static_cast<int64_t*>(p);
	f->samples = reinterpret_cast<uint32_t*>(static_cast<char*>(p) + a_bytes);
	if (hipMemsetAsync(p, 0, a_bytes + s_bytes, s->stream) != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) {
		(void)hipFree(p);
		delete f;
		return fail(PTX_ERR_DEVICE, "ptx_frame_create: clearing the film failed");
	}
	*out = f;
	return PTX_OK;
}
extern "C" int64_t* ptx_frame_accum(ptx_frame_t f) { return f ? f->accum : nullptr; }
extern "C" uint32_t* ptx_frame_samples(ptx_frame_t f) { return f ? f->samples : nullptr; }
extern "C" int ptx_frame_upload(ptx_frame_t f, const int64_t* accum_rgb, const uint32_t* samples) {
	if (!f || !accum_rgb || !samples) return fail(PTX_ERR_INVALID, "ptx_frame_upload: null argument");
	PTX_HIP(hipSetDevice(f->scene->device));
	PTX_HIP(hipMemcpyAsync(f->accum, accum_rgb, f->n_pixels * 3 * sizeof(int64_t), hipMemcpyHostToDevice, f->scene->stream));
	PTX_HIP(hipMemcpyAsync(f->samples, samples, f->n_pixels * sizeof(uint32_t), hipMemcpyHostToDevice, f->scene->stream));
	PTX_HIP(hipStreamSynchronize(f->scene->stream));
	return PTX_OK;
}
extern "C" int ptx_frame_download(ptx_frame_t f, int64_t* accum_rgb, uint32_t* samples) {
	if (!f || !accum_rgb || !samples) return fail(PTX_ERR_INVALID, "ptx_frame_download: null argument");
	PTX_HIP(hipSetDevice(f->scene->device));
	PTX_HIP(hipMemcpyAsync(accum_rgb, f->accum, f->n_pixels * 3 * sizeof(int64_t), hipMemcpyDeviceToHost, f->scene->stream));
	PTX_HIP(hipMemcpyAsync(samples, f->samples, f->n_pixels * sizeof(uint32_t), hipMemcpyDeviceToHost, f->scene->stream));
	PTX_HIP(hipStreamSynchronize(f->scene->stream));
	return PTX_OK;
}
extern "C" int ptx_frame_clear(ptx_frame_t f) {
	if (!f) return fail(PTX_ERR_INVALID, "ptx_frame_clear: null frame");
	PTX_HIP(hipSetDevice(f->scene->device));
	PTX_HIP(hipMemsetAsync(f->accum, 0, f->n_pixels * (3 * sizeof(int64_t) + sizeof(uint32_t)), f->scene->stream));
	PTX_HIP(hipStreamSynchronize(f->scene->stream));
	return PTX_OK;
}

extern "C" void ptx_frame_destroy(ptx_frame_t f) {
	if (!f) return;
	(void)hipSetDevice(f->scene->device);
	for (hipStream_t st : f->copy_streams) (void)hipStreamDestroy(st);
	for (hipEvent_t ev : f->copied) (void)hipEventDestroy(ev);
	if (f->staging) (void)hipFree(f->staging);
	(void)hipFree(f->accum);
	delete f;
}

namespace ptx {

constexpr uint32_t ADD_MANY = 8;   // films one launch of k_frame_add_many reads: a GPU has seven xGMI peers
struct FilmSources {
	const char* film[ADD_MANY];    // a film is one allocation: 3 int64 sums per pixel, then one uint32 count per pixel
	uint32_t n;
};

// dst += sum of the source films: 3 int64 sums and one sample count per pixel (integer adds: any order gives the same
// frame).  Every thread has one 16-byte load per source in flight at a time; sources on other GPUs are read over
// their links side by side.
__global__ void __launch_bounds__(256) k_frame_add_many(unsigned long long* __restrict__ dst_acc, uint32_t* __restrict__ dst_smp, FilmSources src, size_t n_pixels) {
	const size_t stride = size_t(gridDim.x) * blockDim.x, first = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
	const size_t n_words = n_pixels * 3, n_pairs = n_words / 2;
	for (size_t i = first; i < n_pairs; i += stride) {
		ulonglong2 sum = reinterpret_cast<const ulonglong2*>(dst_acc)[i];
		for (uint32_t k = 0; k < src.n; k++) {
			const ulonglong2 v = reinterpret_cast<const ulonglong2*>(src.film[k])[i];
			sum.x += v.x, sum.y += v.y;
		}
		reinterpret_cast<ulonglong2*>(dst_acc)[i] = sum;
	}
	if (first == 0 && (n_words & 1)) {
		unsigned long long sum = dst_acc[n_words - 1];
		for (uint32_t k = 0; k < src.n; k++) sum += reinterpret_cast<const unsigned long long*>(src.film[k])[n_words - 1];
		dst_acc[n_words - 1] = sum;
	}
	const size_t smp_off = n_words * sizeof(unsigned long long);
	for (size_t i = first; i < n_pixels; i += stride) {
		uint32_t sum = dst_smp[i];
		for (uint32_t k = 0; k < src.n; k++) sum += reinterpret_cast<const uint32_t*>(src.film[k] + smp_off)[i];
		dst_smp[i] = sum;
	}
}

// Peer access is a property of a pair of devices in this process: enabled once, from the reading side.
constexpr int PEER_MAX_DEVICES = 64;
static std::mutex g_peer_mutex;
static signed char g_peer[PEER_MAX_DEVICES][PEER_MAX_DEVICES] = {};   // 0 not asked yet, 1 enabled, -1 not available

// Whether kernels on `reader` (the current device) may read memory of `owner`; enables the access on first use.
static bool peer_readable(int reader, int owner) {
	if (reader == owner) return true;
	if (reader < 0 || owner < 0 || reader >= PEER_MAX_DEVICES || owner >= PEER_MAX_DEVICES) return false;
	std::lock_guard<std::mutex> lock(g_peer_mutex);
	signed char& state = g_peer[reader][owner];
	if (state == 0) {
		int can = 0;
		state = -1;
		if (hipDeviceCanAccessPeer(&can, reader, owner) == hipSuccess && can) {
			const hipError_t e = hipDeviceEnablePeerAccess(owner, 0);
			if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) state = 1;
		}
		(void)hipGetLastError();
	}
	return state == 1;
}

// The films are read on other streams than the ones that wrote them (the destination's, or a communicator's): whatever
// their own GPUs still have queued for them has to be finished first.
static int wait_for_films(const ptx_frame_t* films, uint32_t n) {
	for (uint32_t i = 0; i < n; i++) {
		PTX_HIP(hipSetDevice(films[i]->scene->device));
		PTX_HIP(hipStreamSynchronize(films[i]->scene->stream));
	}
	return PTX_OK;
}

}  // namespace ptx

extern "C" int ptx_frame_reduce(ptx_frame_t dst, const ptx_frame_t* parts, uint32_t n_parts) {
	if (!dst || (n_parts && !parts)) return fail(PTX_ERR_INVALID, "ptx_frame_reduce: null argument");
	for (uint32_t i = 0; i < n_parts; i++) {
		if (!parts[i] || parts[i] == dst) return fail(PTX_ERR_INVALID, "ptx_frame_reduce: a part is null or is the destination itself");
		if (parts[i]->n_pixels != dst->n_pixels) return fail(PTX_ERR_INVALID, "ptx_frame_reduce: films of different sizes");
	}
	dst->last = PtxFrameReduceInfo{};
	if (n_parts == 0) return PTX_OK;
	if (int rc = wait_for_films(parts, n_parts)) return rc;
	const int dev = dst->scene->device;
	hipStream_t stream = dst->scene->stream;
	PTX_HIP(hipSetDevice(dev));
	const size_t bytes = dst->n_pixels * (3 * sizeof(int64_t) + sizeof(uint32_t));
	const dim3 block(256), grid(uint32_t(std::max<size_t>(1, std::min<size_t>((dst->n_pixels * 3 / 2 + 255) / 256, 256 * 16))));
	PtxFrameReduceInfo& info = dst->last;
	// Films the destination's GPU can read where they lie, up to eight per launch.
	FilmSources batch{};
	auto flush = [&]() -> hipError_t {
		if (batch.n == 0) return hipSuccess;
		hipLaunchKernelGGL(k_frame_add_many, grid, block, 0, stream, reinterpret_cast<unsigned long long*>(dst->accum), dst->samples, batch, dst->n_pixels);
		batch.n = 0;
		info.add_launches++;
		return hipGetLastError();
	};
	auto add = [&](const void* film) -> hipError_t {
		batch.film[batch.n++] = static_cast<const char*>(film);
		return batch.n == ADD_MANY ? flush() : hipSuccess;
	};
	// test switch: 1 stages every part (the copy path on one GPU), 2 stages every part on another GPU (no direct peer reads)
	const int stage_mode = getenv("PTX_FRAME_REDUCE_STAGE") ? atoi(getenv("PTX_FRAME_REDUCE_STAGE")) : 0;
	std::vector<const PtxFrame*> remote;
	hipError_t e = hipSuccess;
	for (uint32_t i = 0; i < n_parts && e == hipSuccess; i++) {
		const int owner = parts[i]->scene->device;
		if (owner == dev && stage_mode != 1) {
			info.parts_local++;
			e = add(parts[i]->accum);
		} else if (owner != dev && stage_mode == 0 && peer_readable(dev, owner)) {
			info.parts_peer_direct++;
			info.bytes_over_links += bytes;
			e = add(parts[i]->accum);
		} else {
			remote.push_back(parts[i]);
		}
	}
	if (e == hipSuccess) e = flush();
	// The rest is copied into staging films on the destination first, a group of them at a time, each copy on a
	// stream of its own (hipMemcpyPeer goes through the host where two GPUs have no direct path).
	const size_t group = std::max<size_t>(1, std::min<size_t>(remote.size(), (size_t(4) << 30) / bytes));   // at most 4 GB of staging films
	if (e == hipSuccess && !remote.empty()) {
		if (dst->staging_films < group) {
			if (dst->staging) (void)hipFree(dst->staging);
			dst->staging = nullptr, dst->staging_films = 0;
			e = hipMalloc(&dst->staging, group * bytes);
			if (e == hipSuccess) dst->staging_films = group, info.allocations++;
		}
		while (e == hipSuccess && dst->copy_streams.size() < group) {
			hipStream_t st = nullptr;
			e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
			if (e == hipSuccess) dst->copy_streams.push_back(st), info.allocations++;
		}
		while (e == hipSuccess && dst->copied.size() < group) {
			hipEvent_t ev = nullptr;
			e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
			if (e == hipSuccess) dst->copied.push_back(ev), info.allocations++;
		}
	}
	for (size_t first = 0; first < remote.size() && e == hipSuccess; first += group) {
		const size_t count = std::min(group, remote.size() - first);
		for (size_t k = 0; k < count && e == hipSuccess; k++) {
			char* film = static_cast<char*>(dst->staging) + k * bytes;
			const int owner = remote[first + k]->scene->device;
			e = hipMemcpyPeerAsync(film, dev, remote[first + k]->accum, owner, bytes, dst->copy_streams[k]);
			if (e == hipSuccess) e = hipEventRecord(dst->copied[k], dst->copy_streams[k]);
			info.parts_staged++;
			if (owner != dev) info.bytes_over_links += bytes;
		}
		for (size_t k = 0; k < count && e == hipSuccess; k++) {
			e = hipStreamWaitEvent(stream, dst->copied[k], 0);
			if (e == hipSuccess) e = add(static_cast<char*>(dst->staging) + k * bytes);
		}
		if (e == hipSuccess) e = flush();
		if (e == hipSuccess) e = hipStreamSynchronize(stream);   // the staging films are reused by the next group
	}
	if (e == hipSuccess) e = hipStreamSynchronize(stream);
	return e == hipSuccess ? PTX_OK : hip_fail(e, "ptx_frame_reduce");
}

extern "C" int ptx_frame_reduce_info(ptx_frame_t dst, PtxFrameReduceInfo* out) {
	if (!dst || !out) return fail(PTX_ERR_INVALID, "ptx_frame_reduce_info: null argument");
	*out = dst->last;
	return PTX_OK;
}

// ------------------------------------------------------------------ RCCL
namespace ptx {

struct RcclApi {
	void* handle = nullptr;
	std::string path;
	ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static std::mutex g_rccl_mutex;
static RcclApi g_rccl;

// RCCL is loaded when the first communicator is made, not with this library: a process that already holds a copy
// (PyTorch ships one) gets that copy, and a process that never sums films across GPUs never loads it.
static int load_rccl(const RcclApi** out) {
	std::lock_guard<std::mutex> lock(g_rccl_mutex);
	if (!g_rccl.handle) {
		const char* env = getenv("PTX_RCCL_LIB");
		const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
		std::string tried;
		void* h = nullptr;
		for (const char* name : names) {
			if (!name || !*name) continue;
			h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
			if (h) {
				g_rccl.path = name;
				break;
			}
			tried += std::string(tried.empty() ? "" : "; ") + dlerror();
		}
		if (!h) return fail(PTX_ERR_UNSUPPORTED, "RCCL could not be loaded (" + tried + ")");
		RcclApi api;
		api.handle = h, api.path = g_rccl.path;
		api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
		api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
		api.Reduce = reinterpret_cast<decltype(api.Reduce)>(dlsym(h, "ncclReduce"));
		api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
		api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(h, "ncclGroupStart"));
		api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
		api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
		if (!api.CommInitAll || !api.CommDestroy || !api.Reduce || !api.AllReduce || !api.GroupStart || !api.GroupEnd || !api.GetErrorString) {
			dlclose(h);
			return fail(PTX_ERR_UNSUPPORTED, "the RCCL library lacks an entry point this library needs");
		}
		g_rccl = api;
	}
	*out = &g_rccl;
	return PTX_OK;
}

static int rccl_fail(const RcclApi* api, ncclResult_t r, const char* what) { return fail(PTX_ERR_DEVICE, std::string(what) + ": " + api->GetErrorString(r)); }

}  // namespace ptx

struct PtxComm {
	const RcclApi* api = nullptr;
	std::vector<int> devices;
	std::vector<ncclComm_t> comms;
	std::vector<hipStream_t> streams;   // one per rank, on that rank's GPU
};

extern "C" void ptx_comm_destroy(ptx_comm_t c) {
	if (!c) return;
	for (size_t i = 0; i < c->comms.size(); i++)
		if (c->comms[i]) (void)c->api->CommDestroy(c->comms[i]);
	for (size_t i = 0; i < c->streams.size(); i++)
		if (c->streams[i] && hipSetDevice(c->devices[i]) == hipSuccess) (void)hipStreamDestroy(c->streams[i]);
	delete c;
}

extern "C" int ptx_comm_create(const int* devices, uint32_t n, ptx_comm_t* out) {
	if (!devices || !n || !out) return fail(PTX_ERR_INVALID, "ptx_comm_create: null argument or no devices");
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(PTX_ERR_NO_DEVICE, "no HIP device visible");
	for (uint32_t i = 0; i < n; i++) {
		if (devices[i] < 0 || devices[i] >= n_dev) return fail(PTX_ERR_NO_DEVICE, "ptx_comm_create: device index out of range");
		for (uint32_t k = 0; k < i; k++)
			if (devices[k] == devices[i]) return fail(PTX_ERR_INVALID, "ptx_comm_create: a GPU is listed twice (RCCL takes one rank per GPU)");
	}
	const RcclApi* api = nullptr;
	if (int rc = load_rccl(&api)) return rc;
	PtxComm* c = new PtxComm();
	c->api = api;
	c->devices.assign(devices, devices + n);
	c->comms.assign(n, nullptr);
	c->streams.assign(n, nullptr);
	const ncclResult_t r = api->CommInitAll(c->comms.data(), int(n), c->devices.data());
	if (r != ncclSuccess) {
		c->comms.assign(n, nullptr);
		ptx_comm_destroy(c);
		return rccl_fail(api, r, "ncclCommInitAll");
	}
	for (uint32_t i = 0; i < n; i++) {
		hipError_t e = hipSetDevice(c->devices[i]);
		if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->streams[i], hipStreamNonBlocking);
		if (e != hipSuccess) {
			ptx_comm_destroy(c);
			return hip_fail(e, "ptx_comm_create: stream");
		}
	}
	*out = c;
	return PTX_OK;
}

extern "C" uint32_t ptx_comm_size(ptx_comm_t c) { return c ? uint32_t(c->devices.size()) : 0u; }
extern "C" const char* ptx_comm_library(ptx_comm_t c) { return c ? c->api->path.c_str() : ""; }

// The checks both collectives share: one film per rank, on that rank's GPU, all of one size.
static int check_comm_frames(const char* who, ptx_comm_t c, const ptx_frame_t* frames, uint32_t n) {
	if (!c || !frames) return fail(PTX_ERR_INVALID, std::string(who) + ": null argument");
	if (n != c->devices.size()) return fail(PTX_ERR_INVALID, std::string(who) + ": one film per rank of the communicator is needed");
	for (uint32_t i = 0; i < n; i++) {
		if (!frames[i]) return fail(PTX_ERR_INVALID, std::string(who) + ": null film");
		if (frames[i]->scene->device != c->devices[i]) return fail(PTX_ERR_INVALID, std::string(who) + ": film i must live on the communicator's i-th GPU");
		if (frames[i]->n_pixels != frames[0]->n_pixels) return fail(PTX_ERR_INVALID, std::string(who) + ": films of different sizes");
	}
	return PTX_OK;
}

static int comm_wait(ptx_comm_t c) {
	for (size_t i = 0; i < c->streams.size(); i++) {
		PTX_HIP(hipSetDevice(c->devices[i]));
		PTX_HIP(hipStreamSynchronize(c->streams[i]));
	}
	return PTX_OK;
}

extern "C" int ptx_frame_reduce_rccl(ptx_comm_t c, ptx_frame_t dst, const ptx_frame_t* parts, uint32_t n_parts) {
	if (int rc = check_comm_frames("ptx_frame_reduce_rccl", c, parts, n_parts)) return rc;
	if (!dst || dst->n_pixels != parts[0]->n_pixels) return fail(PTX_ERR_INVALID, "ptx_frame_reduce_rccl: destination null or of another size");
	int root = -1;
	for (uint32_t i = 0; i < n_parts; i++)
		if (c->devices[i] == dst->scene->device) root = int(i);
	if (root < 0) return fail(PTX_ERR_INVALID, "ptx_frame_reduce_rccl: the destination's GPU is not in the communicator");
	const size_t n = dst->n_pixels;
	const RcclApi* api = c->api;
	if (int rc = wait_for_films(parts, n_parts)) return rc;
	if (int rc = wait_for_films(&dst, 1)) return rc;
	ncclResult_t r = api->GroupStart();
	for (uint32_t i = 0; i < n_parts && r == ncclSuccess; i++) {
		r = api->Reduce(parts[i]->accum, int(i) == root ? dst->accum : parts[i]->accum, 3 * n, ncclInt64, ncclSum, root, c->comms[i], c->streams[i]);
		if (r == ncclSuccess) r = api->Reduce(parts[i]->samples, int(i) == root ? dst->samples : parts[i]->samples, n, ncclUint32, ncclSum, root, c->comms[i], c->streams[i]);
	}
	const ncclResult_t r_end = api->GroupEnd();
	if (r != ncclSuccess) return rccl_fail(api, r, "ncclReduce");
	if (r_end != ncclSuccess) return rccl_fail(api, r_end, "ncclGroupEnd");
	return comm_wait(c);
}

extern "C" int ptx_frame_allreduce(ptx_comm_t c, const ptx_frame_t* frames, uint32_t n_frames) {
	if (int rc = check_comm_frames("ptx_frame_allreduce", c, frames, n_frames)) return rc;
	const size_t n = frames[0]->n_pixels;
	const RcclApi* api = c->api;
	if (int rc = wait_for_films(frames, n_frames)) return rc;
	ncclResult_t r = api->GroupStart();
	for (uint32_t i = 0; i < n_frames && r == ncclSuccess; i++) {
		r = api->AllReduce(frames[i]->accum, frames[i]->accum, 3 * n, ncclInt64, ncclSum, c->comms[i], c->streams[i]);
		if (r == ncclSuccess) r = api->AllReduce(frames[i]->samples, frames[i]->samples, n, ncclUint32, ncclSum, c->comms[i], c->streams[i]);
	}
	const ncclResult_t r_end = api->GroupEnd();
	if (r != ncclSuccess) return rccl_fail(api, r, "ncclAllReduce");
	if (r_end != ncclSuccess) return rccl_fail(api, r_end, "ncclGroupEnd");
	return comm_wait(c);
}

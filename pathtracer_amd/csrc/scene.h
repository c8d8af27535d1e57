// scene.h -- the host-side scene object behind ptx_scene_t.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/ptx.h"
#include "bvh_build.h"
#include "device_types.h"

namespace ptx {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
int hip_fail(hipError_t e, const char* what);

#define PTX_HIP(call)                                        \
	do {                                                     \
		hipError_t e_ = (call);                              \
		if (e_ != hipSuccess) return ptx::hip_fail(e_, #call); \
	} while (0)

// Material / texture / light tables as the shade stage reads them.
struct TexRec {
	uint32_t type, sampler, w, h;
	float color[3], scale;
	uint32_t pixel_base;   // first texel (float3) in the scene-wide texel array
	uint32_t pad[3];
};
struct MatRec {
	uint32_t type;
	int32_t tex_a, tex_b;
	float ior;
};
struct DeltaLightRec {
	uint32_t type, has_transform;
	float color[3], intensity, inner_angle, outer_angle;
	float T[16], iT[16];
};
struct EnvLightRec {
	uint32_t type;
	int32_t tex;
};

constexpr uint32_t MAX_EMISSIVE_INSTS = 8, MAX_EMISSIVE_PRIMS = 32;

struct ShadeTables {
	const TriAttr* attrs;
	const TexRec* textures;
	const float* texels;
	const MatRec* materials;
	const DeltaLightRec* delta_lights;
	const EnvLightRec* env_lights;
	uint32_t n_delta_lights, n_env_lights, fresnel_mode;
	// A direct-lighting ray (depth 0, pathtracer.cpp:64-71) brings back the emission of the first surface it meets and
	// nothing else.  When the scene has no environment light and its Emissive instances are few and small, a ray
	// that touches none of their primitives provably brings back zero and is not traced (render.hip, direct_ray_may_see_light).
	uint32_t direct_skip;                          // 1: the list below is complete and usable
	uint32_t n_emissive;
	uint32_t emissive_insts[MAX_EMISSIVE_INSTS];   // positions in SceneRec::insts
};

}  // namespace ptx

struct PtxScene;
namespace ptx {
// Grows (never shrinks) one of the scene's device scratch buffers.
int scene_scratch(PtxScene* s, int slot, size_t bytes);
// The scene's 256-byte block of device counters, allocated on first use.
int scene_counter_block(PtxScene* s, unsigned long long** out);
}  // namespace ptx

struct PtxBvh {
	ptx::HostBvh bvh;
	bool built_on_device = false;
};

struct PtxScene {
	int device = 0;
	hipStream_t stream = nullptr;
	ptx::SceneRec rec{};            // device pointers inside
	ptx::ShadeTables shade{};
	std::vector<void*> allocations; // everything hipMalloc'ed for this scene
	std::vector<ptx::HostBvh> mesh_bvhs;
	std::vector<uint8_t> mesh_built_on_device;   // per mesh: 1 when its tree came from the GPU builder
	ptx::HostBvh top_bvh;
	uint32_t stack_need = 0;        // deepest stack a traversal of this scene can need
	uint32_t cu_count = 256;
	bool persistent_ok = false;     // single tame BVH mesh: the replenishing kernel applies
	unsigned long long* next_ray = nullptr;   // device counter of the replenishing kernel
	unsigned char* noalloc_scratch = nullptr; // this device's copy of the fixed scratch array (ptx_trace_closest_noalloc)
	uint64_t bytes_nodes = 0, bytes_tris = 0;
	PtxCounters counters{};
	// Scratch grown on demand (host-pointer calls, render state).
	void* scratch[10] = {};
	size_t scratch_bytes[10] = {};
	// Device-side timing of the dominant traversal kernel: HIP events recorded on the scene's
	// stream right before and after each launch, kept in a ring (ptx_get_kernel_times).
	static constexpr uint32_t TIMING_RING = 64;
	hipEvent_t t_start[TIMING_RING] = {}, t_stop[TIMING_RING] = {};
	uint64_t timed_launches = 0;
	// Host-buffer ray queries in chunks (the upload of one chunk, the traversal of the one before and the
	// download of the one before that overlap): two copy streams and the chunks' events, created on first use.
	hipStream_t up_stream = nullptr, down_stream = nullptr;
	std::vector<hipEvent_t> chunk_events;
};

// bvh_build.h -- host-side binned-SAH builder: the GPU path's PT::BVH<P>::build.
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

#include "../../include/ptx.h"

namespace ptx {

struct HostBvh {
	std::vector<PtxBvhNode> nodes;   // canonical binary tree, root at 0, leaves have l == r
	std::vector<uint32_t> order;     // order[i] = input index of the i-th BVH-ordered primitive
	uint64_t clamped = 0;            // primitives whose bin index had to be clamped to [0,7]
	uint64_t predicate_disagreements = 0;
	uint64_t dropped = 0;            // primitives under nodes no split separates
	uint32_t max_depth = 0;          // root = depth 0
};

// Worker threads the host-side passes over large inputs use: the CPUs this process may run on, at most
// 16 (the environment variable PTX_BUILD_THREADS overrides).
unsigned build_threads();

// fn(begin, end) over [0, n) cut into one slice per thread; runs inline when n is small.
void parallel_slices(uint64_t n, const std::function<void(uint64_t, uint64_t)>& fn);

// Lap timer for the scene-construction passes, printing to stderr when PTX_BUILD_PROFILE is set.
struct BuildLaps {
	BuildLaps();
	void lap(const char* what);
	bool on;
	double last;
};

// Triangle::bbox (tri_mesh.cpp:10-44) for every triangle of an indexed mesh with 8-float vertices.
void triangle_boxes(const float* verts, const uint32_t* indices, uint32_t n_tris, std::vector<float>& boxes);

// PT::BVH<P>::build + buildRecursive (bvh.cpp:31-243) over n boxes (6 floats each).
void build_bvh(const float* boxes, uint32_t n, uint32_t max_leaf_size, HostBvh& out);

// The same tree built on the GPU `device` (bvh_build_gpu.hip).  `used_gpu` is false when the input went to the
// host builder instead (empty, or a negative zero / NaN among the box bounds).  Returns a PtxStatus.
int build_bvh_gpu(int device, const float* boxes, uint32_t n, uint32_t max_leaf_size, HostBvh& out, bool& used_gpu);

// Depth of a tree given as a node array (for injected trees).  Returns false when the
// array is not a tree rooted at 0 (child index out of range, cycle, range out of bounds).
bool validate_tree(const PtxBvhNode* nodes, uint32_t n_nodes, uint32_t n_prims, uint32_t& max_depth);

}  // namespace ptx

// device_types.h -- memory layouts shared by the host flattener and the gfx950 kernels.
//
// Everything here is resident in HBM for the life of a scene.  Sizes are chosen for whole
// 64-byte lines: one interior-node fetch is one line and brings both children's boxes, so a
// traversal step costs one dependent memory access, not two.
#pragma once
#include <stdint.h>

namespace ptx {

// Child reference.  bit 31 set: leaf, bits [29:4] = first primitive, bits [3:0] = count
// (0..15).  bit 31 clear: bits [29:0] = index of an interior node in the same tree's node
// array.  bit 30 (either kind): the subtree holds a primitive that sticks out of the
// subtree's box, so the box's entry distance is not a lower bound for the hits inside and
// the culled traversal modes must not skip it on distance.  (The reference builder can
// leave a primitive in a range whose box was built without it, bvh.cpp:191-221.)
constexpr uint32_t REF_LEAF = 0x80000000u;
constexpr uint32_t REF_NOCULL = 0x40000000u;
constexpr uint32_t REF_INDEX_MASK = 0x3FFFFFFFu;
constexpr uint32_t MAX_LEAF_PRIMS = 15u;
constexpr uint32_t MAX_LEAF_START = (1u << 26) - 2u;   // keeps 0xFFFFFFFF free as the traversal's end marker
inline constexpr uint32_t make_leaf_ref(uint32_t start, uint32_t count) { return REF_LEAF | (start << 4) | count; }

// Interior node of a binary tree: the boxes of both children and where they lead.
// PT::BVH keeps one box per node (bvh.h:15-23) and tests it on entry (bvh.cpp:278-286);
// holding the two child boxes in the parent is the same set of tests, one level earlier.
// The twelve box coordinates are ordered for the two-wide vector ALU operations of the slab test: every 8-byte
// pair below meets one pair of ray registers ((ox, oy) or (oz, oz)) in a single packed subtract and multiply, with
// no register moves between the load and the arithmetic.
struct alignas(64) Node2 {
	float lxy[4];   // left child:  min.x, min.y, max.x, max.y
	float z[4];     // left min.z, left max.z, right min.z, right max.z
	float rxy[4];   // right child: min.x, min.y, max.x, max.y
	uint32_t lref, rref;
	uint32_t pad[2];
	// side 0: left child, 1: right child
	void set_box(int side, const float* bmin, const float* bmax) {
		float* xy = side ? rxy : lxy;
		xy[0] = bmin[0], xy[1] = bmin[1], xy[2] = bmax[0], xy[3] = bmax[1];
		z[2 * side] = bmin[2], z[2 * side + 1] = bmax[2];
	}
	void get_box(int side, float* bmin, float* bmax) const {
		const float* xy = side ? rxy : lxy;
		bmin[0] = xy[0], bmin[1] = xy[1], bmax[0] = xy[2], bmax[1] = xy[3];
		bmin[2] = z[2 * side], bmax[2] = z[2 * side + 1];
	}
};
// The children's boxes from a node's first three 16-byte pieces, as (min.x, min.y, min.z, max.x, max.y, max.z).
#define PTX_NODE_LEFT_BOX(n0, n1, n2) (n0).x, (n0).y, (n1).x, (n0).z, (n0).w, (n1).y
#define PTX_NODE_RIGHT_BOX(n0, n1, n2) (n2).x, (n2).y, (n1).z, (n2).z, (n2).w, (n1).w
static_assert(sizeof(Node2) == 64, "one cache line");

// Interior node of the four-wide form of the same tree: a binary node with its interior children replaced by
// their children (where the grandchild's box lies inside the child's, so that passing the grandchild's slab test
// implies passing the child's -- the chain of tests PT::BVH::hitRecursive makes on the way down, bvh.cpp:278-286).
// Unused children carry REF_NONE4 and a zero box.
constexpr uint32_t REF_NONE4 = 0xFFFFFFFEu;
struct alignas(128) Node4 {
	Node2 half[2];   // children 0,1 and children 2,3, each pair laid out as a binary node's pair (one cache line)
};
static_assert(sizeof(Node4) == 128, "two cache lines");


// One triangle as the intersector wants it: v0 and the two edges e1 = v1 - v0, e2 = v2 - v0
// (the subtractions Triangle::hit performs per call, tri_mesh.cpp:57-59, done once in fp32
// on the host: same operands, same rounding, same bits).
struct alignas(16) TriGeom {
	float v0[3], pad0;
	float e1[3], pad1;
	float e2[3], pad2;
};
static_assert(sizeof(TriGeom) == 48, "three 16-byte loads");

// Shading attributes, read only by the shade stage: per-vertex normals and uvs.
struct TriAttr {
	float n0[3], n1[3], n2[3];
	float uv0[2], uv1[2], uv2[2];
};
static_assert(sizeof(TriAttr) == 60, "");

// One mesh (PT::Tri_Mesh): a tree over a range of the scene-wide triangle arrays.
struct MeshRec {
	float root_min[3], root_max[3];
	uint32_t root_ref;      // leaf ref or interior index (relative to node_base)
	uint32_t node_base;     // first Node2 of this mesh in the scene-wide node array
	uint32_t tri_base;      // first triangle of this mesh in the scene-wide triangle arrays
	uint32_t n_tris;
	uint32_t is_list;       // List<Triangle>: no box test, fold over all n_tris (list.h:25-32)
	uint32_t tame;          // every box coordinate is 0 or in [2^-70, 2^39): the short division form is exact
	uint32_t root4_ref;     // root of the four-wide form (leaf ref or index relative to node4_base); valid when SceneRec::nodes4 is set
	uint32_t node4_base;
};

// One PT::Instance (instance.h:15-40).  Matrices column-major, element [col*4+row].
struct InstRec {
	float T[16];
	float iT[16];
	uint32_t has_transform;
	int32_t mesh;           // >= 0: MeshRec index; -1: Shapes::Sphere of `radius`
	float radius;
	int32_t material;
};

struct SceneRec {
	const Node2* nodes;       // all trees' interior nodes
	const Node4* nodes4;      // four-wide form of the mesh trees (single-mesh scenes only, else null)
	const TriGeom* tris;
	const TriAttr* attrs;
	const MeshRec* meshes;
	const InstRec* insts;     // BVH-ordered instance array (Aggregate's BVH<Instance>::primitives)
	// Top level
	float top_min[3], top_max[3];
	uint32_t top_ref;
	uint32_t top_node_base;
	uint32_t n_insts;
	uint32_t top_is_list;     // List<Instance> (--no_bvh)
	uint32_t mesh_only;       // queries go straight to meshes[0] (PT::Tri_Mesh::hit)
	uint32_t top_tame;        // as MeshRec::tame, for the top-level tree
	uint32_t top_stack_slots; // stack slots reserved for the top-level traversal
	uint32_t lds_mesh;        // the mesh most instances refer to (experiment switch PTX_SCENE_TOP_LDS)
};

// Hit record passed from the extend stage to the shade stage (two 16-byte stores).
// `t` is the ray parameter in the space the primitive was intersected in: the shade stage
// rebuilds Trace::position as point + dir * t there (tri_mesh.cpp:91-99) and carries it to
// world space (trace.h:38-43).
struct alignas(16) HitRec {
	float distance;   // world-space Trace::distance
	float t;
	float u, v;
	uint32_t prim;    // BVH-order index within the mesh, or 0xFFFFFFFF
	uint32_t inst;    // BVH-order instance index
	uint32_t pad[2];
};
static_assert(sizeof(HitRec) == 32, "");

}  // namespace ptx

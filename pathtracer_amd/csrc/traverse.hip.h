// traverse.hip.h -- gfx950 device code for ray queries against the flattened scene.
//
// Arithmetic contract: every value that decides a result is computed with the reference's
// operations in the reference's order, in IEEE binary32 with true division and sqrt, no
// fused multiply-add (this file must be compiled with -ffp-contract=off and without any
// fast-math option) and subnormals kept.  The comments cite the reference lines restated.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace ptx {

constexpr int STACK_CAP = 64;            // traversal stack entries per lane
constexpr float CULL_SCALE = 1.0001f;    // guard band of the culled modes (see ptx.h)
constexpr float CULL_BIAS = 1e-7f;

enum TraverseMode { MODE_REFERENCE = 0, MODE_CULLED = 1 };

struct Ray3 {
	float ox, oy, oz, dx, dy, dz, tmin, tmax;
};

struct MeshHit {
	bool hit;
	float distance;   // Trace::distance in the space the mesh was intersected in
	float t;          // ray parameter of the winning triangle (dist = dir * t, tri_mesh.cpp:91)
	uint32_t prim;
	float u, v;
};

__device__ __forceinline__ float norm3(float x, float y, float z) { return sqrtf(x * x + y * y + z * z); }  // vec3.h:155-160

// BBox::hit(ray, times = (0, inf)) -- lib/bbox.h:81-154.  Returns the accept/reject decision;
// `tnear` receives min_intersect (bbox.h:137), used only to order and cull in CULLED mode.
__device__ __forceinline__ bool slab_test(float minx, float miny, float minz, float maxx, float maxy, float maxz, const Ray3& r,
                                          float& tnear) {
	if (r.dx == 0.0f && (r.ox < minx || r.ox > maxx)) return false;   // bbox.h:98-106
	if (r.dy == 0.0f && (r.oy < miny || r.oy > maxy)) return false;
	if (r.dz == 0.0f && (r.oz < minz || r.oz > maxz)) return false;
	float tx0 = (minx - r.ox) / r.dx, tx1 = (maxx - r.ox) / r.dx;     // bbox.h:110-120
	if (tx0 > tx1) { float s = tx0; tx0 = tx1; tx1 = s; }
	float ty0 = (miny - r.oy) / r.dy, ty1 = (maxy - r.oy) / r.dy;
	if (ty0 > ty1) { float s = ty0; ty0 = ty1; ty1 = s; }
	float tz0 = (minz - r.oz) / r.dz, tz1 = (maxz - r.oz) / r.dz;
	if (tz0 > tz1) { float s = tz0; tz0 = tz1; tz1 = s; }
	if (tx0 > ty1 || tx1 < ty0) return false;                          // bbox.h:124
	float lo = (tx0 > ty0) ? tx0 : ty0;                                // bbox.h:128-129
	float hi = (tx1 < ty1) ? tx1 : ty1;
	if (lo > tz1 || hi < tz0) return false;                            // bbox.h:133
	lo = (lo > tz0) ? lo : tz0;                                        // bbox.h:137-138
	hi = (hi < tz1) ? hi : tz1;
	if (lo > INFINITY || hi < 0.0f) return false;                      // bbox.h:142 with times = (0, inf)
	tnear = lo;
	return true;
}

// Triangle::hit -- pathtracer/tri_mesh.cpp:46-111.
__device__ __forceinline__ bool triangle_test(const TriGeom* __restrict__ tris, uint32_t index, const Ray3& r, float& distance,
                                              float& t_out, float& u_out, float& v_out) {
	const float4* p = reinterpret_cast<const float4*>(tris + index);
	const float4 a = p[0], b = p[1], c = p[2];   // v0, e1, e2
	const float sx = r.ox - a.x, sy = r.oy - a.y, sz = r.oz - a.z;             // s = ray.point - v0
	// cross(e1, dir)
	const float c1x = b.y * r.dz - b.z * r.dy, c1y = b.z * r.dx - b.x * r.dz, c1z = b.x * r.dy - b.y * r.dx;
	const float denom = c1x * c.x + c1y * c.y + c1z * c.z;                     // dot(cross(e1,dir), e2)
	if (denom == 0.0f) return false;
	// cross(s, e2)
	const float c2x = sy * c.z - sz * c.y, c2y = sz * c.x - sx * c.z, c2z = sx * c.y - sy * c.x;
	const float cu = -(c2x * r.dx + c2y * r.dy + c2z * r.dz);                  // -1 * dot(cross(s,e2), dir)
	const float cv = c1x * sx + c1y * sy + c1z * sz;                           // dot(cross(e1,dir), s)
	const float ct = -(c2x * b.x + c2y * b.y + c2z * b.z);                     // -dot(cross(s,e2), e1)
	const float u = cu / denom, v = cv / denom, t = ct / denom;
	if (u < 0.0f || v < 0.0f || u + v > 1.0f || t < 0.0f) return false;        // tri_mesh.cpp:87
	const float d = norm3(r.dx * t, r.dy * t, r.dz * t);                       // (ray.dir * t).norm()
	if (d > r.tmax || d < r.tmin) return false;                                // tri_mesh.cpp:93
	distance = d, t_out = t, u_out = u, v_out = v;
	return true;
}

// Trace::min(cur, candidate) for a candidate that hit -- trace.h:28-36: the candidate
// replaces the running hit unless the running hit is strictly closer.
__device__ __forceinline__ void merge_reference(MeshHit& cur, float d, float t, uint32_t prim, float u, float v) {
	if (cur.hit && cur.distance < d) return;
	cur.hit = true, cur.distance = d, cur.t = t, cur.prim = prim, cur.u = u, cur.v = v;
}
// Order-independent form of the same rule: closest distance, and among equal distances the
// largest array position (what a left-to-right fold with the rule above ends with).
__device__ __forceinline__ void merge_any_order(MeshHit& cur, float d, float t, uint32_t prim, float u, float v) {
	if (cur.hit && !(d < cur.distance || (d == cur.distance && prim > cur.prim))) return;
	cur.hit = true, cur.distance = d, cur.t = t, cur.prim = prim, cur.u = u, cur.v = v;
}

struct Stack {
	uint32_t ref[STACK_CAP];
	float tnear[STACK_CAP];
};

// PT::BVH<Triangle>::hit / hitRecursive (bvh.cpp:246-375) or List<Triangle>::hit
// (list.h:25-32) for one mesh, with the stack slots from `base` upward.
// ANY: stop at the first accepted triangle (occlusion query).
template<int MODE, bool ANY>
__device__ __forceinline__ void traverse_mesh(const SceneRec& sc, const MeshRec& mesh, const Ray3& r, Stack& st, int base, MeshHit& cur) {
	const TriGeom* __restrict__ tris = sc.tris + mesh.tri_base;
	const Node2* __restrict__ nodes = sc.nodes + mesh.node_base;

	if (mesh.is_list) {
		// List::hit starts from a default Trace (hit = false, distance = 0) and never tests a box.
		cur.hit = false, cur.distance = 0.0f, cur.prim = 0xFFFFFFFFu;
		for (uint32_t i = 0; i < mesh.n_tris; i++) {
			float d, t, u, v;
			if (triangle_test(tris, i, r, d, t, u, v)) {
				merge_reference(cur, d, t, i, u, v);
				if (ANY) return;
			}
		}
		return;
	}

	// BVH::hit seeds the running Trace with hit = false, distance = inf (bvh.cpp:257-259).
	cur.hit = false, cur.distance = INFINITY, cur.prim = 0xFFFFFFFFu;
	float tn;
	if (!slab_test(mesh.root_min[0], mesh.root_min[1], mesh.root_min[2], mesh.root_max[0], mesh.root_max[1], mesh.root_max[2], r, tn)) return;

	int sp = base;
	uint32_t ref = mesh.root_ref;
	for (;;) {
		if (ref & REF_LEAF) {
			const uint32_t first = (ref & REF_INDEX_MASK) >> 4, count = ref & 15u;
			for (uint32_t i = first; i < first + count; i++) {
				float d, t, u, v;
				if (triangle_test(tris, i, r, d, t, u, v)) {
					if (MODE == MODE_REFERENCE) merge_reference(cur, d, t, i, u, v);
					else merge_any_order(cur, d, t, i, u, v);
					if (ANY) return;
				} else if (!cur.hit) {
					cur.distance = 0.0f;   // Trace::min of two misses is a default Trace (trace.h:35)
				}
			}
		} else {
			const float4* p = reinterpret_cast<const float4*>(nodes + (ref & REF_INDEX_MASK));
			const float4 n0 = p[0], n1 = p[1], n2 = p[2];
			const uint2 kids = *reinterpret_cast<const uint2*>(p + 3);
			float tl, tr;
			const bool hl = slab_test(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, r, tl);
			const bool hr = slab_test(n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, r, tr);
			if (MODE == MODE_REFERENCE) {
				// Left subtree completely, then right (bvh.cpp:326-342: both entry times are 0).
				if (hl) {
					if (hr) st.ref[sp++] = kids.y;
					ref = kids.x;
					continue;
				}
				if (hr && !(cur.hit && !(cur.distance >= 0.0f))) {   // bvh.cpp:337-339 (NaN distance skips the sibling)
					ref = kids.y;
					continue;
				}
			} else {
				const float limit = ANY ? r.tmax * CULL_SCALE + CULL_BIAS : (cur.hit ? cur.distance * CULL_SCALE + CULL_BIAS : INFINITY);
				const bool gl = hl && ((kids.x & REF_NOCULL) || !(tl > limit)), gr = hr && ((kids.y & REF_NOCULL) || !(tr > limit));
				if (gl && gr) {
					const bool left_first = !(tr < tl);
					st.ref[sp] = left_first ? kids.y : kids.x;
					st.tnear[sp] = left_first ? tr : tl;
					sp++;
					ref = left_first ? kids.x : kids.y;
					continue;
				}
				if (gl) { ref = kids.x; continue; }
				if (gr) { ref = kids.y; continue; }
			}
		}
		// pop
		for (;;) {
			if (sp == base) return;
			--sp;
			ref = st.ref[sp];
			if (MODE == MODE_REFERENCE) {
				if (cur.hit && !(cur.distance >= 0.0f)) continue;   // bvh.cpp:337-339
				break;
			} else {
				if (!ANY && cur.hit && !(ref & REF_NOCULL) && st.tnear[sp] > cur.distance * CULL_SCALE + CULL_BIAS) continue;
				break;
			}
		}
	}
}

// ---- Mat4 * Vec3 forms (lib/mat4.h:175-186) ---------------------------------------------
__device__ __forceinline__ void mat_point(const float* __restrict__ m, float x, float y, float z, float& ox, float& oy, float& oz) {
	// v[0]*cols[0] + v[1]*cols[1] + v[2]*cols[2] + v[3]*cols[3] with v[3] = 1, then project()
	const float rx = ((x * m[0] + y * m[4]) + z * m[8]) + 1.0f * m[12];
	const float ry = ((x * m[1] + y * m[5]) + z * m[9]) + 1.0f * m[13];
	const float rz = ((x * m[2] + y * m[6]) + z * m[10]) + 1.0f * m[14];
	const float rw = ((x * m[3] + y * m[7]) + z * m[11]) + 1.0f * m[15];
	ox = rx / rw, oy = ry / rw, oz = rz / rw;
}
__device__ __forceinline__ void mat_rotate(const float* __restrict__ m, float x, float y, float z, float& ox, float& oy, float& oz) {
	ox = ((x * m[0] + y * m[4]) + z * m[8]) + 0.0f * m[12];
	oy = ((x * m[1] + y * m[5]) + z * m[9]) + 0.0f * m[13];
	oz = ((x * m[2] + y * m[6]) + z * m[10]) + 0.0f * m[14];
}

// Ray::transform (lib/ray.h:32-38)
__device__ __forceinline__ Ray3 ray_transform(const float* __restrict__ m, const Ray3& r) {
	Ray3 o;
	mat_point(m, r.ox, r.oy, r.oz, o.ox, o.oy, o.oz);
	float dx, dy, dz;
	mat_rotate(m, r.dx, r.dy, r.dz, dx, dy, dz);
	const float d = norm3(dx, dy, dz);
	o.tmin = r.tmin * d, o.tmax = r.tmax * d;
	o.dx = dx / d, o.dy = dy / d, o.dz = dz / d;
	return o;
}

// Shapes::Sphere::hit (scene/shape.cpp:21-85); t doubles as distance.
__device__ __forceinline__ bool sphere_test(float radius, const Ray3& r, float& t_out) {
	const float d2 = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;
	const float o2 = r.ox * r.ox + r.oy * r.oy + r.oz * r.oz;
	const float od = r.ox * r.dx + r.oy * r.dy + r.oz * r.dz;
	const float disc = 4 * (od * od) - 4 * d2 * (o2 - radius * radius);
	if (disc < 0.0f) return false;
	const float sq = sqrtf(disc);
	const float t1 = (-2 * od + sq) / (2 * d2);
	const float t2 = (-2 * od - sq) / (2 * d2);
	const bool w1 = t1 >= r.tmin && t1 <= r.tmax, w2 = t2 >= r.tmin && t2 <= r.tmax;
	if (w1 && w2) t_out = fminf(t1, t2);   // std::min(t1, t2): both are ordinary numbers here
	else if (w1) t_out = t1;
	else if (w2) t_out = t2;
	else return false;
	return true;
}

struct SceneHit {
	bool hit;
	float distance;    // world-space Trace::distance
	uint32_t prim, inst;
	float u, v;
};

// Instance::hit (pathtracer/instance.h:32-40): ray into object space, geometry hit, Trace
// back to world space with distance recomputed from the transformed points (trace.h:38-43).
template<int MODE, bool ANY>
__device__ __forceinline__ bool instance_hit(const SceneRec& sc, uint32_t inst_index, const Ray3& world, Stack& st, int base, SceneHit& out) {
	const InstRec& in = sc.insts[inst_index];
	Ray3 r = world;
	if (in.has_transform) r = ray_transform(in.iT, world);
	float t, dist, u = 0.0f, v = 0.0f;
	uint32_t prim = 0;
	if (in.mesh >= 0) {
		MeshHit mh;
		traverse_mesh<MODE, ANY>(sc, sc.meshes[in.mesh], r, st, base, mh);
		if (!mh.hit) return false;
		t = mh.t, dist = mh.distance, u = mh.u, v = mh.v, prim = mh.prim;
	} else {
		if (!sphere_test(in.radius, r, t)) return false;
		dist = t;
	}
	if (in.has_transform) {
		// position = ray.point + dir * t (object space), then T * position, T * origin
		const float px = r.ox + r.dx * t, py = r.oy + r.dy * t, pz = r.oz + r.dz * t;
		float wx, wy, wz, qx, qy, qz;
		mat_point(in.T, px, py, pz, wx, wy, wz);
		mat_point(in.T, r.ox, r.oy, r.oz, qx, qy, qz);
		dist = norm3(wx - qx, wy - qy, wz - qz);
	}
	out.hit = true, out.distance = dist, out.prim = prim, out.inst = inst_index, out.u = u, out.v = v;
	return true;
}

// Aggregate::hit over BVH<Instance> / List<Instance> (aggregate.h:41-43), or Tri_Mesh::hit
// when the scene is a single mesh.
template<int MODE, bool ANY>
__device__ __forceinline__ void traverse_scene(const SceneRec& sc, const Ray3& r, Stack& st, SceneHit& cur) {
	if (sc.mesh_only) {
		MeshHit mh;
		traverse_mesh<MODE, ANY>(sc, sc.meshes[0], r, st, 0, mh);
		cur.hit = mh.hit, cur.distance = mh.distance, cur.prim = mh.prim, cur.inst = 0, cur.u = mh.u, cur.v = mh.v;
		return;
	}
	auto merge = [&](const SceneHit& h) {
		if (MODE == MODE_REFERENCE) {
			if (cur.hit && cur.distance < h.distance) return;
		} else {
			if (cur.hit && !(h.distance < cur.distance || (h.distance == cur.distance && h.inst > cur.inst))) return;
		}
		cur = h;
	};
	if (sc.top_is_list) {
		cur.hit = false, cur.distance = 0.0f, cur.prim = 0xFFFFFFFFu, cur.inst = 0xFFFFFFFFu;
		for (uint32_t i = 0; i < sc.n_insts; i++) {
			SceneHit h;
			if (instance_hit<MODE, ANY>(sc, i, r, st, 0, h)) {
				merge(h);
				if (ANY) return;
			}
		}
		return;
	}
	cur.hit = false, cur.distance = INFINITY, cur.prim = 0xFFFFFFFFu, cur.inst = 0xFFFFFFFFu;
	float tn;
	if (!slab_test(sc.top_min[0], sc.top_min[1], sc.top_min[2], sc.top_max[0], sc.top_max[1], sc.top_max[2], r, tn)) return;
	const Node2* __restrict__ nodes = sc.nodes + sc.top_node_base;
	int sp = 0;
	uint32_t ref = sc.top_ref;
	for (;;) {
		if (ref & REF_LEAF) {
			const uint32_t first = (ref & REF_INDEX_MASK) >> 4, count = ref & 15u;
			for (uint32_t i = first; i < first + count; i++) {
				SceneHit h;
				if (instance_hit<MODE, ANY>(sc, i, r, st, sp, h)) {
					merge(h);
					if (ANY) return;
				} else if (!cur.hit) {
					cur.distance = 0.0f;
				}
			}
		} else {
			const float4* p = reinterpret_cast<const float4*>(nodes + (ref & REF_INDEX_MASK));
			const float4 n0 = p[0], n1 = p[1], n2 = p[2];
			const uint2 kids = *reinterpret_cast<const uint2*>(p + 3);
			float tl, tr;
			const bool hl = slab_test(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, r, tl);
			const bool hr = slab_test(n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, r, tr);
			if (MODE == MODE_REFERENCE) {
				if (hl) {
					if (hr) st.ref[sp++] = kids.y;
					ref = kids.x;
					continue;
				}
				if (hr && !(cur.hit && !(cur.distance >= 0.0f))) {
					ref = kids.y;
					continue;
				}
			} else {
				const float limit = ANY ? r.tmax * CULL_SCALE + CULL_BIAS : (cur.hit ? cur.distance * CULL_SCALE + CULL_BIAS : INFINITY);
				const bool gl = hl && ((kids.x & REF_NOCULL) || !(tl > limit)), gr = hr && ((kids.y & REF_NOCULL) || !(tr > limit));
				if (gl && gr) {
					const bool left_first = !(tr < tl);
					st.ref[sp] = left_first ? kids.y : kids.x;
					st.tnear[sp] = left_first ? tr : tl;
					sp++;
					ref = left_first ? kids.x : kids.y;
					continue;
				}
				if (gl) { ref = kids.x; continue; }
				if (gr) { ref = kids.y; continue; }
			}
		}
		for (;;) {
			if (sp == 0) return;
			--sp;
			ref = st.ref[sp];
			if (MODE == MODE_REFERENCE) {
				if (cur.hit && !(cur.distance >= 0.0f)) continue;
				break;
			} else {
				if (!ANY && cur.hit && !(ref & REF_NOCULL) && st.tnear[sp] > cur.distance * CULL_SCALE + CULL_BIAS) continue;
				break;
			}
		}
	}
}

}  // namespace ptx

// traverse.hip.h -- gfx950 device code for ray queries against the flattened scene.
//
// Arithmetic contract: every value that decides a result is computed with the reference's
// operations in the reference's order, in IEEE binary32 with correctly rounded division and
// sqrt, no fused multiply-add where the reference has a multiply and an add (this file must
// be compiled with -ffp-contract=off and without any fast-math option) and subnormals kept.
// The comments cite the reference lines restated.
//
// Division.  BBox::hit divides six times per box (bbox.h:110-120).  hipcc expands an IEEE
// f32 division into v_div_scale x2, v_rcp, six fma/mul, v_div_fmas and v_div_fixup.  All
// six divisions of a box test share one of three divisors (the ray direction), so the
// divisor-only part of that expansion (v_rcp + two fma) is done once per ray, and each
// quotient then costs the numerator-dependent part only: one multiply and four fma, the
// very operations, in the order, of the compiler's expansion -- which is what makes the
// quotient bit-identical to `a / b`.  The expansion's scaling and fix-up steps exist for
// operands near the ends of the exponent range; the short form is used only when static
// range checks on the ray and on the tree's coordinates prove they would be no-ops
// (`FAST`), otherwise the plain `/` is used.  ptx_selftest_division compares the two forms
// on the device.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace ptx {

constexpr int STACK_CAP = 64;            // traversal stack entries per lane
#ifndef PTX_CULL_SCALE
#define PTX_CULL_SCALE 1.01f
#endif
// Guard band of the culled modes (see ptx.h): a subtree is skipped only when its box entry lies more than 1 % beyond
// the running closest distance.  Triangle::hit's distance carries a relative error of about 2^-22 / sin(grazing
// angle), so the band covers every hit whose ray meets its triangle at more than 0.002 degrees.  Measured cost
// against a band of 0.01 %: 0.5 % on incoherent rays over a 1M-triangle mesh, 1-2 % on instanced-scene renders.
constexpr float CULL_SCALE = PTX_CULL_SCALE;
constexpr float CULL_BIAS = 1e-7f;
constexpr uint32_t REF_NONE = 0xFFFFFFFEu;   // transient: "take the next pending subtree" (see node_decide)
constexpr uint32_t REF_DONE = 0xFFFFFFFFu;   // leaf-shaped sentinel (no real leaf has both flag bits, start 2^26-1 and count 15)

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

struct TraverseStats {
	uint32_t nodes;   // interior nodes fetched (two box tests each)
	uint32_t leaves;  // leaves entered
	uint32_t tris;    // triangle tests
	uint32_t max_sp;  // deepest stack seen
};

__device__ __forceinline__ float norm3(float x, float y, float z) { return sqrtf(x * x + y * y + z * z); }  // vec3.h:155-160

// ---- division -----------------------------------------------------------------------------
// The divisor-only half of the f32 division expansion: v_rcp_f32 refined by one Newton step.
__device__ __forceinline__ float refined_rcp(float b) {
	const float r = __builtin_amdgcn_rcpf(b);
	const float e = fmaf(-b, r, 1.0f);
	return fmaf(e, r, r);
}
// The numerator-dependent half: a / b given y = refined_rcp(b).  Equal to the IEEE quotient
// when no scaling or fix-up would have been needed (see ray_in_fast_range below).
__device__ __forceinline__ float div_by(float a, float b, float y) {
	float q = a * y;
	float r = fmaf(-b, q, a);
	q = fmaf(r, y, q);
	r = fmaf(-b, q, a);
	return fmaf(r, y, q);
}
// |x| in [2^lo, 2^hi] (x finite, normal), decided on the exponent field.
__device__ __forceinline__ bool exp_within(float x, int lo, int hi) {
	const uint32_t e = (__float_as_uint(x) >> 23) & 0xFFu;
	return e >= uint32_t(127 + lo) && e <= uint32_t(127 + hi);
}
// A coordinate is "tame" when it is 0 or 2^-70 <= |c| < 2^39: the difference of two tame
// coordinates is then 0 or at least 2^-94 and at most 2^40 in magnitude.
__device__ __forceinline__ bool tame_coord(float c) { return c == 0.0f || exp_within(c, -70, 38); }
// Ray-side half of the range proof: direction components in [2^-40, 2^11) and tame origin.
// With a tame tree (MeshRec::tame / SceneRec::top_tame) every numerator (plane - origin) is
// 0 or in [2^-94, 2^40], every quotient and every intermediate of div_by is a normal number
// (or an exact zero), and the remainders are exactly representable: the division expansion's
// v_div_scale would not scale and v_div_fixup would not alter the result.
__device__ __forceinline__ bool ray_in_fast_range(const Ray3& r) {
	return exp_within(r.dx, -40, 10) && exp_within(r.dy, -40, 10) && exp_within(r.dz, -40, 10) && tame_coord(r.ox) && tame_coord(r.oy) &&
	       tame_coord(r.oz);
}

// The same with direction components that are exactly zero allowed (orthographic views, straight-down shadow rays): the
// ZERO instances of the tests below treat such an axis as the reference does (bbox.h:98-120).  A zero vector is no direction.
__device__ __forceinline__ bool dir_component_ok_z(float d) { return d == 0.0f || exp_within(d, -40, 10); }
__device__ __forceinline__ bool ray_in_fast_range_z(const Ray3& r) {
	return dir_component_ok_z(r.dx) && dir_component_ok_z(r.dy) && dir_component_ok_z(r.dz) && !(r.dx == 0.0f && r.dy == 0.0f && r.dz == 0.0f) &&
	       tame_coord(r.ox) && tame_coord(r.oy) && tame_coord(r.oz);
}
__device__ __forceinline__ bool ray_has_zero_component(const Ray3& r) { return r.dx == 0.0f || r.dy == 0.0f || r.dz == 0.0f; }

struct RayInv {
	float yx, yy, yz;   // refined reciprocals of the direction (FAST only)
};
// What RayInv holds for a direction component: the refined reciprocal, or +-inf for +-0 (v_rcp_f32's answer), with
// which `numerator * y` is bit for bit the IEEE quotient numerator / +-0 (+-inf by the sign rules, NaN for 0 / 0).
__device__ __forceinline__ float slab_rcp(float d) { return d == 0.0f ? __builtin_amdgcn_rcpf(d) : refined_rcp(d); }

// BBox::hit(ray, times = (0, inf)) -- lib/bbox.h:81-154.  Returns the accept/reject decision;
// `tnear` receives min_intersect (bbox.h:137), used only to order and cull in CULLED mode.
// ZERO (with FAST): direction components may be exactly zero; `inv` then holds slab_rcp() of them.
template<bool FAST, bool ZERO = false>
__device__ __forceinline__ bool slab_test(float minx, float miny, float minz, float maxx, float maxy, float maxz, const Ray3& r,
                                          const RayInv& inv, float& tnear) {
	float tx0, tx1, ty0, ty1, tz0, tz1;
	if (FAST && ZERO) {
		if (r.dx == 0.0f && (r.ox < minx || r.ox > maxx)) return false;   // bbox.h:98-106
		if (r.dy == 0.0f && (r.oy < miny || r.oy > maxy)) return false;
		if (r.dz == 0.0f && (r.oz < minz || r.oz > maxz)) return false;
		// a zero axis: (plane - origin) / +-0 == (plane - origin) * +-inf; the others as below
		tx0 = r.dx == 0.0f ? (minx - r.ox) * inv.yx : div_by(minx - r.ox, r.dx, inv.yx), tx1 = r.dx == 0.0f ? (maxx - r.ox) * inv.yx : div_by(maxx - r.ox, r.dx, inv.yx);
		ty0 = r.dy == 0.0f ? (miny - r.oy) * inv.yy : div_by(miny - r.oy, r.dy, inv.yy), ty1 = r.dy == 0.0f ? (maxy - r.oy) * inv.yy : div_by(maxy - r.oy, r.dy, inv.yy);
		tz0 = r.dz == 0.0f ? (minz - r.oz) * inv.yz : div_by(minz - r.oz, r.dz, inv.yz), tz1 = r.dz == 0.0f ? (maxz - r.oz) * inv.yz : div_by(maxz - r.oz, r.dz, inv.yz);
	} else if (FAST) {
		// direction components are non-zero here, so the dir == 0 rejects (bbox.h:98-106) cannot fire
		tx0 = div_by(minx - r.ox, r.dx, inv.yx), tx1 = div_by(maxx - r.ox, r.dx, inv.yx);
		ty0 = div_by(miny - r.oy, r.dy, inv.yy), ty1 = div_by(maxy - r.oy, r.dy, inv.yy);
		tz0 = div_by(minz - r.oz, r.dz, inv.yz), tz1 = div_by(maxz - r.oz, r.dz, inv.yz);
	} else {
		if (r.dx == 0.0f && (r.ox < minx || r.ox > maxx)) return false;   // bbox.h:98-106
		if (r.dy == 0.0f && (r.oy < miny || r.oy > maxy)) return false;
		if (r.dz == 0.0f && (r.oz < minz || r.oz > maxz)) return false;
		tx0 = (minx - r.ox) / r.dx, tx1 = (maxx - r.ox) / r.dx;           // bbox.h:110-120
		ty0 = (miny - r.oy) / r.dy, ty1 = (maxy - r.oy) / r.dy;
		tz0 = (minz - r.oz) / r.dz, tz1 = (maxz - r.oz) / r.dz;
	}
	if (tx0 > tx1) { float s = tx0; tx0 = tx1; tx1 = s; }
	if (ty0 > ty1) { float s = ty0; ty0 = ty1; ty1 = s; }
	if (tz0 > tz1) { float s = tz0; tz0 = tz1; tz1 = s; }
	if (tx0 > ty1 || tx1 < ty0) return false;                          // bbox.h:124
	float lo = (tx0 > ty0) ? tx0 : ty0;                                // bbox.h:128-129
	float hi = (tx1 < ty1) ? tx1 : ty1;
	if (lo > tz1 || hi < tz0) return false;                            // bbox.h:133
	lo = (lo > tz0) ? lo : tz0;                                        // bbox.h:137-138
	hi = (hi < tz1) ? hi : tz1;
	if (hi < 0.0f) return false;                                       // bbox.h:142 with times = (0, inf): lo > inf is never true
	tnear = lo;
	return true;
}

// The same decision from approximate entry/exit distances, for rays and trees in the exact-division
// range (no zero direction component, no NaN or infinity anywhere).
//
// BBox::hit accepts exactly when LO <= HI and HI >= 0, with LO the largest of the three per-axis
// entry quotients and HI the smallest of the three exit quotients: the per-axis swap (bbox.h:110-120)
// makes entry <= exit on every axis, and the pairwise rejects (bbox.h:124,133) together say that
// every entry is <= every exit.  Here each quotient (plane - origin) / dir is replaced by
// (plane - origin) * y, y = refined_rcp(dir): relative error below 2^-22, same sign, zero exactly
// when the numerator is zero.  So HI >= 0 is decided exactly, and LO <= HI is decided exactly
// whenever the two approximations are further apart than 2^-20 of their magnitudes; otherwise the
// lane reports `ambiguous` and the caller runs the exact test.  `tnear` is the approximate LO
// (ordering and culling only, inside the guard band of cull_limit).
//
// ZERO: direction components may be exactly zero (inv holds slab_rcp() of them).  On such an axis the products below
// ARE the reference's quotients: -inf / +inf when the origin lies strictly between the two planes (the axis then
// constrains nothing, in the reference's compare-and-select sequence as in the min/max merges here), two infinities
// of one sign when it lies outside (the reference rejects at once, bbox.h:98-106; here LO = +inf or HI = -inf, and
// since some other axis is finite that rejects too), NaN when it lies ON a plane (0 / 0: what the reference's
// sequence does with a NaN depends on its position, so the lane reports `ambiguous`).  An infinite LO or HI is a
// certain reject, not a close call.
constexpr float SLAB_AMBIGUITY = 0x1p-20f;
template<bool ZERO = false>
__device__ __forceinline__ bool slab_test_approx(float minx, float miny, float minz, float maxx, float maxy, float maxz, const Ray3& r, const RayInv& inv,
                                                 float& tnear, bool& ambiguous) {
	const float ax = (minx - r.ox) * inv.yx, bx = (maxx - r.ox) * inv.yx;
	const float ay = (miny - r.oy) * inv.yy, by = (maxy - r.oy) * inv.yy;
	const float az = (minz - r.oz) * inv.yz, bz = (maxz - r.oz) * inv.yz;
	const float lo = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
	const float hi = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
	const float spread = fabsf(lo) + fabsf(hi);
	ambiguous = fabsf(hi - lo) <= SLAB_AMBIGUITY * spread;
	if (ZERO) {
		ambiguous = ambiguous && spread < INFINITY;
		ambiguous = ambiguous || ax != ax || bx != bx || ay != ay || by != by || az != az || bz != bz;
	}
	tnear = lo;
	return lo <= hi && hi >= 0.0f;
}

// Triangle::hit -- pathtracer/tri_mesh.cpp:46-111, on triangle data already in registers
// (a = v0, b = e1, c = e2).
__device__ __forceinline__ bool triangle_test(const float4 a, const float4 b, const float4 c, const Ray3& r, float& distance, float& t_out,
                                              float& u_out, float& v_out) {
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

// Per-lane traversal stacks.
//
// Stack: everything in the lane's scratch memory.
//
// LdsStack: the first LDS_SLOTS entries of a lane's stack live in LDS (laid out
// [entry][thread], so the 64 lanes of a wave always touch 64 different banks whatever their
// stack depths are), deeper entries in a scratch Stack it points to.  A near-first traversal
// rarely holds more than a dozen pending subtrees, so scratch traffic (HBM write bandwidth on
// every push, and a vector-memory instruction each way) all but disappears.  The LDS pointers
// sit in their own small object, apart from the dynamically indexed arrays, so the compiler
// keeps them in registers.
struct Stack {
	uint32_t ref[STACK_CAP];
	float tnear[STACK_CAP];
	__device__ __forceinline__ void push(int sp, uint32_t r, float t) { ref[sp] = r, tnear[sp] = t; }
	__device__ __forceinline__ void push(int sp, uint32_t r) { ref[sp] = r; }
	__device__ __forceinline__ uint32_t get_ref(int sp) const { return ref[sp]; }
	__device__ __forceinline__ float get_tnear(int sp) const { return tnear[sp]; }
};

template<int LDS_SLOTS, int STRIDE>
struct LdsStack {
	// LDS pointers carry their address space in the type: through a generic pointer the reads below
	// become flat loads, which go through the vector-memory path even though they land in LDS.
	typedef __attribute__((address_space(3))) uint32_t LdsU32;
	typedef __attribute__((address_space(3))) float LdsF32;
	LdsU32* lds_ref;    // &lds_ref_array[threadIdx.x]
	LdsF32* lds_tnear;
	__device__ __forceinline__ LdsStack(uint32_t* ref, float* tnear, Stack* sp) : lds_ref((LdsU32*)ref), lds_tnear((LdsF32*)tnear), spill(sp) {}
	Stack* spill;
	__device__ __forceinline__ void push(int sp, uint32_t r, float t) {
		if (sp < LDS_SLOTS) lds_ref[sp * STRIDE] = r, lds_tnear[sp * STRIDE] = t;
		else spill->push(sp - LDS_SLOTS, r, t);
	}
	__device__ __forceinline__ void push(int sp, uint32_t r) {
		if (sp < LDS_SLOTS) lds_ref[sp * STRIDE] = r;
		else spill->push(sp - LDS_SLOTS, r);
	}
	// The LDS entry is read whatever sp is (clamped, through a volatile pointer) and replaced under a
	// branch when the entry lives in scratch: left free, the compiler picks between the two ADDRESSES and
	// issues one flat load.
	__device__ __forceinline__ uint32_t get_ref(int sp) const {
		uint32_t v = *(const volatile LdsU32*)(lds_ref + (sp < LDS_SLOTS ? sp : LDS_SLOTS - 1) * STRIDE);
		if (__builtin_expect(sp >= LDS_SLOTS, 0)) v = spill->get_ref(sp - LDS_SLOTS);
		return v;
	}
	__device__ __forceinline__ float get_tnear(int sp) const {
		float v = *(const volatile LdsF32*)(lds_tnear + (sp < LDS_SLOTS ? sp : LDS_SLOTS - 1) * STRIDE);
		if (__builtin_expect(sp >= LDS_SLOTS, 0)) v = spill->get_tnear(sp - LDS_SLOTS);
		return v;
	}
};

// ---- traversal steps ------------------------------------------------------------------------
// One traversal of PT::BVH<P>::hit / hitRecursive (bvh.cpp:246-375) over a flattened tree is a
// sequence of these steps on the state (ref, sp):
//   node_step  `ref` is an interior node: test both children's boxes, go to one, maybe push
//              the other, or pop
//   pop_ref    take the next pending subtree off the stack (or REF_DONE when none is left)
// run_hit / run_dist are the caller's running Trace::hit and Trace::distance.
//
// REFERENCE: depth-first, left subtree completely before right (both entry times are 0 in
// bvh.cpp:326-342); a sibling is skipped only when the running hit has a NaN distance
// (bvh.cpp:337-339).  CULLED: nearer child first; a subtree whose entry distance lies beyond
// the running closest distance (closest hit) or beyond dist_bounds.y (occlusion) by more
// than the guard band is skipped, unless its reference is marked REF_NOCULL.
template<bool ANY>
__device__ __forceinline__ float cull_limit(const Ray3& r, bool run_hit, float run_dist) {
	if (ANY) return r.tmax * CULL_SCALE + CULL_BIAS;
	return run_hit ? run_dist * CULL_SCALE + CULL_BIAS : INFINITY;
}

template<int MODE, bool ANY, typename StackType>
__device__ __forceinline__ void pop_ref(const Ray3& r, StackType& st, const int base, int& sp, uint32_t& ref, bool run_hit, float run_dist) {
	for (;;) {
		if (sp == base) {
			ref = REF_DONE;
			return;
		}
		--sp;
		ref = st.get_ref(sp);
		if (MODE == MODE_REFERENCE) {
			if (run_hit && !(run_dist >= 0.0f)) continue;   // bvh.cpp:337-339
			return;
		} else {
			if (!ANY && !(ref & REF_NOCULL) && st.get_tnear(sp) > cull_limit<ANY>(r, run_hit, run_dist)) continue;
			return;
		}
	}
}

// The decision half of a traversal step, on node data already in registers: n0, n1, n2 are the node's
// first three 16-byte pieces (left box, right box), kids its two child references.
// DEFER_POP: when neither child is taken, leave REF_NONE in `ref` instead of popping, so that a caller
// with other reasons to pop (a leaf it parks) runs one pop loop for all of them.
// APPROX: decide with slab_test_approx (exact test where it cannot tell).  The culled order always does (it also
// wants the entry distances); the reference order needs the decisions only, and the replenishing kernel asks for the
// cheaper form while the one-ray-per-lane kernels keep the plain exact test.
template<int MODE, bool ANY, bool FAST, bool DEFER_POP = false, bool ZERO = false, bool APPROX = (MODE == MODE_CULLED), typename StackType>
__device__ __forceinline__ void node_decide(const float4 n0, const float4 n1, const float4 n2, const uint2 kids, const Ray3& r, const RayInv& inv, StackType& st,
                                            const int base, int& sp, uint32_t& ref, bool run_hit, float run_dist) {
	float tl, tr;
	bool hl, hr;
	if (FAST && APPROX) {
		bool al, ar;
		hl = slab_test_approx<ZERO>(PTX_NODE_LEFT_BOX(n0, n1, n2), r, inv, tl, al);
		hr = slab_test_approx<ZERO>(PTX_NODE_RIGHT_BOX(n0, n1, n2), r, inv, tr, ar);
		if (__builtin_expect(__any(al || ar), 0)) {   // a ray grazing a box within rounding: decide it exactly
			if (al) hl = slab_test<true, ZERO>(PTX_NODE_LEFT_BOX(n0, n1, n2), r, inv, tl);
			if (ar) hr = slab_test<true, ZERO>(PTX_NODE_RIGHT_BOX(n0, n1, n2), r, inv, tr);
		}
	} else {
		hl = slab_test<FAST, FAST && ZERO>(PTX_NODE_LEFT_BOX(n0, n1, n2), r, inv, tl);
		hr = slab_test<FAST, FAST && ZERO>(PTX_NODE_RIGHT_BOX(n0, n1, n2), r, inv, tr);
	}
	if (MODE == MODE_REFERENCE) {
		if (hl) {
			if (hr) st.push(sp++, kids.y);
			ref = kids.x;
		} else if (hr && !(run_hit && !(run_dist >= 0.0f))) {
			ref = kids.y;
		} else {
			pop_ref<MODE, ANY>(r, st, base, sp, ref, run_hit, run_dist);
		}
	} else {
		const float limit = cull_limit<ANY>(r, run_hit, run_dist);
		const bool gl = hl && ((kids.x & REF_NOCULL) || !(tl > limit)), gr = hr && ((kids.y & REF_NOCULL) || !(tr > limit));
		if (gl && gr) {
			const bool left_first = !(tr < tl);
			st.push(sp, left_first ? kids.y : kids.x, left_first ? tr : tl);
			sp++;
			ref = left_first ? kids.x : kids.y;
		} else if (gl) {
			ref = kids.x;
		} else if (gr) {
			ref = kids.y;
		} else if (DEFER_POP) {
			ref = REF_NONE;
		} else {
			pop_ref<MODE, ANY>(r, st, base, sp, ref, run_hit, run_dist);
		}
	}
}

// The culled order's step on a four-wide node (device_types.h: Node4), for rays and trees in the exact-division
// range.  Every child's accept/reject decision is the exact one (approximate test, exact test where the
// approximation cannot tell: the node's half is then read again rather than kept in registers); the accepted
// children are visited nearest first, the others pushed farthest first.  Leaves REF_NONE in `ref` when no child
// is taken (the caller pops).
template<bool ANY, typename StackType>
__device__ __forceinline__ void node4_decide(const Node4* __restrict__ node, const Ray3& r, const RayInv& inv, StackType& st, int& sp, uint32_t& ref, bool run_hit,
                                             float run_dist) {
	const float limit = cull_limit<ANY>(r, run_hit, run_dist);
	float key[4];
	uint32_t val[4];
	int n = 0;
#pragma unroll
	for (int half = 0; half < 2; half++) {
		const float4* p = reinterpret_cast<const float4*>(node->half + half);
		const float4 n0 = p[0], n1 = p[1], n2 = p[2];
		const uint2 kids = *reinterpret_cast<const uint2*>(p + 3);
		float tl, tr;
		bool al, ar;
		bool hl = slab_test_approx(PTX_NODE_LEFT_BOX(n0, n1, n2), r, inv, tl, al);
		bool hr = slab_test_approx(PTX_NODE_RIGHT_BOX(n0, n1, n2), r, inv, tr, ar);
		if (__builtin_expect(__any(al || ar), 0)) {   // a ray grazing a box within rounding: decide it exactly
			const volatile float4* q = reinterpret_cast<const volatile float4*>(node->half + half);
			const float4 m0 = const_cast<const float4&>(q[0]), m1 = const_cast<const float4&>(q[1]), m2 = const_cast<const float4&>(q[2]);
			if (al) hl = slab_test<true>(PTX_NODE_LEFT_BOX(m0, m1, m2), r, inv, tl);
			if (ar) hr = slab_test<true>(PTX_NODE_RIGHT_BOX(m0, m1, m2), r, inv, tr);
		}
		const bool gl = hl && kids.x != REF_NONE4 && ((kids.x & REF_NOCULL) || !(tl > limit));
		const bool gr = hr && kids.y != REF_NONE4 && ((kids.y & REF_NOCULL) || !(tr > limit));
		key[2 * half] = gl ? tl : INFINITY, val[2 * half] = gl ? kids.x : REF_NONE4;
		key[2 * half + 1] = gr ? tr : INFINITY, val[2 * half + 1] = gr ? kids.y : REF_NONE4;
		n += (gl ? 1 : 0) + (gr ? 1 : 0);
	}
	// sorting network on four (key, value) pairs; rejected entries are told apart by their value, so an accepted
	// child whose entry distance is +inf cannot be lost among them
	auto cswap = [&](int a, int b) {
		const bool sw = (val[a] == REF_NONE4 && val[b] != REF_NONE4) || (val[a] != REF_NONE4 && val[b] != REF_NONE4 && key[b] < key[a]);
		const float ka = sw ? key[b] : key[a], kb = sw ? key[a] : key[b];
		const uint32_t va = sw ? val[b] : val[a], vb = sw ? val[a] : val[b];
		key[a] = ka, key[b] = kb, val[a] = va, val[b] = vb;
	};
	cswap(0, 1), cswap(2, 3), cswap(0, 2), cswap(1, 3), cswap(1, 2);
	if (n > 3) st.push(sp++, val[3], key[3]);
	if (n > 2) st.push(sp++, val[2], key[2]);
	if (n > 1) st.push(sp++, val[1], key[1]);
	ref = n > 0 ? val[0] : REF_NONE;
}

// The four 16-byte pieces of a node record, requested back to back and waited for TOGETHER.  Left to the compiler
// the loads are followed by one wait per piece with the first arithmetic in between (a wave that wakes up four
// times per node): measured on the 1M-triangle soup that form walks 16.7 M rays in 13.3 ms, this one in 11.8 ms.
// The block below is gfx9-family assembly (global_load_dwordx4 ... off, ds_read_b128, s_waitcnt vmcnt / lgkmcnt): it is
// the default on those targets only; any other target (and the host pass) takes the plain C++ form.
#ifndef PTX_NODE_FETCH
#if defined(__gfx950__) || defined(__gfx942__) || defined(__gfx940__) || defined(__gfx941__) || defined(__gfx90a__) || defined(__gfx908__) || defined(__gfx906__) || defined(__gfx900__)
#define PTX_NODE_FETCH 1
#else
#define PTX_NODE_FETCH 0
#endif
#endif
__device__ __forceinline__ void fetch_node(const Node2* __restrict__ node, float4& n0, float4& n1, float4& n2, uint2& kids) {
#if PTX_NODE_FETCH
	typedef float V4 __attribute__((ext_vector_type(4)));
	V4 q0, q1, q2, q3;
	asm volatile(
	    "global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\tglobal_load_dwordx4 %2, %4, off offset:32\n\t"
	    "global_load_dwordx4 %3, %4, off offset:48\n\ts_waitcnt vmcnt(0)"
	    : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
	    : "v"(node)
	    : "memory");
	n0 = make_float4(q0.x, q0.y, q0.z, q0.w), n1 = make_float4(q1.x, q1.y, q1.z, q1.w), n2 = make_float4(q2.x, q2.y, q2.z, q2.w);
	kids = make_uint2(__float_as_uint(q3.x), __float_as_uint(q3.y));
#else
	const float4* p = reinterpret_cast<const float4*>(node);
	n0 = p[0], n1 = p[1], n2 = p[2];
	kids = *reinterpret_cast<const uint2*>(p + 3);
#endif
}

// The same record out of an LDS copy (address-space-typed reads: through a generic pointer they would be flat loads).
__device__ __forceinline__ void fetch_node_lds(const float4* lds_nodes, uint32_t index, float4& n0, float4& n1, float4& n2, uint2& kids) {
	typedef float V4 __attribute__((ext_vector_type(4)));
	typedef __attribute__((address_space(3))) V4 LdsV4;
	const LdsV4* q = (const LdsV4*)lds_nodes + index * 4u;
#if PTX_NODE_FETCH
	V4 q0, q1, q2, q3;
	asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
	             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
	             : "v"(q)
	             : "memory");
#else
	const V4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
#endif
	n0 = make_float4(q0.x, q0.y, q0.z, q0.w), n1 = make_float4(q1.x, q1.y, q1.z, q1.w), n2 = make_float4(q2.x, q2.y, q2.z, q2.w);
	kids = make_uint2(__float_as_uint(q3.x), __float_as_uint(q3.y));
}

template<int MODE, bool ANY, bool FAST, bool STATS, bool ZERO = false, bool APPROX = (MODE == MODE_CULLED), typename StackType>
__device__ __forceinline__ void node_step(const Node2* __restrict__ nodes, const Ray3& r, const RayInv& inv, StackType& st, const int base, int& sp,
                                          uint32_t& ref, bool run_hit, float run_dist, TraverseStats* stats) {
	float4 n0, n1, n2;
	uint2 kids;
	fetch_node(nodes + (ref & REF_INDEX_MASK), n0, n1, n2, kids);
	if (STATS) stats->nodes++;
	if (STATS) stats->max_sp = max(stats->max_sp, uint32_t(sp - base));
	node_decide<MODE, ANY, FAST, false, ZERO, APPROX>(n0, n1, n2, kids, r, inv, st, base, sp, ref, run_hit, run_dist);
}

// A whole traversal: walk interior nodes until every lane of the wave holds a leaf (or is
// finished), then intersect leaves -- two nested phases, so the box-test code and the
// primitive-test code each run with as many lanes active as possible.
//   leaf(first, count) -> bool   tests primitives [first, first+count) in array order, folds them
//                                into the caller's running hit, returns true to stop (ANY mode)
template<int MODE, bool ANY, bool FAST, bool STATS, typename Leaf, typename StackType>
__device__ __forceinline__ void traverse_tree(const Node2* __restrict__ nodes, uint32_t root_ref, const Ray3& r, const RayInv& inv, StackType& st,
                                              const int base, Leaf&& leaf, const bool& run_hit, const float& run_dist, TraverseStats* stats) {
	int sp = base;
	uint32_t ref = root_ref;
	for (;;) {
		while (!(ref & REF_LEAF)) node_step<MODE, ANY, FAST, STATS>(nodes, r, inv, st, base, sp, ref, run_hit, run_dist, stats);
		if (ref == REF_DONE) return;
		if (STATS) stats->leaves++;
		if (leaf((ref & REF_INDEX_MASK) >> 4, ref & 15u)) return;
		pop_ref<MODE, ANY>(r, st, base, sp, ref, run_hit, run_dist);
	}
}

// Triangles [first, first+count) of a mesh folded into the running hit, in array order.
// The next triangle's three 16-byte loads are in flight while the current one is tested, so a
// leaf costs about one memory round trip, not one per triangle, at the price of one extra
// triangle's worth of registers.
template<int MODE, bool ANY, bool STATS>
__device__ __forceinline__ bool mesh_leaf(const TriGeom* __restrict__ tris, uint32_t first, uint32_t count, const Ray3& r, MeshHit& cur,
                                          TraverseStats* stats) {
	if (count == 0) return false;
	const float4* p = reinterpret_cast<const float4*>(tris + first);
	float4 a = p[0], b = p[1], c = p[2];
	for (uint32_t k = 0; k < count; k++) {
		float4 na = a, nb = b, nc = c;
		if (k + 1 < count) na = p[3 * (k + 1)], nb = p[3 * (k + 1) + 1], nc = p[3 * (k + 1) + 2];
		const uint32_t i = first + k;
		float d, t, u, v;
		if (STATS) stats->tris++;
		if (triangle_test(a, b, c, r, d, t, u, v)) {
			// Trace::min(cur, candidate), trace.h:28-36: the candidate replaces the running hit unless
			// the running hit is strictly closer.  In array order that leaves, among equal distances,
			// the largest array position; the culled mode visits in another order and asks for that
			// outcome explicitly.
			const bool keep = MODE == MODE_REFERENCE ? (cur.hit && cur.distance < d)
			                                         : (cur.hit && !(d < cur.distance || (d == cur.distance && i > cur.prim)));
			if (!keep) cur.hit = true, cur.distance = d, cur.t = t, cur.prim = i, cur.u = u, cur.v = v;
			if (ANY) return true;
		} else if (!cur.hit) {
			cur.distance = 0.0f;   // Trace::min of two misses is a default Trace (trace.h:35)
		}
		a = na, b = nb, c = nc;
	}
	return false;
}

// PT::BVH<Triangle>::hit (bvh.cpp:246-270) or List<Triangle>::hit (list.h:25-32) for one mesh.
template<int MODE, bool ANY, bool STATS, typename StackType>
__device__ __forceinline__ void traverse_mesh(const SceneRec& sc, const MeshRec& mesh, const Ray3& r, StackType& st, int base, MeshHit& cur,
                                              TraverseStats* stats) {
	const TriGeom* __restrict__ tris = sc.tris + mesh.tri_base;
	auto leaf = [&](uint32_t first, uint32_t count) -> bool { return mesh_leaf<MODE, ANY, STATS>(tris, first, count, r, cur, stats); };
	if (mesh.is_list) {
		// List::hit starts from a default Trace (hit = false, distance = 0) and never tests a box.
		cur.hit = false, cur.distance = 0.0f, cur.prim = 0xFFFFFFFFu;
		for (uint32_t first = 0; first < mesh.n_tris; first += 8)
			if (leaf(first, min(8u, mesh.n_tris - first))) return;
		return;
	}
	// BVH::hit seeds the running Trace with hit = false, distance = inf (bvh.cpp:257-259).
	cur.hit = false, cur.distance = INFINITY, cur.prim = 0xFFFFFFFFu;
	const Node2* __restrict__ nodes = sc.nodes + mesh.node_base;
	RayInv inv{};
	float tn;
	if (mesh.tame && ray_in_fast_range(r)) {
		inv.yx = refined_rcp(r.dx), inv.yy = refined_rcp(r.dy), inv.yz = refined_rcp(r.dz);
		if (!slab_test<true>(mesh.root_min[0], mesh.root_min[1], mesh.root_min[2], mesh.root_max[0], mesh.root_max[1], mesh.root_max[2], r, inv, tn)) return;
		traverse_tree<MODE, ANY, true, STATS>(nodes, mesh.root_ref, r, inv, st, base, leaf, cur.hit, cur.distance, stats);
	} else {
		if (!slab_test<false>(mesh.root_min[0], mesh.root_min[1], mesh.root_min[2], mesh.root_max[0], mesh.root_max[1], mesh.root_max[2], r, inv, tn)) return;
		traverse_tree<MODE, ANY, false, STATS>(nodes, mesh.root_ref, r, inv, st, base, leaf, cur.hit, cur.distance, stats);
	}
}

// ---- Mat4 * Vec3 forms (lib/mat4.h:175-186) ---------------------------------------------
__device__ __forceinline__ void mat_point(const float* __restrict__ m, float x, float y, float z, float& ox, float& oy, float& oz) {
	// v[0]*cols[0] + v[1]*cols[1] + v[2]*cols[2] + v[3]*cols[3] with v[3] = 1, then project()
	const float rx = ((x * m[0] + y * m[4]) + z * m[8]) + 1.0f * m[12];
	const float ry = ((x * m[1] + y * m[5]) + z * m[9]) + 1.0f * m[13];
	const float rz = ((x * m[2] + y * m[6]) + z * m[10]) + 1.0f * m[14];
	const float rw = ((x * m[3] + y * m[7]) + z * m[11]) + 1.0f * m[15];
	// project(): divide by w.  An affine matrix gives w == 1 exactly, and x / 1 == x in IEEE arithmetic,
	// so the three divisions are skipped on that condition without changing a bit.
	ox = rx, oy = ry, oz = rz;
	if (rw != 1.0f) ox = rx / rw, oy = ry / rw, oz = rz / rw;
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
	if (w1 && w2) t_out = (t2 < t1) ? t2 : t1;   // std::min(t1, t2)
	else if (w1) t_out = t1;
	else if (w2) t_out = t2;
	else return false;
	return true;
}

struct SceneHit {
	bool hit;
	float distance;    // world-space Trace::distance
	float t;           // ray parameter in the hit instance's object space
	uint32_t prim, inst;
	float u, v;
};

// Instance::hit (pathtracer/instance.h:32-40): ray into object space, geometry hit, Trace
// back to world space with distance recomputed from the transformed points (trace.h:38-43).
template<int MODE, bool ANY, bool STATS>
__device__ __forceinline__ bool instance_hit(const SceneRec& sc, uint32_t inst_index, const Ray3& world, Stack& st, int base, SceneHit& out,
                                             TraverseStats* stats) {
	const InstRec& in = sc.insts[inst_index];
	Ray3 r = world;
	if (in.has_transform) r = ray_transform(in.iT, world);
	float t, dist, u = 0.0f, v = 0.0f;
	uint32_t prim = 0;
	if (in.mesh >= 0) {
		MeshHit mh;
		traverse_mesh<MODE, ANY, STATS>(sc, sc.meshes[in.mesh], r, st, base, mh, stats);
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
	out.hit = true, out.distance = dist, out.t = t, out.prim = prim, out.inst = inst_index, out.u = u, out.v = v;
	return true;
}

// Where the culled order cannot stand in for the reference's: Trace::min keeps its left argument only
// when `l.distance < r.distance` (trace.h:28-36), so once a distance is NaN the winner is whichever hit
// was folded in last -- a matter of test order, which only the reference order reproduces.  NaN
// distances come from rays with a NaN or infinite component (or a direction too short to normalise,
// in an instance); such rays are rare and are walked again in the reference order.
__device__ __forceinline__ bool ray_is_finite(const Ray3& r) {
	const float s = (r.ox + r.oy + r.oz) + (r.dx + r.dy + r.dz);   // NaN or inf anywhere makes the sum NaN or inf (inf - inf: NaN)
	return fabsf(s) <= 3.402823466e+38f && r.tmin == r.tmin && r.tmax == r.tmax;
}
__device__ __forceinline__ bool needs_reference_order(const Ray3& r, bool hit, float distance) { return !ray_is_finite(r) || (hit && distance != distance); }

// Aggregate::hit over BVH<Instance> / List<Instance> (aggregate.h:41-43), or Tri_Mesh::hit
// when the scene is a single mesh.
// MESH_ONLY is a compile-time promise that sc.mesh_only is set: the two-level code is then
// not even compiled into the kernel, which keeps its register count (and so occupancy) down.
template<int MODE, bool ANY, bool STATS = false, bool MESH_ONLY = false>
__device__ __forceinline__ void traverse_scene(const SceneRec& sc, const Ray3& r, Stack& st, SceneHit& cur, TraverseStats* stats = nullptr) {
	if (MESH_ONLY || sc.mesh_only) {
		MeshHit mh;
		traverse_mesh<MODE, ANY, STATS>(sc, sc.meshes[0], r, st, 0, mh, stats);
		cur.hit = mh.hit, cur.distance = mh.distance, cur.t = mh.t, cur.prim = mh.prim, cur.inst = 0, cur.u = mh.u, cur.v = mh.v;
		return;
	}
	if (MESH_ONLY) return;
	// The top level owns the first top_stack_slots stack slots; instances use the rest.
	const int inst_base = int(sc.top_stack_slots);
	auto leaf = [&](uint32_t first, uint32_t count) -> bool {
		for (uint32_t i = first; i < first + count; i++) {
			SceneHit h;
			if (instance_hit<MODE, ANY, STATS>(sc, i, r, st, inst_base, h, stats)) {
				const bool keep = MODE == MODE_REFERENCE ? (cur.hit && cur.distance < h.distance)
				                                         : (cur.hit && !(h.distance < cur.distance || (h.distance == cur.distance && h.inst > cur.inst)));
				if (!keep) cur = h;
				if (ANY) return true;
			} else if (!cur.hit) {
				cur.distance = 0.0f;
			}
		}
		return false;
	};
	if (sc.top_is_list) {
		cur.hit = false, cur.distance = 0.0f, cur.prim = 0xFFFFFFFFu, cur.inst = 0xFFFFFFFFu;
		for (uint32_t i = 0; i < sc.n_insts; i++)
			if (leaf(i, 1)) return;
		return;
	}
	cur.hit = false, cur.distance = INFINITY, cur.prim = 0xFFFFFFFFu, cur.inst = 0xFFFFFFFFu;
	const Node2* __restrict__ nodes = sc.nodes + sc.top_node_base;
	RayInv inv{};
	float tn;
	if (sc.top_tame && ray_in_fast_range(r)) {
		inv.yx = refined_rcp(r.dx), inv.yy = refined_rcp(r.dy), inv.yz = refined_rcp(r.dz);
		if (!slab_test<true>(sc.top_min[0], sc.top_min[1], sc.top_min[2], sc.top_max[0], sc.top_max[1], sc.top_max[2], r, inv, tn)) return;
		traverse_tree<MODE, ANY, true, STATS>(nodes, sc.top_ref, r, inv, st, 0, leaf, cur.hit, cur.distance, stats);
	} else {
		if (!slab_test<false>(sc.top_min[0], sc.top_min[1], sc.top_min[2], sc.top_max[0], sc.top_max[1], sc.top_max[2], r, inv, tn)) return;
		traverse_tree<MODE, ANY, false, STATS>(nodes, sc.top_ref, r, inv, st, 0, leaf, cur.hit, cur.distance, stats);
	}
}

}  // namespace ptx

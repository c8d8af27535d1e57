// oracle/ref_cpu.h -- CPU restatement of the reference path tracer's hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing the product ships may include, link or call
// this file: it is the checker used by tests/, by __graft_entry__.smoke() and by
// the cpu_baseline leg of bench.py.  It is written for clarity and fidelity (same
// recursion, same AoS layouts, same operation order as the reference), not speed.
//
// Pinning: this restatement is compared bit-for-bit (hit distance bits, primitive
// order, node arrays, BSDF known answers) with the real reference sources compiled
// by oracle/Makefile into oracle/_ref/libptref.so, and with the golden fixtures
// under tests/golden/ that were generated from that library.  Pathtracer::trace and
// the tile loop (pathtracer.cpp) and Camera::sample_ray (camera.cpp) cannot be
// compiled in this image (they include SDL.h, which is absent), so render_* below
// is "parity unpinned" as a whole; every function it calls is pinned individually.
//
// Build with:  g++ -std=c++17 -O2 -ffp-contract=off   (x86-64: no FMA contraction,
// IEEE divide/sqrt, denormals preserved -- the reference's Linux flags).
//
// All citations are relative to the reference checkout (src/...).
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <random>
#include <utility>
#include <vector>

namespace orc {

constexpr float EPS_F = 0.00001f;                                       // lib/mathlib.h:16
constexpr float PI_F = 3.14159265358979323846264338327950288f;          // lib/mathlib.h:17
constexpr float INF_F = std::numeric_limits<float>::infinity();

// ---------------------------------------------------------------- vectors (lib/vec2.h, vec3.h, vec4.h)
struct V2 {
	float x = 0.0f, y = 0.0f;
};
inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V2 operator*(float s, V2 v) { return {v.x * s, v.y * s}; }
inline V2 operator*(V2 v, float s) { return {v.x * s, v.y * s}; }

struct V3 {
	float x = 0.0f, y = 0.0f, z = 0.0f;
	float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
	float norm_squared() const { return x * x + y * y + z * z; }      // vec3.h:155
	float norm() const { return std::sqrt(norm_squared()); }           // vec3.h:158
	V3 unit() const {                                                   // vec3.h:150
		float n = norm();
		return {x / n, y / n, z / n};
	}
	bool valid() const { return std::isfinite(x) && std::isfinite(y) && std::isfinite(z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline float dot(V3 l, V3 r) { return l.x * r.x + l.y * r.y + l.z * r.z; }   // vec3.h:215
inline V3 cross(V3 l, V3 r) {                                                  // vec3.h:220
	return {l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x};
}
inline V3 hmin(V3 l, V3 r) { return {std::min(l.x, r.x), std::min(l.y, r.y), std::min(l.z, r.z)}; }
inline V3 hmax(V3 l, V3 r) { return {std::max(l.x, r.x), std::max(l.y, r.y), std::max(l.z, r.z)}; }

struct V4 {
	float x = 0.0f, y = 0.0f, z = 0.0f, w = 0.0f;
};
inline V4 operator*(float s, V4 v) { return {v.x * s, v.y * s, v.z * s, v.w * s}; }  // vec4.h:211
inline V4 operator+(V4 a, V4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }

// Spectrum is three floats with the same arithmetic (lib/spectrum.h).
using Spec = V3;
inline float luma(Spec s) { return 0.2126f * s.x + 0.7152f * s.y + 0.0722f * s.z; }   // spectrum.h:135

// ---------------------------------------------------------------- Mat4 (lib/mat4.h), column-major m[col][row]
struct M4 {
	float m[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
	V4 col(int i) const { return {m[i][0], m[i][1], m[i][2], m[i][3]}; }
	// mat4.h:175  v[0]*cols[0] + v[1]*cols[1] + v[2]*cols[2] + v[3]*cols[3]
	V4 mul(V4 v) const { return v.x * col(0) + v.y * col(1) + v.z * col(2) + v.w * col(3); }
	V3 point(V3 v) const {  // mat4.h:180  operator*(Vec3): w = 1, then project()
		V4 r = mul({v.x, v.y, v.z, 1.0f});
		return {r.x / r.w, r.y / r.w, r.z / r.w};
	}
	V3 rotate(V3 v) const {  // mat4.h:184  w = 0, xyz()
		V4 r = mul({v.x, v.y, v.z, 0.0f});
		return {r.x, r.y, r.z};
	}
	M4 T() const {  // mat4.h:315
		M4 r;
		for (int i = 0; i < 4; i++)
			for (int j = 0; j < 4; j++) r.m[i][j] = m[j][i];
		return r;
	}
	bool is_identity() const {  // mat4.h:276 (T != Mat4::I), element-wise float compare
		for (int i = 0; i < 4; i++)
			for (int j = 0; j < 4; j++)
				if (m[i][j] != (i == j ? 1.0f : 0.0f)) return false;
		return true;
	}
	M4 operator*(const M4& o) const {  // mat4.h:162
		M4 ret;
		for (int i = 0; i < 4; i++)
			for (int j = 0; j < 4; j++) {
				ret.m[i][j] = 0.0f;
				for (int k = 0; k < 4; k++) ret.m[i][j] += o.m[i][k] * m[k][j];
			}
		return ret;
	}
	float det() const;    // mat4.h:237
	M4 inverse() const;   // mat4.h:325
	static M4 rotate_to(V3 dir);  // mat4.h:379
};

inline float M4::det() const {
	const auto& c = m;
	return c[0][3] * c[1][2] * c[2][1] * c[3][0] - c[0][2] * c[1][3] * c[2][1] * c[3][0] -
	       c[0][3] * c[1][1] * c[2][2] * c[3][0] + c[0][1] * c[1][3] * c[2][2] * c[3][0] +
	       c[0][2] * c[1][1] * c[2][3] * c[3][0] - c[0][1] * c[1][2] * c[2][3] * c[3][0] -
	       c[0][3] * c[1][2] * c[2][0] * c[3][1] + c[0][2] * c[1][3] * c[2][0] * c[3][1] +
	       c[0][3] * c[1][0] * c[2][2] * c[3][1] - c[0][0] * c[1][3] * c[2][2] * c[3][1] -
	       c[0][2] * c[1][0] * c[2][3] * c[3][1] + c[0][0] * c[1][2] * c[2][3] * c[3][1] +
	       c[0][3] * c[1][1] * c[2][0] * c[3][2] - c[0][1] * c[1][3] * c[2][0] * c[3][2] -
	       c[0][3] * c[1][0] * c[2][1] * c[3][2] + c[0][0] * c[1][3] * c[2][1] * c[3][2] +
	       c[0][1] * c[1][0] * c[2][3] * c[3][2] - c[0][0] * c[1][1] * c[2][3] * c[3][2] -
	       c[0][2] * c[1][1] * c[2][0] * c[3][3] + c[0][1] * c[1][2] * c[2][0] * c[3][3] +
	       c[0][2] * c[1][0] * c[2][1] * c[3][3] - c[0][0] * c[1][2] * c[2][1] * c[3][3] -
	       c[0][1] * c[1][0] * c[2][2] * c[3][3] + c[0][0] * c[1][1] * c[2][2] * c[3][3];
}

inline M4 M4::inverse() const {
	const auto& a = m;
	M4 r;
	r.m[0][0] = a[1][2] * a[2][3] * a[3][1] - a[1][3] * a[2][2] * a[3][1] + a[1][3] * a[2][1] * a[3][2] -
	            a[1][1] * a[2][3] * a[3][2] - a[1][2] * a[2][1] * a[3][3] + a[1][1] * a[2][2] * a[3][3];
	r.m[0][1] = a[0][3] * a[2][2] * a[3][1] - a[0][2] * a[2][3] * a[3][1] - a[0][3] * a[2][1] * a[3][2] +
	            a[0][1] * a[2][3] * a[3][2] + a[0][2] * a[2][1] * a[3][3] - a[0][1] * a[2][2] * a[3][3];
	r.m[0][2] = a[0][2] * a[1][3] * a[3][1] - a[0][3] * a[1][2] * a[3][1] + a[0][3] * a[1][1] * a[3][2] -
	            a[0][1] * a[1][3] * a[3][2] - a[0][2] * a[1][1] * a[3][3] + a[0][1] * a[1][2] * a[3][3];
	r.m[0][3] = a[0][3] * a[1][2] * a[2][1] - a[0][2] * a[1][3] * a[2][1] - a[0][3] * a[1][1] * a[2][2] +
	            a[0][1] * a[1][3] * a[2][2] + a[0][2] * a[1][1] * a[2][3] - a[0][1] * a[1][2] * a[2][3];
	r.m[1][0] = a[1][3] * a[2][2] * a[3][0] - a[1][2] * a[2][3] * a[3][0] - a[1][3] * a[2][0] * a[3][2] +
	            a[1][0] * a[2][3] * a[3][2] + a[1][2] * a[2][0] * a[3][3] - a[1][0] * a[2][2] * a[3][3];
	r.m[1][1] = a[0][2] * a[2][3] * a[3][0] - a[0][3] * a[2][2] * a[3][0] + a[0][3] * a[2][0] * a[3][2] -
	            a[0][0] * a[2][3] * a[3][2] - a[0][2] * a[2][0] * a[3][3] + a[0][0] * a[2][2] * a[3][3];
	r.m[1][2] = a[0][3] * a[1][2] * a[3][0] - a[0][2] * a[1][3] * a[3][0] - a[0][3] * a[1][0] * a[3][2] +
	            a[0][0] * a[1][3] * a[3][2] + a[0][2] * a[1][0] * a[3][3] - a[0][0] * a[1][2] * a[3][3];
	r.m[1][3] = a[0][2] * a[1][3] * a[2][0] - a[0][3] * a[1][2] * a[2][0] + a[0][3] * a[1][0] * a[2][2] -
	            a[0][0] * a[1][3] * a[2][2] - a[0][2] * a[1][0] * a[2][3] + a[0][0] * a[1][2] * a[2][3];
	r.m[2][0] = a[1][1] * a[2][3] * a[3][0] - a[1][3] * a[2][1] * a[3][0] + a[1][3] * a[2][0] * a[3][1] -
	            a[1][0] * a[2][3] * a[3][1] - a[1][1] * a[2][0] * a[3][3] + a[1][0] * a[2][1] * a[3][3];
	r.m[2][1] = a[0][3] * a[2][1] * a[3][0] - a[0][1] * a[2][3] * a[3][0] - a[0][3] * a[2][0] * a[3][1] +
	            a[0][0] * a[2][3] * a[3][1] + a[0][1] * a[2][0] * a[3][3] - a[0][0] * a[2][1] * a[3][3];
	r.m[2][2] = a[0][1] * a[1][3] * a[3][0] - a[0][3] * a[1][1] * a[3][0] + a[0][3] * a[1][0] * a[3][1] -
	            a[0][0] * a[1][3] * a[3][1] - a[0][1] * a[1][0] * a[3][3] + a[0][0] * a[1][1] * a[3][3];
	r.m[2][3] = a[0][3] * a[1][1] * a[2][0] - a[0][1] * a[1][3] * a[2][0] - a[0][3] * a[1][0] * a[2][1] +
	            a[0][0] * a[1][3] * a[2][1] + a[0][1] * a[1][0] * a[2][3] - a[0][0] * a[1][1] * a[2][3];
	r.m[3][0] = a[1][2] * a[2][1] * a[3][0] - a[1][1] * a[2][2] * a[3][0] - a[1][2] * a[2][0] * a[3][1] +
	            a[1][0] * a[2][2] * a[3][1] + a[1][1] * a[2][0] * a[3][2] - a[1][0] * a[2][1] * a[3][2];
	r.m[3][1] = a[0][1] * a[2][2] * a[3][0] - a[0][2] * a[2][1] * a[3][0] + a[0][2] * a[2][0] * a[3][1] -
	            a[0][0] * a[2][2] * a[3][1] - a[0][1] * a[2][0] * a[3][2] + a[0][0] * a[2][1] * a[3][2];
	r.m[3][2] = a[0][2] * a[1][1] * a[3][0] - a[0][1] * a[1][2] * a[3][0] - a[0][2] * a[1][0] * a[3][1] +
	            a[0][0] * a[1][2] * a[3][1] + a[0][1] * a[1][0] * a[3][2] - a[0][0] * a[1][1] * a[3][2];
	r.m[3][3] = a[0][1] * a[1][2] * a[2][0] - a[0][2] * a[1][1] * a[2][0] + a[0][2] * a[1][0] * a[2][1] -
	            a[0][0] * a[1][2] * a[2][1] - a[0][1] * a[1][0] * a[2][2] + a[0][0] * a[1][1] * a[2][2];
	float d = det();
	for (int i = 0; i < 4; i++)
		for (int j = 0; j < 4; j++) r.m[i][j] = r.m[i][j] / d;   // mat4.h:375  r /= m.det()
	return r;
}

inline M4 M4::rotate_to(V3 dir) {
	dir = dir.unit();  // dir.normalize(): same three divisions by the norm
	M4 r;
	if (std::abs(dir.y - 1.0f) < EPS_F) return r;
	if (std::abs(dir.y + 1.0f) < EPS_F) {
		r.m[1][1] = -1.0f;
		return r;
	}
	V3 x = cross(dir, V3{0.0f, 1.0f, 0.0f}).unit();
	V3 z = cross(x, dir).unit();
	r.m[0][0] = x.x, r.m[0][1] = x.y, r.m[0][2] = x.z, r.m[0][3] = 0.0f;
	r.m[1][0] = dir.x, r.m[1][1] = dir.y, r.m[1][2] = dir.z, r.m[1][3] = 0.0f;
	r.m[2][0] = z.x, r.m[2][1] = z.y, r.m[2][2] = z.z, r.m[2][3] = 0.0f;
	return r;
}

// ---------------------------------------------------------------- Ray (lib/ray.h)
struct Ray {
	V3 point, dir;
	V2 dist_bounds{0.0f, INF_F};
	uint32_t depth = 0;
	Ray() = default;
	// ray.h:16  the constructor normalises dir
	Ray(V3 p, V3 d, V2 b = V2{0.0f, INF_F}, uint32_t depth_ = 0) : point(p), dir(d.unit()), dist_bounds(b), depth(depth_) {}
	void transform(const M4& t) {  // ray.h:32
		point = t.point(point);
		dir = t.rotate(dir);
		float d = dir.norm();
		dist_bounds = dist_bounds * d;
		dir = dir / d;
	}
};

// ---------------------------------------------------------------- BBox (lib/bbox.h)
struct BBox {
	V3 min{FLT_MAX, FLT_MAX, FLT_MAX}, max{-FLT_MAX, -FLT_MAX, -FLT_MAX};
	void enclose(V3 p) { min = hmin(min, p), max = hmax(max, p); }
	void enclose(const BBox& b) { min = hmin(min, b.min), max = hmax(max, b.max); }
	V3 center() const { return (min + max) * 0.5f; }                          // bbox.h:45
	bool empty() const { return min.x > max.x || min.y > max.y || min.z > max.z; }
	float surface_area() const {                                               // bbox.h:55
		if (empty()) return 0.0f;
		V3 e = max - min;
		return 2.0f * (e.x * e.z + e.x * e.y + e.y * e.z);
	}
	void transform(const M4& t) {  // bbox.h:62
		V3 amin = min, amax = max;
		float lo[3] = {t.m[3][0], t.m[3][1], t.m[3][2]}, hi[3] = {t.m[3][0], t.m[3][1], t.m[3][2]};
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) {
				float a = t.m[j][i] * amin[j], b = t.m[j][i] * amax[j];
				if (a < b) lo[i] += a, hi[i] += b;
				else lo[i] += b, hi[i] += a;
			}
		min = {lo[0], lo[1], lo[2]}, max = {hi[0], hi[1], hi[2]};
	}
	// bbox.h:81-154.  True divisions; every reject is a strict comparison, so a NaN
	// (0/0 when dir.k == 0 and the origin lies on a slab plane) never rejects.
	bool hit(const Ray& ray, V2& times) const {
		float t_init_min = times.x, t_init_max = times.y;
		if (ray.dir.x == 0 && (ray.point.x < min.x || ray.point.x > max.x)) return false;
		if (ray.dir.y == 0 && (ray.point.y < min.y || ray.point.y > max.y)) return false;
		if (ray.dir.z == 0 && (ray.point.z < min.z || ray.point.z > max.z)) return false;
		float tx0 = (min.x - ray.point.x) / ray.dir.x, tx1 = (max.x - ray.point.x) / ray.dir.x;
		if (tx0 > tx1) std::swap(tx0, tx1);
		float ty0 = (min.y - ray.point.y) / ray.dir.y, ty1 = (max.y - ray.point.y) / ray.dir.y;
		if (ty0 > ty1) std::swap(ty0, ty1);
		float tz0 = (min.z - ray.point.z) / ray.dir.z, tz1 = (max.z - ray.point.z) / ray.dir.z;
		if (tz0 > tz1) std::swap(tz0, tz1);
		if (tx0 > ty1 || tx1 < ty0) return false;
		float lo = (tx0 > ty0) ? tx0 : ty0;
		float hi = (tx1 < ty1) ? tx1 : ty1;
		if (lo > tz1 || hi < tz0) return false;
		lo = (lo > tz0) ? lo : tz0;
		hi = (hi < tz1) ? hi : tz1;
		if (lo > t_init_max || hi < t_init_min) return false;
		if (lo > t_init_min) times.x = lo;
		if (hi < t_init_max) times.y = hi;
		return true;
	}
};

// ---------------------------------------------------------------- Trace (pathtracer/trace.h)
struct Trace {
	bool hit = false;
	float distance = 0.0f;
	V3 position, normal, origin;
	V2 uv;
	int32_t material = -1;   // reference: const Material*; here an index, -1 = nullptr
	// Not in the reference record; carried so identity can be compared:
	uint32_t prim = 0xFFFFFFFFu;   // index into the BVH-ordered primitive array of the mesh
	uint32_t inst = 0xFFFFFFFFu;   // index into the BVH-ordered instance array of the scene
	float bu = 0.0f, bv = 0.0f;    // barycentrics u, v of Triangle::hit

	static Trace min(const Trace& l, const Trace& r) {  // trace.h:28
		if (l.hit && r.hit) {
			if (l.distance < r.distance) return l;
			return r;
		}
		if (l.hit) return l;
		if (r.hit) return r;
		return {};
	}
	void transform(const M4& T, const M4& N) {  // trace.h:38
		position = T.point(position);
		origin = T.point(origin);
		normal = N.rotate(normal).unit();
		distance = (position - origin).norm();
	}
};

struct Counters {
	uint64_t nodes = 0;  // nodes whose box was tested (hitRecursive entries)
	uint64_t tris = 0;   // Triangle::hit calls
};

// ---------------------------------------------------------------- Triangle (pathtracer/tri_mesh.{h,cpp})
struct Vert {  // Tri_Mesh_Vert, tri_mesh.h:13 -- 32-byte AoS
	V3 position, normal;
	V2 uv;
};

struct Triangle {
	uint32_t v0, v1, v2;
	const Vert* vertex_list;
	uint32_t id;  // index of this triangle in the input order (not in the reference)

	BBox bbox() const {  // tri_mesh.cpp:10-44
		const V3 &a = vertex_list[v0].position, &b = vertex_list[v1].position, &c = vertex_list[v2].position;
		float eps = 0.0001f;
		float min_x = std::min(a.x, std::min(b.x, c.x)), min_y = std::min(a.y, std::min(b.y, c.y)), min_z = std::min(a.z, std::min(b.z, c.z));
		float max_x = std::max(a.x, std::max(b.x, c.x)), max_y = std::max(a.y, std::max(b.y, c.y)), max_z = std::max(a.z, std::max(b.z, c.z));
		if (min_x == max_x) min_x -= eps, max_x += eps;
		if (min_y == max_y) min_y -= eps, max_y += eps;
		if (min_z == max_z) min_z -= eps, max_z += eps;
		BBox box;
		box.enclose(V3{min_x, min_y, min_z});
		box.enclose(V3{max_x, max_y, max_z});
		return box;
	}

	Trace hit(const Ray& ray) const {  // tri_mesh.cpp:46-111
		const Vert &p0 = vertex_list[v0], &p1 = vertex_list[v1], &p2 = vertex_list[v2];
		V3 s = ray.point - p0.position;
		V3 e1 = p1.position - p0.position;
		V3 e2 = p2.position - p0.position;
		V3 dir = ray.dir;
		Trace ret;
		ret.origin = ray.point;
		float denom = dot(cross(e1, dir), e2);
		if (denom == 0) return ret;
		V3 cramer{-1 * dot(cross(s, e2), dir), dot(cross(e1, dir), s), -dot(cross(s, e2), e1)};
		V3 uvt = cramer / denom;
		float u = uvt.x, v = uvt.y, t = uvt.z;
		if (u < 0 || v < 0 || u + v > 1 || t < 0) return ret;
		V3 dist = ray.dir * t;
		float distance = dist.norm();
		if (distance > ray.dist_bounds.y || distance < ray.dist_bounds.x) return ret;
		ret.hit = true;
		ret.distance = distance;
		ret.position = ray.point + dist;
		ret.normal = (1 - u - v) * p0.normal + u * p1.normal + v * p2.normal;
		ret.uv = (1 - u - v) * p0.uv + u * p1.uv + v * p2.uv;
		ret.bu = u, ret.bv = v;
		return ret;
	}
};

// ---------------------------------------------------------------- BVH (pathtracer/bvh.{h,cpp})
// Guarded restatement of the binned-SAH builder: the reference algorithm (bvh.cpp:31-243)
// step for step -- including its habit of taking the child sizes and boxes from the bin
// counts (bvh.cpp:213-221) while partitioning with a separately evaluated predicate
// (bvh.cpp:191-199), which can leave a primitive in a range whose box does not contain it.
// The one guard: the bin index is clamped to [0,7], where the reference indexes bins[] with
// an unclamped uint32_t cast (bvh.cpp:111-112: undefined behaviour, a crash on large meshes).
// `clamped` counts the primitives the guard touched; while it is 0 this is the reference's
// tree.  `predicate_disagreements` counts primitives the predicate put on the other side
// from their bin (information only).
template<typename Prim> struct BVH {
	struct Node {  // bvh.h:15 -- 56 bytes, as in the reference
		BBox bbox;
		size_t start = 0, size = 0, l = 0, r = 0;
		bool is_leaf() const { return l == r; }
	};
	std::vector<Prim> primitives;
	std::vector<Node> nodes;
	size_t root_idx = 0;
	uint64_t predicate_disagreements = 0, clamped = 0, dropped = 0;

	void build(std::vector<Prim>&& prims, size_t max_leaf_size) {  // bvh.cpp:31-80
		nodes.clear();
		primitives = std::move(prims);
		BBox scene;
		for (size_t i = 0; i < primitives.size(); i++) scene.enclose(primitives[i].bbox());
		Node root;
		root.bbox = scene;
		root.start = 0;
		root.size = primitives.size();
		nodes.push_back(root);
		if (nodes[0].size > max_leaf_size) build_recursive(0, max_leaf_size);
		else nodes[0].l = nodes[0].r = 0;
	}

	void build_recursive(size_t parent, size_t max_leaf_size) {  // bvh.cpp:85-243
		constexpr uint32_t NB = 8;
		const BBox pbox = nodes[parent].bbox;
		const size_t pstart = nodes[parent].start, psize = nodes[parent].size;
		float lowest = INFINITY;
		uint32_t best_axis = 0, best_split = 0;
		BBox best_lbox, best_rbox;
		size_t best_nl = 0, best_nr = 0;
		std::vector<uint8_t> bin_of(psize), best_bin_of;

		for (uint32_t axis = 0; axis < 3; axis++) {
			BBox bin_box[NB];
			uint32_t bin_n[NB] = {};
			float span_min = pbox.min[axis], span_max = pbox.max[axis];
			float bin_len = (span_max - span_min) / NB;
			for (size_t p = pstart; p < pstart + psize; p++) {
				BBox pb = primitives[p].bbox();
				float c = pb.center()[axis];
				float f = std::floor((c - span_min) / bin_len);
				uint32_t b;
				if (!(f >= 0.0f)) b = 0, clamped++;          // negative or NaN: UB in the reference
				else if (f > float(NB - 1)) b = NB - 1, clamped++;  // >= 8: out-of-range write in the reference
				else b = uint32_t(f);
				bin_of[p - pstart] = uint8_t(b);
				bin_n[b]++;
				bin_box[b].enclose(pb);
			}
			for (uint32_t split = 1; split < NB; split++) {
				BBox lbox, rbox;
				uint32_t nl = 0, nr = 0;
				for (uint32_t i = 0; i < split; i++) nl += bin_n[i], lbox.enclose(bin_box[i]);
				for (uint32_t i = split; i < NB; i++) nr += bin_n[i], rbox.enclose(bin_box[i]);
				float cost = lbox.surface_area() * nl + rbox.surface_area() * nr;   // bvh.cpp:139
				if (cost < lowest && nl > 0 && nr > 0) {                             // bvh.cpp:142
					lowest = cost, best_axis = axis, best_split = split;
					best_lbox = lbox, best_rbox = rbox, best_nl = nl, best_nr = nr;
					best_bin_of = bin_of;
				}
			}
		}

		if (best_nl == 0 && best_nr == 0) {
			// No split separates the primitives (bvh.cpp:183-188): the reference appends two
			// empty leaves and the node's primitives are never intersected again.
			dropped += psize;
			best_bin_of.assign(psize, 0);
		} else {
			// How often would the reference's predicate (bvh.cpp:191-197) disagree with the bins?
			float span_min = pbox.min[best_axis], span_max = pbox.max[best_axis];
			float bin_len = (span_max - span_min) / NB;
			for (size_t p = pstart; p < pstart + psize; p++) {
				float c = primitives[p].bbox().center()[best_axis];
				bool pred_left = c < span_min + bin_len * best_split;
				bool bin_left = best_bin_of[p - pstart] < best_split;
				if (pred_left != bin_left) predicate_disagreements++;
			}
		}

		// std::partition as libstdc++ implements it for bidirectional iterators
		// (bits/stl_algo.h, __partition), with the reference's predicate (bvh.cpp:191-197).  The
		// resulting order matters: equal-distance hits are resolved by array position
		// (trace.h:28-36), and the children's ranges are cut at the bin count, not at the
		// partition point.
		{
			float span_min = pbox.min[best_axis], span_max = pbox.max[best_axis];
			float bin_len = (span_max - span_min) / NB;
			auto is_left = [&](size_t i) {
				float c = primitives[pstart + i].bbox().center()[best_axis];
				return c < span_min + bin_len * best_split;
			};
			size_t first = 0, last = psize;
			while (true) {
				while (true) {
					if (first == last) goto done;
					else if (is_left(first)) ++first;
					else break;
				}
				--last;
				while (true) {
					if (first == last) goto done;
					else if (!is_left(last)) --last;
					else break;
				}
				std::swap(primitives[pstart + first], primitives[pstart + last]);
				++first;
			}
		done:;
		}

		Node left, right;
		left.bbox = best_lbox, left.start = pstart, left.size = best_nl;
		right.bbox = best_rbox, right.start = left.start + left.size, right.size = best_nr;
		nodes.push_back(left);
		nodes.push_back(right);
		size_t li = nodes.size() - 2, ri = nodes.size() - 1;
		nodes[parent].l = li, nodes[parent].r = ri;
		if (best_nl > max_leaf_size) build_recursive(li, max_leaf_size);
		else nodes[li].l = nodes[li].r = nodes[li].start;
		if (best_nr > max_leaf_size) build_recursive(ri, max_leaf_size);
		else nodes[ri].l = nodes[ri].r = nodes[ri].start;
	}

	BBox bbox() const {  // bvh.cpp:433
		if (nodes.empty()) return BBox{V3{0, 0, 0}, V3{0, 0, 0}};
		return nodes[root_idx].bbox;
	}

	Trace hit(const Ray& ray, Counters* ctr = nullptr) const {  // bvh.cpp:246-270
		Trace cur;
		cur.distance = INFINITY;
		cur.hit = false;
		hit_recursive(ray, 0, cur, ctr);
		return cur;
	}

	void hit_recursive(const Ray& ray, size_t idx, Trace& cur, Counters* ctr) const {  // bvh.cpp:272-375
		Node node = nodes[idx];
		V2 bounds{0.0f, INFINITY};
		if (ctr) ctr->nodes++;
		if (!node.bbox.hit(ray, bounds)) return;
		if (node.l == node.r) {
			for (size_t i = node.start; i < node.start + node.size; i++) {
				Trace h = primitives[i].hit(ray, uint32_t(i), ctr);
				cur = Trace::min(cur, h);
			}
			return;
		}
		// Child boxes are not pre-tested (bvh.cpp:320-321 are commented out), so both entry
		// times are 0: left is always first, and right follows unless the running closest
		// hit has a NaN distance (bvh.cpp:337-342: firstHit && distance >= 0, or !firstHit).
		hit_recursive(ray, node.l, cur, ctr);
		bool first_hit = cur.hit;
		bool still = first_hit && cur.distance >= 0.0f;
		if (still || !first_hit) hit_recursive(ray, node.r, cur, ctr);
	}
};

// List (pathtracer/list.h:25-32): linear fold from a default Trace.
template<typename Prim> struct List {
	std::vector<Prim> prims;
	BBox bbox() const {
		BBox ret;
		for (const auto& p : prims) ret.enclose(p.bbox());
		return ret;
	}
	Trace hit(const Ray& ray, Counters* ctr = nullptr) const {
		Trace ret;
		for (size_t i = 0; i < prims.size(); i++) {
			ret = Trace::min(ret, prims[i].hit(ray, uint32_t(i), ctr));
		}
		return ret;
	}
};

// A Triangle as the BVH sees it: hit() tagged with its array position.
struct TriPrim : Triangle {
	Trace hit(const Ray& ray, uint32_t array_index, Counters* ctr) const {
		if (ctr) ctr->tris++;
		Trace t = Triangle::hit(ray);
		if (t.hit) t.prim = array_index;
		return t;
	}
};

// ---------------------------------------------------------------- Tri_Mesh (tri_mesh.cpp:160-196)
struct TriMesh {
	bool use_bvh = true;
	std::vector<Vert> verts;
	BVH<TriPrim> bvh;
	List<TriPrim> list;

	TriMesh() = default;
	TriMesh(const TriMesh&) = delete;
	TriMesh& operator=(const TriMesh&) = delete;

	// verts: n_verts x {pos3, normal3, uv2}; idx: n_tris x 3  (Indexed_Mesh layout)
	void build(const float* v, uint32_t n_verts, const uint32_t* idx, uint32_t n_tris, bool use_bvh_, size_t max_leaf = 4) {
		use_bvh = use_bvh_;
		verts.resize(n_verts);
		for (uint32_t i = 0; i < n_verts; i++) {
			verts[i].position = {v[i * 8 + 0], v[i * 8 + 1], v[i * 8 + 2]};
			verts[i].normal = {v[i * 8 + 3], v[i * 8 + 4], v[i * 8 + 5]};
			verts[i].uv = {v[i * 8 + 6], v[i * 8 + 7]};
		}
		std::vector<TriPrim> tris(n_tris);
		for (uint32_t i = 0; i < n_tris; i++) {
			tris[i].v0 = idx[i * 3], tris[i].v1 = idx[i * 3 + 1], tris[i].v2 = idx[i * 3 + 2];
			tris[i].vertex_list = verts.data();
			tris[i].id = i;
		}
		if (use_bvh) bvh.build(std::move(tris), max_leaf);   // tri_mesh.cpp:173 (max_leaf_size 4)
		else list.prims = std::move(tris);
	}
	BBox bbox() const { return use_bvh ? bvh.bbox() : list.bbox(); }
	Trace hit(const Ray& ray, Counters* ctr = nullptr) const { return use_bvh ? bvh.hit(ray, ctr) : list.hit(ray, ctr); }
	const std::vector<TriPrim>& prims() const { return use_bvh ? bvh.primitives : list.prims; }
};

// ---------------------------------------------------------------- Sphere shape (scene/shape.cpp)
inline V2 sphere_uv(V3 dir) {  // shape.cpp:7-12
	float u = std::atan2(dir.z, dir.x) / (2.0f * PI_F);
	if (u < 0.0f) u += 1.0f;
	float v = std::acos(-1.0f * std::clamp(dir.y, -1.0f, 1.0f)) / PI_F;
	return {u, v};
}

struct Sphere {
	float radius = 1.0f;
	BBox bbox() const {  // shape.cpp:14-19
		BBox b;
		b.enclose(V3{-radius, -radius, -radius});
		b.enclose(V3{radius, radius, radius});
		return b;
	}
	Trace hit(Ray ray) const {  // shape.cpp:21-85
		Trace ret;
		ret.origin = ray.point;
		V3 o = ray.point, d = ray.dir;
		float d2 = d.norm_squared(), o2 = o.norm_squared(), r = radius;
		float disc = 4 * (dot(o, d) * dot(o, d)) - 4 * d2 * (o2 - r * r);
		if (disc < 0) return ret;
		float sq = std::sqrt(disc);
		float t1 = (-2 * dot(o, d) + sq) / (2 * d2);
		float t2 = (-2 * dot(o, d) - sq) / (2 * d2);
		auto within = [&](float t) { return t >= ray.dist_bounds.x && t <= ray.dist_bounds.y; };
		float t = 0;
		if (within(t1) && within(t2)) t = std::min(t1, t2);
		else if (within(t1)) t = t1;
		else if (within(t2)) t = t2;
		else return ret;
		ret.hit = true;
		ret.distance = t;
		ret.position = ray.point + t * ray.dir;
		ret.normal = ret.position.unit();
		ret.uv = sphere_uv(ret.normal);
		return ret;
	}
};

// ---------------------------------------------------------------- Instance (pathtracer/instance.h:15-40)
struct Instance {
	M4 T, iT;
	bool has_transform = false;
	int32_t material = -1;
	const TriMesh* mesh = nullptr;   // exactly one of mesh / sphere is set
	const Sphere* sphere = nullptr;
	uint32_t id = 0;  // input order

	void set(const M4& T_) {
		T = T_;
		iT = T_.inverse();
		has_transform = !T_.is_identity();
	}
	BBox bbox() const {
		BBox box = mesh ? mesh->bbox() : sphere->bbox();
		if (has_transform) box.transform(T);
		return box;
	}
	Trace hit(Ray ray, uint32_t array_index, Counters* ctr) const {
		if (has_transform) ray.transform(iT);
		Trace trace = mesh ? mesh->hit(ray, ctr) : sphere->hit(ray);
		if (trace.hit) {
			trace.material = material;
			trace.inst = array_index;
			if (!mesh) trace.prim = 0;   // a sphere has one primitive
			if (has_transform) trace.transform(T, iT.T());
		}
		return trace;
	}
};

// ---------------------------------------------------------------- RNG (util/rand.cpp) + counter-based twin
// mode 0: std::mt19937 exactly as the reference (rand.cpp:17-21, unit = scalbn(float(mt()), -32)).
// mode 1: Philox-4x32-10 keyed by (key0, key1), draw i = word i%4 of block i/4 -- the stream
//         the GPU pipeline uses, so CPU and GPU paths can be compared draw for draw.
struct Philox {
	static inline void round(uint32_t c[4], const uint32_t k[2]) {
		const uint64_t p0 = uint64_t(0xD2511F53u) * c[0], p1 = uint64_t(0xCD9E8D57u) * c[2];
		uint32_t n0 = uint32_t(p1 >> 32) ^ c[1] ^ k[0], n1 = uint32_t(p1), n2 = uint32_t(p0 >> 32) ^ c[3] ^ k[1], n3 = uint32_t(p0);
		c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
	}
	static inline void block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
		uint32_t c[4] = {c0, c1, c2, c3}, k[2] = {k0, k1};
		for (int i = 0; i < 10; i++) {
			round(c, k);
			k[0] += 0x9E3779B9u, k[1] += 0xBB67AE85u;
		}
		out[0] = c[0], out[1] = c[1], out[2] = c[2], out[3] = c[3];
	}
};

struct RNG {
	int mode = 0;
	std::mt19937 mt;
	uint32_t key0 = 0, key1 = 0, stream_lo = 0, stream_hi = 0, draws = 0, buf[4] = {};
	RNG() = default;
	explicit RNG(uint32_t seed) : mode(0), mt(seed) {}
	static RNG philox(uint32_t k0, uint32_t k1, uint32_t s_lo, uint32_t s_hi) {
		RNG r;
		r.mode = 1, r.key0 = k0, r.key1 = k1, r.stream_lo = s_lo, r.stream_hi = s_hi;
		return r;
	}
	uint32_t next() {
		if (mode == 0) return uint32_t(mt());
		if ((draws & 3u) == 0) Philox::block(draws >> 2, stream_lo, stream_hi, 0, key0, key1, buf);
		return buf[draws++ & 3u];
	}
	float unit() { return std::scalbn(float(next()), -32); }
	bool coin_flip(float p) { return unit() < p; }
};

// ---------------------------------------------------------------- Samplers (pathtracer/samplers.cpp)
inline V3 cosine_hemisphere_sample(RNG& rng) {  // samplers.cpp:67-78
	float phi = rng.unit() * 2.0f * PI_F;
	float cos_t = std::sqrt(rng.unit());
	float sin_t = std::sqrt(1 - cos_t * cos_t);
	float x = std::cos(phi) * sin_t;
	float z = std::sin(phi) * sin_t;
	float y = cos_t;
	return {x, y, z};
}
inline float cosine_hemisphere_pdf(V3 dir) {  // samplers.cpp:80-83
	if (dir.y < 0.0f) return 0.0f;
	return dir.y / PI_F;
}

// ---------------------------------------------------------------- Textures (scene/texture.cpp)
struct Texture {
	int type = 0;  // 0 Constant, 1 Image
	Spec color{0.75f, 0.75f, 0.75f};
	float scale = 1.0f;
	int sampler = 0;  // 0 nearest, 1 bilinear (trilinear at lod 0 == bilinear, texture.cpp:78-80)
	uint32_t w = 0, h = 0;
	std::vector<Spec> pixels;
	Spec at(int32_t x, int32_t y) const { return pixels[size_t(y) * w + x]; }
	Spec evaluate(V2 uv) const {
		if (type == 0) return color * scale;  // texture.cpp:216-218
		float x = w * std::clamp(uv.x, 0.0f, 1.0f), y = h * std::clamp(uv.y, 0.0f, 1.0f);
		if (sampler == 0) {  // texture.cpp:9-24
			int32_t ix = std::min(int32_t(std::floor(x)), int32_t(w) - 1), iy = std::min(int32_t(std::floor(y)), int32_t(h) - 1);
			return at(ix, iy);
		}
		// texture.cpp:26-67.  The lower index is not clamped at 0 in the reference either
		// (int32_t(-1.0f) indexes out of bounds when x < 0.5); it is clamped here.
		float i = std::floor(x - 0.5f), j = std::floor(y - 0.5f);
		auto cx = [&](float f) { return std::max(0, std::min(int32_t(f), int32_t(w) - 1)); };
		auto cy = [&](float f) { return std::max(0, std::min(int32_t(f), int32_t(h) - 1)); };
		Spec lb = at(cx(i), cy(j)), lt = at(cx(i), cy(j + 1.0f)), rb = at(cx(i + 1.0f), cy(j)), rt = at(cx(i + 1.0f), cy(j + 1.0f));
		float s = x - (i + 0.5f), t = y - (j + 0.5f);
		return (1 - t) * ((1 - s) * lb + s * rb) + t * ((1 - s) * lt + s * rt);
	}
};

// ---------------------------------------------------------------- Materials (scene/material.cpp)
enum MatType { LAMBERTIAN = 0, MIRROR = 1, REFRACT = 2, GLASS = 3, EMISSIVE = 4 };
enum FresnelMode { GCC_INT_ABS = 0, FLOAT_ABS = 1 };

struct Scatter {
	V3 direction;
	Spec attenuation;
	bool specular = false;
};

inline V3 reflect(V3 dir) {  // material.cpp:8-19
	V3 ret = -dir;
	ret.y = -ret.y;
	return ret;
}

inline V3 refract(V3 out_dir, float ior, bool& was_internal) {  // material.cpp:21-54
	float cos_i = dot(out_dir, V3{0, 1, 0});
	bool entering = cos_i > 0;
	float eta_i = entering ? 1.0f : ior, eta_t = entering ? ior : 1.0f;
	float sin2_i = std::max(0.0f, 1.0f - cos_i * cos_i);
	float sin2_t = eta_i * eta_i * sin2_i / (eta_t * eta_t);
	if (sin2_t >= 1) {
		was_internal = true;
		return V3{};
	}
	float cos_t = std::sqrt(1 - sin2_t);
	V3 in_dir{-out_dir.x, 0.0f, -out_dir.z};
	in_dir = in_dir.unit();
	in_dir = in_dir * std::sqrt(sin2_t);
	in_dir.y = cos_t;
	if (0 < out_dir.y) in_dir.y = -in_dir.y;
	return in_dir;
}

// material.cpp:58-71.  `abs(dot(...))` is unqualified: with g++/libstdc++ it resolves to
// ::abs(int), so |cos| truncates to an integer (GCC_INT_ABS); with the float overload
// visible (MSVC, or `using std::abs`) it is fabsf (FLOAT_ABS).
inline float schlick(V3 in_dir, float ior, int mode) {
	float d = dot(in_dir, V3{0, 1, 0});
	float cos_theta = mode == GCC_INT_ABS ? float(std::abs(int(d))) : std::abs(d);
	float r0 = (1 - ior) / (1 + ior);
	r0 = r0 * r0;
	return float(double(r0) + double(1.0f - r0) * (std::pow(double(1 - cos_theta), 5)));
}

struct Material {
	int type = LAMBERTIAN;
	int tex_a = -1;  // albedo / reflectance / emissive (Refract: unused)
	int tex_b = -1;  // transmittance (Refract, Glass)
	float ior = 1.5f;
	bool is_emissive() const { return type == EMISSIVE; }
	bool is_specular() const { return type != LAMBERTIAN; }                 // material.h + material.cpp:203,280,321
	bool is_sided() const { return type == REFRACT || type == GLASS; }      // material.cpp:207,284
};

struct MaterialTable {
	std::vector<Texture> textures;
	std::vector<Material> materials;
	int fresnel_mode = GCC_INT_ABS;
	Spec tex(int id, V2 uv) const { return textures[size_t(id)].evaluate(uv); }

	Spec evaluate(const Material& m, V3 out, V3 in, V2 uv) const {  // material.cpp:73-84; others return {}
		(void)out;
		if (m.type != LAMBERTIAN) return {};
		float cos_theta = std::max(0.0f, in.y);
		return tex(m.tex_a, uv) / PI_F * cos_theta;
	}
	float pdf(const Material& m, V3 out, V3 in) const {  // material.cpp:104-110; others 0
		(void)out;
		if (m.type != LAMBERTIAN) return 0.0f;
		return cosine_hemisphere_pdf(in);
	}
	Spec emission(const Material& m, V2 uv) const {  // material.cpp:313-315
		if (m.type != EMISSIVE) return {};
		return tex(m.tex_a, uv);
	}
	Scatter scatter(const Material& m, RNG& rng, V3 out, V2 uv) const {
		Scatter ret;
		switch (m.type) {
		case LAMBERTIAN:  // material.cpp:86-102
			ret.direction = cosine_hemisphere_sample(rng);
			ret.attenuation = evaluate(m, out, ret.direction, uv);
			ret.specular = false;
			break;
		case MIRROR:  // material.cpp:128-140
			ret.direction = reflect(out);
			ret.attenuation = tex(m.tex_a, uv);
			ret.specular = true;
			break;
		case REFRACT: {  // material.cpp:162-189
			bool internal = false;
			ret.direction = refract(out, m.ior, internal).unit();
			if (internal) {
				ret.direction = reflect(out).unit();
				ret.attenuation = Spec{1.0f, 1.0f, 1.0f};
			} else {
				bool entering = dot(out, V3{0, 1, 0}) > 0;
				float ratio = entering ? 1.0f / m.ior : m.ior;
				ret.attenuation = tex(m.tex_b, uv) * (ratio * ratio);
			}
			ret.specular = true;
			break;
		}
		case GLASS: {  // material.cpp:223-266
			bool internal = false;
			ret.direction = refract(out, m.ior, internal).unit();
			float fresnel = schlick(out, m.ior, fresnel_mode);
			if (internal || rng.coin_flip(fresnel)) {
				ret.direction = reflect(out).unit();
				ret.attenuation = tex(m.tex_a, uv);
			} else {
				bool entering = dot(out, V3{0, 1, 0}) > 0;
				float ratio = entering ? 1.0f / m.ior : m.ior;
				ret.attenuation = tex(m.tex_b, uv) * (ratio * ratio);
			}
			ret.specular = true;
			break;
		}
		default:  // Emissive, material.cpp:301-307
			ret.specular = true;
			break;
		}
		return ret;
	}
};

// ---------------------------------------------------------------- Lights
struct DeltaLight {  // scene/delta_light.cpp:7-40 + Light_Instance (instance.h:74-91)
	int type = 0;    // 0 Point, 1 Directional, 2 Spot
	Spec color{1, 1, 1};
	float intensity = 1.0f, inner_angle = 30.0f, outer_angle = 45.0f;
	M4 T, iT;
	bool has_transform = false;
	void set(const M4& T_) { T = T_, iT = T_.inverse(), has_transform = !T_.is_identity(); }
	struct Incoming {
		Spec radiance;
		V3 direction;
		float distance = 0.0f;
	};
	Incoming incoming(V3 from) const {
		if (has_transform) from = iT.point(from);
		Incoming ret;
		if (type == 1) {
			ret.direction = V3{0.0f, -1.0f, 0.0f};
			ret.distance = INF_F;
			ret.radiance = color * intensity;
		} else {
			ret.direction = -from.unit();
			ret.distance = from.norm();
			ret.radiance = color * intensity;
			if (type == 2) {
				float angle = std::atan2(std::sqrt(from.x * from.x + from.z * from.z), from.y);
				angle = std::abs(angle * (180.0f / PI_F));
				float e0 = inner_angle / 2.0f, e1 = outer_angle / 2.0f;
				float t = std::clamp((angle - e0) / (e1 - e0), 0.0f, 1.0f);
				float ss = t * t * (3.0f - 2.0f * t);
				ret.radiance = (1.0f - ss) * color * intensity;
			}
		}
		// Delta_Lights::Incoming::transform (delta_light.h:29-31) rotates the direction only:
		// it is not re-normalised and the distance stays in the light's local units.
		if (has_transform) ret.direction = T.rotate(ret.direction);
		return ret;
	}
};

struct EnvLight {  // scene/env_light.cpp:11-31
	int type = 0;  // 0 Hemisphere, 1 Sphere
	int tex = -1;
	Spec evaluate(const MaterialTable& mt, V3 dir) const {
		if (type == 0 && dir.y < 0.0f) return {};
		return mt.tex(tex, sphere_uv(dir));
	}
};

// ---------------------------------------------------------------- Camera (scene/camera.cpp:7-48)
struct Camera {
	float vertical_fov = 60.0f, aspect_ratio = 1.77778f;
	uint32_t film_w = 1280, film_h = 720, film_samples = 256, film_max_ray_depth = 8;
	M4 camera_to_world;
	// Two draws feed Vec2(rng.unit()*size.x, rng.unit()*size.y) (samplers.cpp:13); the order in
	// which a compiler evaluates the two arguments is unspecified.  g++ evaluates them right
	// to left, which is what x_second reproduces; the counter-based stream uses x first.
	std::pair<Ray, float> sample_ray(RNG& rng, uint32_t px, uint32_t py, bool x_second) const {
		float a = rng.unit(), b = rng.unit();
		V2 offset = x_second ? V2{b * 1.0f, a * 1.0f} : V2{a * 1.0f, b * 1.0f};
		float offset_pdf = 1.0f / (1.0f * 1.0f);  // samplers.cpp:16-19 (Rect of size (1,1))
		if (offset.x < 0.0f || offset.x > 1.0f || offset.y < 0.0f || offset.y > 1.0f) offset_pdf = 0.0f;
		return {ray_through(px, py, offset), offset_pdf};
	}
	Ray ray_through(uint32_t px, uint32_t py, V2 offset) const {  // camera.cpp:26-46
		V2 sensor = V2{float(px), float(py)} + offset;
		float h = 2.0f * std::tan((vertical_fov / 2.0f) * (PI_F / 180.0f));
		float w = aspect_ratio * h;
		float w_start = -0.5f * w, h_start = -0.5f * h;
		V3 pos{sensor.x / film_w * w + w_start, sensor.y / film_h * h + h_start, -1.0f};
		Ray ray;
		ray.point = V3{};
		ray.dir = pos - ray.point;   // left un-normalised here (camera.cpp:44)
		ray.depth = film_max_ray_depth;
		return ray;
	}
};

// ---------------------------------------------------------------- Scene + Pathtracer::trace (pathtracer.cpp)
struct Scene {
	std::vector<TriMesh*> meshes;    // owned
	std::vector<Sphere> spheres;
	std::vector<Instance> instances_in;   // input order, before the top-level build
	MaterialTable mats;
	std::vector<DeltaLight> delta_lights;
	std::vector<EnvLight> env_lights;
	bool use_bvh = true;

	struct InstPrim : Instance {
		Trace hit(const Ray& ray, uint32_t array_index, Counters* ctr) const { return Instance::hit(ray, array_index, ctr); }
	};
	BVH<InstPrim> top;      // Aggregate(BVH<Instance>) -- pathtracer.cpp:381-385, max_leaf_size 1 (bvh.h:26)
	List<InstPrim> top_list;

	~Scene() {
		for (auto* m : meshes) delete m;
	}
	void build_top() {
		std::vector<InstPrim> objs(instances_in.size());
		for (size_t i = 0; i < objs.size(); i++) static_cast<Instance&>(objs[i]) = instances_in[i], objs[i].id = uint32_t(i);
		if (use_bvh) top.build(std::move(objs), 1);
		else top_list.prims = std::move(objs);
	}
	Trace hit(const Ray& ray, Counters* ctr = nullptr) const { return use_bvh ? top.hit(ray, ctr) : top_list.hit(ray, ctr); }
};

constexpr float RAY_MINIMUM_BOUND = 0.001f;  // pathtracer.cpp:17

struct Shading {
	const Material* bsdf;
	M4 world_to_object, object_to_world;
	V3 pos, out_dir, normal;
	V2 uv;
	uint32_t depth;
};

struct PathStats {
	uint64_t rays = 0, shadow_rays = 0;
	Counters visits;   // nodes tested and Triangle::hit calls of all those rays (reference visit semantics)
};

struct Pathtracer {
	const Scene* scene = nullptr;
	Camera camera;
	PathStats* stats = nullptr;

	Spec sum_delta_lights(const Shading& hit) const {  // pathtracer.cpp:645-666
		if (hit.bsdf->is_specular()) return {};
		Spec radiance;
		for (const auto& light : scene->delta_lights) {
			auto incoming = light.incoming(hit.pos);
			V3 in_dir = hit.world_to_object.rotate(incoming.direction);
			Spec attenuation = scene->mats.evaluate(*hit.bsdf, hit.out_dir, in_dir, hit.uv);
			if (luma(attenuation) == 0.0f) continue;
			Ray shadow_ray(hit.pos, incoming.direction, V2{EPS_F, incoming.distance - EPS_F});
			if (stats) stats->shadow_rays++;
			Trace shadow = scene->hit(shadow_ray, stats ? &stats->visits : nullptr);
			if (!shadow.hit) radiance = radiance + attenuation * incoming.radiance;
		}
		return radiance;
	}

	Spec sample_direct(RNG& rng, const Shading& hit) const {  // pathtracer.cpp:34-78
		Spec radiance = sum_delta_lights(hit);
		Scatter sc = scene->mats.scatter(*hit.bsdf, rng, hit.out_dir, hit.uv);
		V3 in_world = hit.object_to_world.rotate(sc.direction);
		Spec attenuation = sc.specular ? sc.attenuation : scene->mats.evaluate(*hit.bsdf, sc.direction, hit.out_dir, hit.uv);
		Ray r(hit.pos, in_world, V2{RAY_MINIMUM_BOUND, INF_F}, 0);
		Spec rn = trace(rng, r).first;
		rn = rn * attenuation;
		if (!sc.specular) rn = rn / scene->mats.pdf(*hit.bsdf, sc.direction, hit.out_dir);
		return radiance + rn;
	}

	Spec sample_indirect(RNG& rng, const Shading& hit) const {  // pathtracer.cpp:81-124
		Scatter sc = scene->mats.scatter(*hit.bsdf, rng, hit.out_dir, hit.uv);
		V3 in_world = hit.object_to_world.rotate(sc.direction);
		Spec attenuation = sc.specular ? sc.attenuation : scene->mats.evaluate(*hit.bsdf, sc.direction, hit.out_dir, hit.uv);
		Ray r(hit.pos, in_world, V2{RAY_MINIMUM_BOUND, INF_F}, hit.depth - 1);
		Spec radiance = trace(rng, r).second;
		radiance = radiance * attenuation;
		if (!sc.specular) radiance = radiance / scene->mats.pdf(*hit.bsdf, sc.direction, hit.out_dir);
		return radiance;
	}

	std::pair<Spec, Spec> trace(RNG& rng, const Ray& ray) const {  // pathtracer.cpp:126-173
		if (stats) stats->rays++;
		Trace result = scene->hit(ray, stats ? &stats->visits : nullptr);
		if (!result.hit) {
			if (!scene->env_lights.empty()) {
				Spec radiance;
				for (const auto& l : scene->env_lights) radiance = radiance + l.evaluate(scene->mats, ray.dir);
				return {radiance, {}};
			}
			return {};
		}
		if (result.material < 0) return {};
		const Material* bsdf = &scene->mats.materials[size_t(result.material)];
		if (!bsdf->is_sided() && dot(result.normal, ray.dir) > 0.0f) result.normal = -result.normal;
		M4 o2w = M4::rotate_to(result.normal);
		M4 w2o = o2w.T();
		V3 out_dir = w2o.rotate(ray.point - result.position).unit();
		Shading info{bsdf, w2o, o2w, result.position, out_dir, result.normal, result.uv, ray.depth};
		Spec emissive = scene->mats.emission(*bsdf, info.uv);
		if (ray.depth == 0 || bsdf->is_emissive()) return {emissive, {}};
		Spec direct = sample_direct(rng, info);
		return {emissive, direct + sample_indirect(rng, info)};
	}
};

// ---------------------------------------------------------------- tiles + accumulator (pathtracer.cpp:407-477, 522-563)
struct Tile {
	uint32_t seed, x_begin, x_end, y_begin, y_end, s_begin, s_end;
};

inline std::vector<Tile> make_tiles(uint32_t W, uint32_t H, uint32_t samples, uint32_t fixed_seed) {
	constexpr uint32_t tw = 100, th = 100, ts = 50;
	std::mt19937 seeds(fixed_seed);  // pathtracer.cpp:533-534 (fixed_seed != 0)
	std::vector<Tile> tiles;
	for (uint32_t y = 0; y < H; y += th)
		for (uint32_t x = 0; x < W; x += tw)
			for (uint32_t s = 0; s < samples; s += ts)
				tiles.push_back(Tile{uint32_t(seeds()), x, std::min(x + tw, W), y, std::min(y + th, H), s, std::min(s + ts, samples)});
	std::stable_sort(tiles.begin(), tiles.end(), [&](const Tile& a, const Tile& b) {
		auto dist = [&](const Tile& t) {
			float dx = 0.5f * (t.x_begin + t.x_end) - W * 0.5f, dy = 0.5f * (t.y_begin + t.y_end) - H * 0.5f;
			return std::sqrt(dx * dx + dy * dy);
		};
		float da = dist(a), db = dist(b);
		if (da != db) return da < db;
		return a.s_begin < b.s_begin;
	});
	return tiles;
}

struct Accumulator {
	uint32_t w = 0, h = 0;
	std::vector<int64_t> rgb;      // 3 per pixel, 40.24 fixed point (pathtracer.h:80-83)
	std::vector<uint32_t> samples;
	void reset(uint32_t W, uint32_t H) { w = W, h = H, rgb.assign(size_t(W) * H * 3, 0), samples.assign(size_t(W) * H, 0); }
};

// rng_mode 0: one mt19937 per tile seeded with tile.seed (the reference, pathtracer.cpp:571).
// rng_mode 1: one Philox stream per (pixel, sample): key = (seed_lo, seed_hi), stream = (pixel index, sample index).
inline void do_trace_tile(const Pathtracer& pt, const Tile& tile, int rng_mode, uint64_t philox_seed, Accumulator& acc) {
	const uint32_t W = pt.camera.film_w;
	RNG tile_rng(tile.seed);
	for (uint32_t py = tile.y_begin; py < tile.y_end; ++py)
		for (uint32_t px = tile.x_begin; px < tile.x_end; ++px) {
			Spec sum;  // the tile-private HDR_Image pixel (pathtracer.cpp:447)
			for (uint32_t s = tile.s_begin; s < tile.s_end; ++s) {
				RNG local = RNG::philox(uint32_t(philox_seed), uint32_t(philox_seed >> 32), py * W + px, s);
				RNG& rng = rng_mode == 0 ? tile_rng : local;
				auto [ray, pdf] = pt.camera.sample_ray(rng, px, py, rng_mode == 0);
				ray.transform(pt.camera.camera_to_world);
				auto [emissive, light] = pt.trace(rng, ray);
				Spec p = (emissive + light) / pdf;
				if (p.valid()) sum = sum + p;
			}
			size_t idx = size_t(py) * acc.w + px;   // pathtracer.cpp:411-425
			acc.rgb[idx * 3 + 0] += int64_t(sum.x * (1ll << 24ll));
			acc.rgb[idx * 3 + 1] += int64_t(sum.y * (1ll << 24ll));
			acc.rgb[idx * 3 + 2] += int64_t(sum.z * (1ll << 24ll));
			acc.samples[idx] += (tile.s_end - tile.s_begin);
		}
}

inline void accumulator_to_image(const Accumulator& acc, float* rgb_out) {  // pathtracer.cpp:429-442
	for (size_t i = 0; i < acc.samples.size(); i++)
		for (int c = 0; c < 3; c++)
			rgb_out[i * 3 + c] = acc.samples[i] > 0 ? float(acc.rgb[i * 3 + c] / double(1ll << 24ll) / double(acc.samples[i])) : 0.0f;
}

inline float to_srgb(float f) {  // lib/spectrum.h:83-89
	if (f > 0.0031308f) return 1.055f * (std::pow(f, (1.0f / 2.4f))) - 0.055f;
	return f * 12.92f;
}
inline void tonemap(const float* rgb, size_t npix, float exposure, uint8_t* rgba) {  // util/hdr_image.cpp:201-225
	for (size_t i = 0; i < npix; i++) {
		for (int c = 0; c < 3; c++) {
			float v = 1.0f - std::exp(-rgb[i * 3 + c] * exposure);
			rgba[i * 4 + c] = static_cast<uint8_t>(std::round(to_srgb(v) * 255.0f));
		}
		rgba[i * 4 + 3] = 255;
	}
}

}  // namespace orc

// shade.hip.h -- device restatement of the reference's shading functions: RNG stream,
// camera rays, textures, BSDFs, lights, shading frame.  Same operations in the same order as
// the reference (citations inline); transcendental functions (sin, cos, atan2, acos, pow) come
// from the device math library and can differ from a host libm in the last bit.
#pragma once
#include <hip/hip_runtime.h>

#include "scene.h"
#include "traverse.hip.h"

namespace ptx {

constexpr float SH_PI = 3.14159265358979323846264338327950288f;   // PI_F, lib/mathlib.h:17
constexpr float SH_EPS = 0.00001f;                                 // EPS_F, lib/mathlib.h:16

struct V3f {
	float x, y, z;
};
__device__ __forceinline__ V3f v3(float x, float y, float z) { return V3f{x, y, z}; }
__device__ __forceinline__ V3f operator+(V3f a, V3f b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3f operator-(V3f a, V3f b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3f operator*(V3f a, V3f b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3f operator*(V3f a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3f operator*(float s, V3f a) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3f operator/(V3f a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ V3f operator-(V3f a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ float dot3(V3f l, V3f r) { return l.x * r.x + l.y * r.y + l.z * r.z; }   // vec3.h:215
__device__ __forceinline__ V3f cross3(V3f l, V3f r) { return {l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x}; }
__device__ __forceinline__ float norm(V3f a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3f unit(V3f a) {   // vec3.h:150
	const float n = norm(a);
	return {a.x / n, a.y / n, a.z / n};
}
__device__ __forceinline__ bool finite3(V3f a) { return isfinite(a.x) && isfinite(a.y) && isfinite(a.z); }

// ---- random numbers: Philox-4x32-10, key = render seed, counter = (block, pixel, sample, 0).
// Draw i of a path is word i%4 of block i/4; RNG::unit() = scalbn(float(u32), -32) (rand.cpp:17-21).
// One Philox-4x32-10 block (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"): counter c, key k.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
	for (int i = 0; i < 10; i++) {
		const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
		const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
		const uint32_t n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
		c0 = n0, c1 = n1, c2 = n2, c3 = n3;
		k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
	}
	out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

struct PathRng {
	uint32_t key0, key1, pixel, sample, draws;
	uint32_t buf[4];
	__device__ __forceinline__ void refill() { philox4x32_10(draws >> 2, pixel, sample, 0u, key0, key1, buf); }
	__device__ __forceinline__ void start(uint32_t k0, uint32_t k1, uint32_t px, uint32_t s, uint32_t d) {
		key0 = k0, key1 = k1, pixel = px, sample = s, draws = d;
		if (d & 3u) refill();   // resume in the middle of a block
	}
	__device__ __forceinline__ float next_unit() {
		if ((draws & 3u) == 0) refill();
		const uint32_t w = draws & 3u;
		const uint32_t v = w == 0 ? buf[0] : (w == 1 ? buf[1] : (w == 2 ? buf[2] : buf[3]));
		draws++;
		return __uint2float_rn(v) * 2.3283064365386963e-10f;   // * 2^-32, exact
	}
};

// ---- textures (scene/texture.cpp)
__device__ __forceinline__ V3f texel(const ShadeTables& sh, const TexRec& t, int x, int y) {
	const float* p = sh.texels + (size_t(t.pixel_base) + size_t(y) * t.w + x) * 3;
	return {p[0], p[1], p[2]};
}
__device__ __forceinline__ V3f tex_evaluate(const ShadeTables& sh, int id, float u, float v) {
	const TexRec& t = sh.textures[id];
	if (t.type == PTX_TEX_CONSTANT) return v3(t.color[0], t.color[1], t.color[2]) * t.scale;   // texture.cpp:216-218
	const float x = t.w * fminf(fmaxf(u, 0.0f), 1.0f), y = t.h * fminf(fmaxf(v, 0.0f), 1.0f);
	if (t.sampler == PTX_SAMPLER_NEAREST) {   // texture.cpp:9-24
		const int ix = min(int(floorf(x)), int(t.w) - 1), iy = min(int(floorf(y)), int(t.h) - 1);
		return texel(sh, t, ix, iy);
	}
	// texture.cpp:26-67 (lower index clamped at 0; the reference reads out of bounds there)
	const float i = floorf(x - 0.5f), j = floorf(y - 0.5f);
	const int x0 = max(0, min(int(i), int(t.w) - 1)), x1 = max(0, min(int(i + 1.0f), int(t.w) - 1));
	const int y0 = max(0, min(int(j), int(t.h) - 1)), y1 = max(0, min(int(j + 1.0f), int(t.h) - 1));
	const V3f lb = texel(sh, t, x0, y0), lt = texel(sh, t, x0, y1), rb = texel(sh, t, x1, y0), rt = texel(sh, t, x1, y1);
	const float s = x - (i + 0.5f), tt = y - (j + 0.5f);
	return (1 - tt) * ((1 - s) * lb + s * rb) + tt * ((1 - s) * lt + s * rt);
}

// ---- Shapes::Sphere::uv (scene/shape.cpp:7-12)
__device__ __forceinline__ void sphere_uv(V3f dir, float& u, float& v) {
	u = atan2f(dir.z, dir.x) / (2.0f * SH_PI);
	if (u < 0.0f) u += 1.0f;
	v = acosf(-1.0f * fminf(fmaxf(dir.y, -1.0f), 1.0f)) / SH_PI;
}

// ---- environment lights (scene/env_light.cpp:11-31), summed over all of them (pathtracer.cpp:130-136)
__device__ __forceinline__ V3f env_radiance(const ShadeTables& sh, V3f dir) {
	V3f sum = v3(0, 0, 0);
	for (uint32_t i = 0; i < sh.n_env_lights; i++) {
		const EnvLightRec& l = sh.env_lights[i];
		if (l.type == PTX_ENV_HEMISPHERE && dir.y < 0.0f) continue;   // adds a zero Spectrum in the reference
		float u, v;
		sphere_uv(dir, u, v);
		sum = sum + tex_evaluate(sh, l.tex, u, v);
	}
	return sum;
}

// ---- materials (scene/material.cpp)
__device__ __forceinline__ bool mat_is_specular(uint32_t type) { return type != PTX_MAT_LAMBERTIAN; }
__device__ __forceinline__ bool mat_is_sided(uint32_t type) { return type == PTX_MAT_REFRACT || type == PTX_MAT_GLASS; }

__device__ __forceinline__ V3f bsdf_reflect(V3f dir) {   // material.cpp:8-19
	V3f r = -dir;
	r.y = -r.y;
	return r;
}

__device__ __forceinline__ V3f bsdf_refract(V3f out_dir, float ior, bool& was_internal) {   // material.cpp:21-54
	const float cos_i = out_dir.x * 0.0f + out_dir.y * 1.0f + out_dir.z * 0.0f;   // dot(out_dir, (0,1,0))
	const bool entering = cos_i > 0;
	const float eta_i = entering ? 1.0f : ior, eta_t = entering ? ior : 1.0f;
	const float sin2_i = fmaxf(0.0f, 1.0f - cos_i * cos_i);
	const float sin2_t = eta_i * eta_i * sin2_i / (eta_t * eta_t);
	if (sin2_t >= 1) {
		was_internal = true;
		return v3(0, 0, 0);
	}
	const float cos_t = sqrtf(1 - sin2_t);
	V3f in_dir = unit(v3(-out_dir.x, 0.0f, -out_dir.z));
	in_dir = in_dir * sqrtf(sin2_t);
	in_dir.y = cos_t;
	if (0 < out_dir.y) in_dir.y = -in_dir.y;
	return in_dir;
}

// material.cpp:58-71.  `abs` resolves to abs(int) under g++ (PTX_FRESNEL_GCC_INT_ABS) and to the
// float overload elsewhere (PTX_FRESNEL_FLOAT_ABS); pow is evaluated in double.
__device__ __forceinline__ float bsdf_schlick(V3f in_dir, float ior, uint32_t mode) {
	const float d = in_dir.x * 0.0f + in_dir.y * 1.0f + in_dir.z * 0.0f;
	const float cos_theta = mode == PTX_FRESNEL_GCC_INT_ABS ? float(abs(int(d))) : fabsf(d);
	float r0 = (1 - ior) / (1 + ior);
	r0 = r0 * r0;
	return float(double(r0) + double(1.0f - r0) * pow(double(1 - cos_theta), 5.0));
}

__device__ __forceinline__ V3f bsdf_evaluate(const ShadeTables& sh, const MatRec& m, V3f in, float u, float v) {   // material.cpp:73-84
	if (m.type != PTX_MAT_LAMBERTIAN) return v3(0, 0, 0);
	const float cos_theta = fmaxf(0.0f, in.y);
	return tex_evaluate(sh, m.tex_a, u, v) / SH_PI * cos_theta;
}
__device__ __forceinline__ float bsdf_pdf(const MatRec& m, V3f in) {   // material.cpp:104-110 -> samplers.cpp:80-83
	if (m.type != PTX_MAT_LAMBERTIAN) return 0.0f;
	if (in.y < 0.0f) return 0.0f;
	return in.y / SH_PI;
}
__device__ __forceinline__ V3f bsdf_emission(const ShadeTables& sh, const MatRec& m, float u, float v) {   // material.cpp:313-315
	if (m.type != PTX_MAT_EMISSIVE) return v3(0, 0, 0);
	return tex_evaluate(sh, m.tex_a, u, v);
}

struct ScatterRec {
	V3f direction, attenuation;
	bool specular;
};

__device__ __forceinline__ ScatterRec bsdf_scatter(const ShadeTables& sh, const MatRec& m, PathRng& rng, V3f out, float u, float v) {
	ScatterRec ret;
	ret.direction = v3(0, 0, 0), ret.attenuation = v3(0, 0, 0), ret.specular = true;
	switch (m.type) {
	case PTX_MAT_LAMBERTIAN: {   // material.cpp:86-102, samplers.cpp:67-78
		const float phi = rng.next_unit() * 2.0f * SH_PI;
		const float cos_t = sqrtf(rng.next_unit());
		const float sin_t = sqrtf(1 - cos_t * cos_t);
		float sin_phi, cos_phi;
		sincosf(phi, &sin_phi, &cos_phi);   // the same values as sinf / cosf, one argument reduction for both
		ret.direction = v3(cos_phi * sin_t, cos_t, sin_phi * sin_t);
		ret.attenuation = bsdf_evaluate(sh, m, ret.direction, u, v);
		ret.specular = false;
		break;
	}
	case PTX_MAT_MIRROR:   // material.cpp:128-140
		ret.direction = bsdf_reflect(out);
		ret.attenuation = tex_evaluate(sh, m.tex_a, u, v);
		break;
	case PTX_MAT_REFRACT:     // material.cpp:162-189
	case PTX_MAT_GLASS: {     // material.cpp:223-266
		bool internal = false;
		ret.direction = unit(bsdf_refract(out, m.ior, internal));
		bool reflect_it = internal;
		if (m.type == PTX_MAT_GLASS) {
			const float fresnel = bsdf_schlick(out, m.ior, sh.fresnel_mode);
			if (!internal) reflect_it = rng.next_unit() < fresnel;   // `was_internal || rng.coin_flip(fresnel)`: no draw when internal
		}
		if (reflect_it) {
			ret.direction = unit(bsdf_reflect(out));
			ret.attenuation = m.type == PTX_MAT_GLASS ? tex_evaluate(sh, m.tex_a, u, v) : v3(1.0f, 1.0f, 1.0f);
		} else {
			const bool entering = (out.x * 0.0f + out.y * 1.0f + out.z * 0.0f) > 0;
			const float ratio = entering ? 1.0f / m.ior : m.ior;
			ret.attenuation = tex_evaluate(sh, m.tex_b, u, v) * (ratio * ratio);
		}
		break;
	}
	default: break;   // Emissive: material.cpp:301-307
	}
	return ret;
}

// ---- delta lights (scene/delta_light.cpp:7-40, Light_Instance::incoming instance.h:80-85)
struct IncomingRec {
	V3f radiance, direction;
	float distance;
};
__device__ __forceinline__ IncomingRec light_incoming(const DeltaLightRec& l, V3f from) {
	if (l.has_transform) mat_point(l.iT, from.x, from.y, from.z, from.x, from.y, from.z);
	IncomingRec r;
	const V3f color = v3(l.color[0], l.color[1], l.color[2]);
	if (l.type == PTX_LIGHT_DIRECTIONAL) {
		r.direction = v3(0.0f, -1.0f, 0.0f);
		r.distance = INFINITY;
		r.radiance = color * l.intensity;
	} else {
		r.direction = -unit(from);
		r.distance = norm(from);
		r.radiance = color * l.intensity;
		if (l.type == PTX_LIGHT_SPOT) {
			float angle = atan2f(sqrtf(from.x * from.x + from.z * from.z), from.y);
			angle = fabsf(angle * (180.0f / SH_PI));
			const float e0 = l.inner_angle / 2.0f, e1 = l.outer_angle / 2.0f;
			const float t = fminf(fmaxf((angle - e0) / (e1 - e0), 0.0f), 1.0f);
			const float ss = t * t * (3.0f - 2.0f * t);
			r.radiance = (1.0f - ss) * color * l.intensity;
		}
	}
	// Incoming::transform rotates the direction only (delta_light.h:29-31)
	if (l.has_transform) mat_rotate(l.T, r.direction.x, r.direction.y, r.direction.z, r.direction.x, r.direction.y, r.direction.z);
	return r;
}

// ---- shading frame: Mat4::rotate_to(normal) and its transpose (lib/mat4.h:379-393, pathtracer.cpp:151-153)
struct Frame {
	V3f x, y, z;   // columns of object_to_world
	__device__ __forceinline__ V3f to_world(V3f v) const {   // object_to_world.rotate(v)
		return {((v.x * x.x + v.y * y.x) + v.z * z.x) + 0.0f * 0.0f, ((v.x * x.y + v.y * y.y) + v.z * z.y) + 0.0f * 0.0f,
		        ((v.x * x.z + v.y * y.z) + v.z * z.z) + 0.0f * 0.0f};
	}
	__device__ __forceinline__ V3f to_local(V3f v) const {   // world_to_object.rotate(v), world_to_object = object_to_world.T()
		return {((v.x * x.x + v.y * x.y) + v.z * x.z) + 0.0f * 0.0f, ((v.x * y.x + v.y * y.y) + v.z * y.z) + 0.0f * 0.0f,
		        ((v.x * z.x + v.y * z.y) + v.z * z.z) + 0.0f * 0.0f};
	}
};
__device__ __forceinline__ Frame frame_from_normal(V3f n) {
	const V3f dir = unit(n);   // dir.normalize()
	Frame f;
	if (fabsf(dir.y - 1.0f) < SH_EPS) {
		f.x = v3(1, 0, 0), f.y = v3(0, 1, 0), f.z = v3(0, 0, 1);
	} else if (fabsf(dir.y + 1.0f) < SH_EPS) {
		f.x = v3(1, 0, 0), f.y = v3(0, -1, 0), f.z = v3(0, 0, 1);
	} else {
		f.x = unit(cross3(dir, v3(0.0f, 1.0f, 0.0f)));
		f.z = unit(cross3(f.x, dir));
		f.y = dir;
	}
	return f;
}

// ---- Camera::sample_ray + ray.transform(camera_to_world) (scene/camera.cpp:7-48, pathtracer.cpp:453-454)
struct CameraRec {
	float h, w;   // sensor plane size at z = -1: h = 2*tan(radians(vfov/2)), w = aspect*h (computed on the host)
	uint32_t film_w, film_h, max_depth, pad;
	float cam_to_world[16];
};
__device__ __forceinline__ Ray3 camera_ray(const CameraRec& c, uint32_t px, uint32_t py, float ox, float oy) {
	const float sx = float(px) + ox, sy = float(py) + oy;
	const float w_start = -0.5f * c.w, h_start = -0.5f * c.h;
	Ray3 r;
	r.ox = 0.0f, r.oy = 0.0f, r.oz = 0.0f;
	r.dx = (sx / c.film_w * c.w + w_start) - 0.0f;   // sensor point minus ray.point (the origin)
	r.dy = (sy / c.film_h * c.h + h_start) - 0.0f;
	r.dz = -1.0f - 0.0f;
	r.tmin = 0.0f, r.tmax = INFINITY;
	return ray_transform(c.cam_to_world, r);
}

}  // namespace ptx

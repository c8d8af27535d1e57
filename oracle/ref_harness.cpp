// oracle/ref_harness.cpp -- C entry points over the REAL reference sources.
//
// TEST INFRASTRUCTURE ONLY.  This file is compiled by oracle/Makefile together with the
// reference's own translation units, taken where they lie under /root/reference, into
// oracle/_ref/libptref.so.  Nothing from the reference is copied into this repository and
// no stand-in headers are used: only translation units that compile as they are with the
// image's g++ are linked (pathtracer.cpp and camera.cpp include SDL.h, which is absent, so
// Pathtracer::trace, the tile loop and Camera::sample_ray are not available here).
//
// It exists to (1) pin oracle/ref_cpu.h against the reference's own code and (2) generate the
// golden fixtures under tests/golden/ (tests/golden/make_golden.py).  The product never
// loads it.
#include "geometry/indexed.h"
#include "geometry/util.h"
#include "lib/mathlib.h"
#include "pathtracer/aggregate.h"
#include "pathtracer/bvh.h"
#include "pathtracer/instance.h"
#include "pathtracer/list.h"
#include "pathtracer/tri_mesh.h"
#include "scene/delta_light.h"
#include "scene/env_light.h"
#include "pathtracer/samplers.h"
#include "util/hdr_image.h"
#include "scene/material.h"
#include "scene/shape.h"
#include "scene/texture.h"
#include "util/rand.h"
#include "scene/io.h"
#include "scene/scene.h"
#include "scene/animator.h"

#include "../include/ptx.h"

#include <atomic>
#include <cstring>
#include <algorithm>
#include <map>
#include <memory>
#include <string>
#include <thread>

namespace {

struct RefMesh {
	std::vector<PT::Tri_Mesh_Vert> verts;  // 3 per triangle, expanded
	PT::BVH<PT::Triangle> bvh;
	PT::List<PT::Triangle> list;
	bool use_bvh = true;
	bool id_coded = false;
	PT::Trace hit(const Ray& r) const { return use_bvh ? bvh.hit(r) : list.hit(r); }
};

Vec3 id_code(uint32_t id) { return Vec3(float(id & 1023u), float((id >> 10) & 1023u), float(id >> 20)); }
uint32_t id_decode(Vec3 n) { return uint32_t(std::lround(n.x)) | (uint32_t(std::lround(n.y)) << 10) | (uint32_t(std::lround(n.z)) << 20); }

Mat4 mat_from(const float* t) {
	Mat4 m;
	for (int c = 0; c < 4; c++)
		for (int r = 0; r < 4; r++) m[c][r] = t[c * 4 + r];
	return m;
}

Ray ray_from(const PtxRay& p) {
	Ray r;  // bypass the normalising constructor: the batch holds constructed rays
	r.point = Vec3(p.o[0], p.o[1], p.o[2]);
	r.dir = Vec3(p.d[0], p.d[1], p.d[2]);
	r.dist_bounds = Vec2(p.tmin, p.tmax);
	return r;
}

template<typename F> void parallel_for(uint64_t n, int nthreads, F&& f) {
	if (nthreads <= 1 || n < 256) {
		f(0, n);
		return;
	}
	std::vector<std::thread> th;
	std::atomic<uint64_t> next{0};
	const uint64_t chunk = 4096;
	for (int t = 0; t < nthreads; t++)
		th.emplace_back([&] {
			for (;;) {
				uint64_t b = next.fetch_add(chunk);
				if (b >= n) break;
				f(b, std::min(n, b + chunk));
			}
		});
	for (auto& t : th) t.join();
}

struct RefScene {
	std::vector<std::unique_ptr<PT::Tri_Mesh>> meshes;
	std::vector<std::unique_ptr<Shape>> shapes;
	std::vector<std::shared_ptr<Texture>> textures;
	std::vector<std::unique_ptr<Material>> materials;
	PT::Aggregate scene;
};

}  // namespace

extern "C" {

// A BVH<Triangle> (or List<Triangle>) built by the reference's own builder over an indexed
// mesh (verts: 8 floats each).  With id_coded != 0 the vertex normals are replaced by a code
// of the triangle's input index so a hit can name its triangle (Trace has no primitive id).
void* ref_mesh_create(const float* verts, uint32_t n_verts, const uint32_t* idx, uint32_t n_tris, int use_bvh,
                      uint32_t max_leaf, int id_coded, const PtxBvhNode* inject_nodes, uint32_t n_inject) {
	(void)n_verts;
	auto* m = new RefMesh();
	m->use_bvh = use_bvh != 0;
	m->id_coded = id_coded != 0;
	m->verts.resize(size_t(n_tris) * 3);
	std::vector<PT::Triangle> tris;
	tris.reserve(n_tris);
	for (uint32_t i = 0; i < n_tris; i++) {
		for (int k = 0; k < 3; k++) {
			const float* v = verts + size_t(idx[i * 3 + k]) * 8;
			auto& o = m->verts[size_t(i) * 3 + k];
			o.position = Vec3(v[0], v[1], v[2]);
			o.normal = id_coded ? id_code(i) : Vec3(v[3], v[4], v[5]);
			o.uv = Vec2(v[6], v[7]);
		}
		tris.push_back(PT::Triangle(m->verts.data(), i * 3, i * 3 + 1, i * 3 + 2));
	}
	if (inject_nodes) {
		// BVH::nodes and BVH::primitives are public (bvh.h:53-55): install a tree built elsewhere.
		m->bvh.primitives = std::move(tris);
		m->bvh.nodes.resize(n_inject);
		for (uint32_t i = 0; i < n_inject; i++) {
			auto& n = m->bvh.nodes[i];
			const auto& s = inject_nodes[i];
			n.bbox = BBox(Vec3(s.bmin[0], s.bmin[1], s.bmin[2]), Vec3(s.bmax[0], s.bmax[1], s.bmax[2]));
			n.start = s.start, n.size = s.size, n.l = s.l, n.r = s.r;
		}
	} else if (m->use_bvh) {
		m->bvh.build(std::move(tris), max_leaf ? max_leaf : 4);
	} else {
		m->list = PT::List<PT::Triangle>(std::move(tris));
	}
	return m;
}
void ref_mesh_destroy(void* h) { delete static_cast<RefMesh*>(h); }

uint32_t ref_mesh_num_nodes(void* h) { return uint32_t(static_cast<RefMesh*>(h)->bvh.nodes.size()); }
void ref_mesh_get_nodes(void* h, PtxBvhNode* out) {
	const auto& nodes = static_cast<RefMesh*>(h)->bvh.nodes;
	for (size_t i = 0; i < nodes.size(); i++) {
		const auto& n = nodes[i];
		out[i] = PtxBvhNode{{n.bbox.min.x, n.bbox.min.y, n.bbox.min.z}, {n.bbox.max.x, n.bbox.max.y, n.bbox.max.z},
		                    uint32_t(n.start), uint32_t(n.size), uint32_t(n.l), uint32_t(n.r)};
	}
}
// Boxes of the BVH-ordered primitives (6 floats each): Triangle::bbox of BVH::primitives[i].
void ref_mesh_get_prim_boxes(void* h, float* out) {
	const auto& p = static_cast<RefMesh*>(h)->bvh.primitives;
	for (size_t i = 0; i < p.size(); i++) {
		BBox b = p[i].bbox();
		out[i * 6 + 0] = b.min.x, out[i * 6 + 1] = b.min.y, out[i * 6 + 2] = b.min.z;
		out[i * 6 + 3] = b.max.x, out[i * 6 + 4] = b.max.y, out[i * 6 + 5] = b.max.z;
	}
}

// Closest hit.  hits[i].prim is the triangle's INPUT index (id-coded meshes only, else 0 for
// a hit), PTX_NO_HIT for a miss.  pos / normal / uv are optional.
void ref_mesh_hit(void* h, const PtxRay* rays, uint64_t n, PtxHit* hits, float* pos, float* normal, float* uv, int nthreads) {
	auto* m = static_cast<RefMesh*>(h);
	parallel_for(n, nthreads, [&](uint64_t b, uint64_t e) {
		for (uint64_t i = b; i < e; i++) {
			PT::Trace t = m->hit(ray_from(rays[i]));
			hits[i].distance = t.distance;
			hits[i].prim = t.hit ? (m->id_coded ? id_decode(t.normal) : 0u) : PTX_NO_HIT;
			if (pos) pos[i * 3] = t.position.x, pos[i * 3 + 1] = t.position.y, pos[i * 3 + 2] = t.position.z;
			if (normal) normal[i * 3] = t.normal.x, normal[i * 3 + 1] = t.normal.y, normal[i * 3 + 2] = t.normal.z;
			if (uv) uv[i * 2] = t.uv.x, uv[i * 2 + 1] = t.uv.y;
		}
	});
}

// Two-level scene: Tri_Mesh per mesh (built by the reference, max_leaf 4), Instance per
// instance, Aggregate(BVH<Instance>) or Aggregate(List<Instance>) on top (pathtracer.cpp:381-385).
// Materials are default-constructed placeholders; a hit reports the material's index.
void* ref_scene_create(const PtxSceneDesc* d) {
	auto* s = new RefScene();
	for (uint32_t i = 0; i < d->n_meshes; i++) {
		const auto& md = d->meshes[i];
		std::vector<Indexed_Mesh::Vert> verts(md.n_verts);
		for (uint32_t v = 0; v < md.n_verts; v++) {
			const float* p = md.verts + size_t(v) * 8;
			verts[v] = Indexed_Mesh::Vert{Vec3(p[0], p[1], p[2]), Vec3(p[3], p[4], p[5]), Vec2(p[6], p[7]), v};
		}
		std::vector<Indexed_Mesh::Index> inds(md.indices, md.indices + size_t(md.n_tris) * 3);
		s->meshes.push_back(std::make_unique<PT::Tri_Mesh>(Indexed_Mesh(std::move(verts), std::move(inds)), md.use_bvh != 0));
	}
	for (uint32_t i = 0; i <= d->n_materials; i++) s->materials.push_back(std::make_unique<Material>());
	std::vector<PT::Instance> objects;
	for (uint32_t i = 0; i < d->n_instances; i++) {
		const auto& in = d->instances[i];
		Material* mat = s->materials[in.material < 0 ? d->n_materials : uint32_t(in.material)].get();
		if (in.mesh >= 0) objects.emplace_back(s->meshes[size_t(in.mesh)].get(), mat, mat_from(in.T));
		else {
			s->shapes.push_back(std::make_unique<Shape>(Shapes::Sphere(in.sphere_radius)));
			objects.emplace_back(s->shapes.back().get(), mat, mat_from(in.T));
		}
	}
	if (d->use_bvh) s->scene = PT::Aggregate(PT::BVH<PT::Instance>(std::move(objects)));
	else s->scene = PT::Aggregate(PT::List<PT::Instance>(std::move(objects)));
	return s;
}
void ref_scene_destroy(void* h) { delete static_cast<RefScene*>(h); }

void ref_scene_hit(void* h, const PtxRay* rays, uint64_t n, uint8_t* hit, float* distance, float* pos, float* normal, float* uv,
                   int32_t* material, int nthreads) {
	auto* s = static_cast<RefScene*>(h);
	parallel_for(n, nthreads, [&](uint64_t b, uint64_t e) {
		for (uint64_t i = b; i < e; i++) {
			PT::Trace t = s->scene.hit(ray_from(rays[i]));
			hit[i] = t.hit ? 1 : 0;
			distance[i] = t.distance;
			pos[i * 3] = t.position.x, pos[i * 3 + 1] = t.position.y, pos[i * 3 + 2] = t.position.z;
			normal[i * 3] = t.normal.x, normal[i * 3 + 1] = t.normal.y, normal[i * 3 + 2] = t.normal.z;
			uv[i * 2] = t.uv.x, uv[i * 2 + 1] = t.uv.y;
			int32_t mi = -1;
			for (size_t k = 0; k < s->materials.size(); k++)
				if (s->materials[k].get() == t.material) mi = int32_t(k);
			material[i] = mi;
		}
	});
}

// ---- single-function probes ---------------------------------------------------------------
int ref_triangle_hit(const float* verts24, const PtxRay* ray, float* out12) {
	PT::Tri_Mesh_Vert v[3];
	for (int i = 0; i < 3; i++) {
		const float* p = verts24 + i * 8;
		v[i] = PT::Tri_Mesh_Vert{Vec3(p[0], p[1], p[2]), Vec3(p[3], p[4], p[5]), Vec2(p[6], p[7])};
	}
	PT::Triangle t(v, 0, 1, 2);
	PT::Trace tr = t.hit(ray_from(*ray));
	float o[12] = {tr.distance, tr.position.x, tr.position.y, tr.position.z, tr.normal.x, tr.normal.y, tr.normal.z, tr.uv.x, tr.uv.y, 0, 0, 0};
	std::memcpy(out12, o, sizeof(o));
	return tr.hit ? 1 : 0;
}

void ref_triangle_bbox(const float* verts24, float* out6) {
	PT::Tri_Mesh_Vert v[3];
	for (int i = 0; i < 3; i++) v[i].position = Vec3(verts24[i * 8], verts24[i * 8 + 1], verts24[i * 8 + 2]);
	PT::Triangle t(v, 0, 1, 2);
	BBox b = t.bbox();
	out6[0] = b.min.x, out6[1] = b.min.y, out6[2] = b.min.z, out6[3] = b.max.x, out6[4] = b.max.y, out6[5] = b.max.z;
}

int ref_bbox_hit(const float* box6, const PtxRay* ray, float* times2) {
	BBox b(Vec3(box6[0], box6[1], box6[2]), Vec3(box6[3], box6[4], box6[5]));
	Vec2 t(times2[0], times2[1]);
	bool h = b.hit(ray_from(*ray), t);
	times2[0] = t.x, times2[1] = t.y;
	return h ? 1 : 0;
}

int ref_sphere_hit(float radius, const PtxRay* ray, float* out9) {
	Shapes::Sphere s(radius);
	PT::Trace tr = s.hit(ray_from(*ray));
	float o[9] = {tr.distance, tr.position.x, tr.position.y, tr.position.z, tr.normal.x, tr.normal.y, tr.normal.z, tr.uv.x, tr.uv.y};
	std::memcpy(out9, o, sizeof(o));
	return tr.hit ? 1 : 0;
}

void ref_material_scatter(uint32_t type, const float* color_a, const float* color_b, float ior, uint32_t seed, const float* out_dirs3,
                          uint32_t n, float* out7) {
	auto ta = std::make_shared<Texture>(Textures::Constant{Spectrum(color_a[0], color_a[1], color_a[2]), 1.0f});
	auto tb = std::make_shared<Texture>(Textures::Constant{Spectrum(color_b[0], color_b[1], color_b[2]), 1.0f});
	Material m;
	switch (type) {
	case PTX_MAT_LAMBERTIAN: m = Material(Materials::Lambertian{ta}); break;
	case PTX_MAT_MIRROR: {
		Materials::Mirror x;
		x.reflectance = ta;
		m = Material(x);
		break;
	}
	case PTX_MAT_REFRACT: {
		Materials::Refract x;
		x.transmittance = tb, x.ior = ior;
		m = Material(x);
		break;
	}
	case PTX_MAT_GLASS: {
		Materials::Glass x;
		x.reflectance = ta, x.transmittance = tb, x.ior = ior;
		m = Material(x);
		break;
	}
	default: {
		Materials::Emissive x;
		x.emissive = ta;
		m = Material(x);
		break;
	}
	}
	RNG rng(seed);
	for (uint32_t i = 0; i < n; i++) {
		Materials::Scatter s = m.scatter(rng, Vec3(out_dirs3[i * 3], out_dirs3[i * 3 + 1], out_dirs3[i * 3 + 2]), Vec2{});
		float* w = out7 + size_t(i) * 7;
		w[0] = s.direction.x, w[1] = s.direction.y, w[2] = s.direction.z;
		w[3] = s.attenuation.r, w[4] = s.attenuation.g, w[5] = s.attenuation.b, w[6] = s.specular ? 1.0f : 0.0f;
	}
}

void ref_mat4_inverse(const float* in16, float* out16) {
	Mat4 r = mat_from(in16).inverse();
	for (int c = 0; c < 4; c++)
		for (int k = 0; k < 4; k++) out16[c * 4 + k] = r[c][k];
}

void ref_rotate_to(const float* dir3, float* out16) {
	Mat4 r = Mat4::rotate_to(Vec3(dir3[0], dir3[1], dir3[2]));
	for (int c = 0; c < 4; c++)
		for (int k = 0; k < 4; k++) out16[c * 4 + k] = r[c][k];
}

void ref_rng_units(uint32_t seed, uint32_t n, float* out) {
	RNG r(seed);
	for (uint32_t i = 0; i < n; i++) out[i] = r.unit();
}

// Environment_Lights::Sphere::evaluate over an image texture (w*h*3 floats)
void ref_env_sphere_evaluate(uint32_t w, uint32_t h, const float* pixels, int sampler, const float* dirs3, uint32_t n, float* out3) {
	std::vector<Spectrum> px(size_t(w) * h);
	for (size_t i = 0; i < px.size(); i++) px[i] = Spectrum(pixels[i * 3], pixels[i * 3 + 1], pixels[i * 3 + 2]);
	auto sm = sampler == 0 ? Textures::Image::Sampler::nearest : Textures::Image::Sampler::bilinear;
	auto tex = std::make_shared<Texture>(Texture{Textures::Image{sm, HDR_Image(w, h, px)}});
	Environment_Lights::Sphere map;
	map.radiance = tex;
	for (uint32_t i = 0; i < n; i++) {
		Spectrum s = map.evaluate(Vec3(dirs3[i * 3], dirs3[i * 3 + 1], dirs3[i * 3 + 2]));
		out3[i * 3] = s.r, out3[i * 3 + 1] = s.g, out3[i * 3 + 2] = s.b;
	}
}

// HDR_Image::tonemap_to (util/hdr_image.cpp:201-225): w*h RGB floats -> w*h RGBA bytes.
void ref_tonemap(uint32_t w, uint32_t h, const float* rgb, float exposure, uint8_t* rgba) {
	std::vector<Spectrum> px(size_t(w) * h);
	for (size_t i = 0; i < px.size(); i++) px[i] = Spectrum(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]);
	HDR_Image img(w, h, px);
	std::vector<uint8_t> data;
	img.tonemap_to(data, exposure);
	std::memcpy(rgba, data.data(), data.size());
}

// Delta_Lights::{Point,Directional,Spot}::incoming (scene/delta_light.cpp:7-40) through PT::Light_Instance::incoming
// (pathtracer/instance.h:74-91) with local_to_world T16 (null: identity).  out7 per point: radiance rgb, direction xyz, distance.
void ref_delta_incoming(uint32_t type, const float* color3, float intensity, float inner, float outer, const float* T16, const float* pts3,
                        uint32_t n, float* out7) {
	Delta_Light light;
	const Spectrum color(color3[0], color3[1], color3[2]);
	if (type == PTX_LIGHT_POINT) {
		Delta_Lights::Point l;
		l.color = color, l.intensity = intensity;
		light.light = l;
	} else if (type == PTX_LIGHT_DIRECTIONAL) {
		Delta_Lights::Directional l;
		l.color = color, l.intensity = intensity;
		light.light = l;
	} else {
		Delta_Lights::Spot l;
		l.color = color, l.intensity = intensity, l.inner_angle = inner, l.outer_angle = outer;
		light.light = l;
	}
	PT::Light_Instance inst(&light, T16 ? mat_from(T16) : Mat4::I);
	for (uint32_t i = 0; i < n; i++) {
		Delta_Lights::Incoming in = inst.incoming(Vec3(pts3[i * 3], pts3[i * 3 + 1], pts3[i * 3 + 2]));
		float* o = out7 + size_t(i) * 7;
		o[0] = in.radiance.r, o[1] = in.radiance.g, o[2] = in.radiance.b;
		o[3] = in.direction.x, o[4] = in.direction.y, o[5] = in.direction.z, o[6] = in.distance;
	}
}

// Samplers::Rect::sample / pdf (pathtracer/samplers.cpp:7-19): n samples from RNG(seed); out3 = x, y, pdf(sample).
void ref_sampler_rect(uint32_t seed, float sx, float sy, uint32_t n, float* out3) {
	RNG rng(seed);
	Samplers::Rect r(Vec2(sx, sy));
	for (uint32_t i = 0; i < n; i++) {
		Vec2 v = r.sample(rng);
		out3[i * 3] = v.x, out3[i * 3 + 1] = v.y, out3[i * 3 + 2] = r.pdf(v);
	}
}

// Samplers::Hemisphere::Cosine::sample / pdf (pathtracer/samplers.cpp:67-83): out4 = direction xyz, pdf(direction).
void ref_sampler_cosine(uint32_t seed, uint32_t n, float* out4) {
	RNG rng(seed);
	Samplers::Hemisphere::Cosine c;
	for (uint32_t i = 0; i < n; i++) {
		Vec3 d = c.sample(rng);
		out4[i * 4] = d.x, out4[i * 4 + 1] = d.y, out4[i * 4 + 2] = d.z, out4[i * 4 + 3] = c.pdf(d);
	}
}

// Util::sphere_mesh(r, subdivisions) (geometry/util.cpp): returns counts; fills when non-null.
void ref_sphere_mesh(float r, uint32_t subdiv, uint32_t* n_verts, uint32_t* n_tris, float* verts8, uint32_t* idx) {
	Indexed_Mesh m = Util::sphere_mesh(r, subdiv);
	*n_verts = uint32_t(m.vertices().size());
	*n_tris = m.tris();
	if (verts8)
		for (size_t i = 0; i < m.vertices().size(); i++) {
			const auto& v = m.vertices()[i];
			float* o = verts8 + i * 8;
			o[0] = v.pos.x, o[1] = v.pos.y, o[2] = v.pos.z, o[3] = v.norm.x, o[4] = v.norm.y, o[5] = v.norm.z, o[6] = v.uv.x, o[7] = v.uv.y;
		}
	if (idx) std::copy(m.indices().begin(), m.indices().end(), idx);
}

// The generator of tests/a3/test.a3.task3.bvh.fuzz.cpp:13-44 driven by the reference RNG:
// first the triangle count n = uint32(5000 * (unit() + 0.5)), then n triangles, then rays.
// Returns n; fills verts8 (3 per triangle) when non-null, then n_rays rays (Ray's ctor applied).
uint32_t ref_fuzz_mesh(uint32_t seed, float* verts8, uint32_t cap_tris, PtxRay* rays, uint32_t n_rays) {
	RNG gen(seed);
	uint32_t n = static_cast<uint32_t>(5000 * (gen.unit() + 0.5f));
	if (!verts8) return n;
	if (n > cap_tris) return n;
	auto v = [&]() { return Vec3{gen.unit(), gen.unit(), gen.unit()} * 1.0f; };
	auto t = [&]() { return Vec3{gen.unit(), gen.unit(), gen.unit()} * 10.0f; };
	for (uint32_t i = 0; i < n; i++) {
		Vec3 o = t();
		Vec3 p[3];
		p[0] = v() + o;
		p[1] = v() + o;
		p[2] = v() + o;
		for (int k = 0; k < 3; k++) {
			float* w = verts8 + (size_t(i) * 3 + k) * 8;
			w[0] = p[k].x, w[1] = p[k].y, w[2] = p[k].z, w[3] = 0, w[4] = 1, w[5] = 0, w[6] = 0, w[7] = 0;
		}
	}
	for (uint32_t j = 0; j < n_rays; j++) {
		Ray r{Vec3{gen.unit(), gen.unit(), gen.unit()} * 10.0f, Vec3{gen.unit(), gen.unit(), gen.unit()}};
		rays[j] = PtxRay{{r.point.x, r.point.y, r.point.z}, {r.dir.x, r.dir.y, r.dir.z}, r.dist_bounds.x, r.dist_bounds.y};
	}
	return n;
}

// Ray's constructor (ray.h:16-20) and Ray::transform (ray.h:32-38)
void ref_ray_make(const float* o3, const float* d3, const float* T16, PtxRay* out) {
	Ray r(Vec3(o3[0], o3[1], o3[2]), Vec3(d3[0], d3[1], d3[2]));
	if (T16) r.transform(mat_from(T16));
	*out = PtxRay{{r.point.x, r.point.y, r.point.z}, {r.dir.x, r.dir.y, r.dir.z}, r.dist_bounds.x, r.dist_bounds.y};
}

// ---- scene files: the reference's own loader (scene/io.cpp:15-70) and mesh flattening
// (Indexed_Mesh::from_halfedge_mesh(..., SplitEdges), geometry/indexed.cpp:10-34; Transform::local_to_world,
// scene/transform.cpp:4-24), serialised to a flat byte stream:
//   u32 n_meshes        { str name, u32 n_verts, u32 n_indices, f32 verts[n_verts*8] (pos, normal, uv), u32 indices[] }
//   u32 n_mesh_insts    { str name, str mesh, str material ("" = none), u32 visible, f32 T[16] }
//   u32 n_shape_insts   { str name, f32 radius, str material, u32 visible, f32 T[16] }
//   u32 n_camera_insts  { str name, f32 T[16], f32 vfov, f32 aspect, u32 w, u32 h, u32 samples, u32 depth }
//   u32 n_light_insts   { str name, u32 type (0 point, 1 directional, 2 spot), f32 color[3], f32 intensity, f32 inner, f32 outer, u32 visible, f32 T[16] }
// str = u32 length + bytes.  Names in sorted order.  Returns the stream's size; writes it if it fits `cap`.
uint64_t ref_js3d_flatten(const char* path, uint8_t* out, uint64_t cap) {
	Scene scene;
	Animator animator;
	try {
		load(path, &scene, &animator);
	} catch (std::exception const&) {
		return 0;
	}
	std::vector<uint8_t> buf;
	auto put = [&](const void* p, size_t n) { buf.insert(buf.end(), static_cast<const uint8_t*>(p), static_cast<const uint8_t*>(p) + n); };
	auto u32 = [&](uint32_t v) { put(&v, 4); };
	auto f32 = [&](float v) { put(&v, 4); };
	auto str = [&](const std::string& v) { u32(uint32_t(v.size())), put(v.data(), v.size()); };
	auto mat = [&](const Mat4& m) { for (int c = 0; c < 4; c++) for (int k = 0; k < 4; k++) f32(m[c][k]); };
	auto sorted = [](auto const& storage) {
		std::vector<std::string> names;
		for (auto const& kv : storage) names.push_back(kv.first);
		std::sort(names.begin(), names.end());
		return names;
	};
	u32(uint32_t(scene.meshes.size()));
	for (auto const& name : sorted(scene.meshes)) {
		Indexed_Mesh im = Indexed_Mesh::from_halfedge_mesh(*scene.meshes.at(name), Indexed_Mesh::SplitEdges);
		str(name), u32(uint32_t(im.vertices().size())), u32(uint32_t(im.indices().size()));
		for (auto const& v : im.vertices()) f32(v.pos.x), f32(v.pos.y), f32(v.pos.z), f32(v.norm.x), f32(v.norm.y), f32(v.norm.z), f32(v.uv.x), f32(v.uv.y);
		for (auto i : im.indices()) u32(i);
	}
	auto name_of = [&](auto const& storage, auto const& weak) -> std::string {
		auto sp = weak.lock();
		if (!sp) return "";
		for (auto const& kv : storage) if (kv.second == sp) return kv.first;
		return "";
	};
	u32(uint32_t(scene.instances.meshes.size()));
	for (auto const& name : sorted(scene.instances.meshes)) {
		auto const& in = *scene.instances.meshes.at(name);
		str(name), str(name_of(scene.meshes, in.mesh)), str(name_of(scene.materials, in.material)), u32(in.settings.visible ? 1 : 0);
		mat(in.transform.lock()->local_to_world());
	}
	u32(uint32_t(scene.instances.shapes.size()));
	for (auto const& name : sorted(scene.instances.shapes)) {
		auto const& in = *scene.instances.shapes.at(name);
		auto shape = in.shape.lock();
		str(name), f32(shape ? std::get<Shapes::Sphere>(shape->shape).radius : 0.0f), str(name_of(scene.materials, in.material)), u32(in.settings.visible ? 1 : 0);
		mat(in.transform.lock()->local_to_world());
	}
	u32(uint32_t(scene.instances.cameras.size()));
	for (auto const& name : sorted(scene.instances.cameras)) {
		auto const& in = *scene.instances.cameras.at(name);
		auto cam = in.camera.lock();
		str(name), mat(in.transform.lock()->local_to_world());
		f32(cam->vertical_fov), f32(cam->aspect_ratio), u32(cam->film.width), u32(cam->film.height), u32(cam->film.samples), u32(cam->film.max_ray_depth);
	}
	u32(uint32_t(scene.instances.delta_lights.size()));
	for (auto const& name : sorted(scene.instances.delta_lights)) {
		auto const& in = *scene.instances.delta_lights.at(name);
		auto l = in.light.lock();
		str(name);
		if (l->is<Delta_Lights::Point>()) {
			auto const& p = std::get<Delta_Lights::Point>(l->light);
			u32(0), f32(p.color.r), f32(p.color.g), f32(p.color.b), f32(p.intensity), f32(0), f32(0);
		} else if (l->is<Delta_Lights::Directional>()) {
			auto const& p = std::get<Delta_Lights::Directional>(l->light);
			u32(1), f32(p.color.r), f32(p.color.g), f32(p.color.b), f32(p.intensity), f32(0), f32(0);
		} else {
			auto const& p = std::get<Delta_Lights::Spot>(l->light);
			u32(2), f32(p.color.r), f32(p.color.g), f32(p.color.b), f32(p.intensity), f32(p.inner_angle), f32(p.outer_angle);
		}
		u32(in.settings.visible ? 1 : 0), mat(in.transform.lock()->local_to_world());
	}
	if (out && buf.size() <= cap) std::memcpy(out, buf.data(), buf.size());
	return buf.size();
}

}  // extern "C"

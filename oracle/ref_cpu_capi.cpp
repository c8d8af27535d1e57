// oracle/ref_cpu_capi.cpp -- C entry points over oracle/ref_cpu.h for ctypes.
//
// TEST INFRASTRUCTURE ONLY (see the header of ref_cpu.h): loaded by tests/, by
// __graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the product.
// It accepts the same POD scene description as the product's C ABI (include/ptx.h) so a
// test can hand one description to both sides.
#include "ref_cpu.h"

#include "../include/ptx.h"

#include <atomic>
#include <thread>

using namespace orc;

namespace {

M4 mat_from(const float* t) {
	M4 m;
	for (int c = 0; c < 4; c++)
		for (int r = 0; r < 4; r++) m.m[c][r] = t[c * 4 + r];
	return m;
}

struct OScene {
	Scene scene;
	bool mesh_only = false;  // queries are Tri_Mesh::hit on meshes[0]
	Trace hit(const Ray& r, Counters* c) const {
		if (!mesh_only) return scene.hit(r, c);
		Trace t = scene.meshes[0]->hit(r, c);
		if (t.hit) t.inst = 0;
		return t;
	}
};

TriMesh* make_mesh(const PtxMeshDesc& d) {
	auto* m = new TriMesh();
	size_t max_leaf = d.max_leaf_size ? d.max_leaf_size : 4;
	if (d.nodes) {
		// Injected tree: triangles are already in BVH order.
		m->use_bvh = true;
		m->verts.resize(d.n_verts);
		for (uint32_t i = 0; i < d.n_verts; i++) {
			const float* v = d.verts + size_t(i) * 8;
			m->verts[i] = Vert{{v[0], v[1], v[2]}, {v[3], v[4], v[5]}, {v[6], v[7]}};
		}
		m->bvh.primitives.resize(d.n_tris);
		for (uint32_t i = 0; i < d.n_tris; i++) {
			auto& t = m->bvh.primitives[i];
			t.v0 = d.indices[i * 3], t.v1 = d.indices[i * 3 + 1], t.v2 = d.indices[i * 3 + 2];
			t.vertex_list = m->verts.data();
			t.id = i;
		}
		m->bvh.nodes.resize(d.n_nodes);
		for (uint32_t i = 0; i < d.n_nodes; i++) {
			auto& n = m->bvh.nodes[i];
			const auto& s = d.nodes[i];
			n.bbox.min = {s.bmin[0], s.bmin[1], s.bmin[2]};
			n.bbox.max = {s.bmax[0], s.bmax[1], s.bmax[2]};
			n.start = s.start, n.size = s.size, n.l = s.l, n.r = s.r;
		}
	} else {
		m->build(d.verts, d.n_verts, d.indices, d.n_tris, d.use_bvh != 0, max_leaf);
	}
	return m;
}

template<typename F> void parallel_for(uint64_t n, int nthreads, F&& f) {
	if (nthreads <= 1 || n < 256) {
		f(0, n, 0);
		return;
	}
	std::vector<std::thread> th;
	std::atomic<uint64_t> next{0};
	const uint64_t chunk = 4096;
	for (int t = 0; t < nthreads; t++)
		th.emplace_back([&, t] {
			for (;;) {
				uint64_t b = next.fetch_add(chunk);
				if (b >= n) break;
				f(b, std::min(n, b + chunk), t);
			}
		});
	for (auto& t : th) t.join();
}

}  // namespace

extern "C" {

void* orc_scene_create(const PtxSceneDesc* d) {
	auto* s = new OScene();
	Scene& sc = s->scene;
	for (uint32_t i = 0; i < d->n_meshes; i++) sc.meshes.push_back(make_mesh(d->meshes[i]));
	sc.mats.fresnel_mode = int(d->fresnel_mode);
	for (uint32_t i = 0; i < d->n_textures; i++) {
		const auto& t = d->textures[i];
		Texture x;
		x.type = int(t.type), x.color = {t.color[0], t.color[1], t.color[2]}, x.scale = t.scale;
		x.sampler = int(t.sampler), x.w = t.w, x.h = t.h;
		if (t.type == PTX_TEX_IMAGE) {
			x.pixels.resize(size_t(t.w) * t.h);
			for (size_t p = 0; p < x.pixels.size(); p++) x.pixels[p] = {t.pixels[p * 3], t.pixels[p * 3 + 1], t.pixels[p * 3 + 2]};
		}
		sc.mats.textures.push_back(std::move(x));
	}
	// default texture + default material (pathtracer.cpp:240-252)
	int default_tex = int(sc.mats.textures.size());
	{
		Texture x;
		x.type = 0, x.color = {0, 0, 0}, x.scale = 1.0f;
		sc.mats.textures.push_back(x);
	}
	for (uint32_t i = 0; i < d->n_materials; i++) {
		const auto& m = d->materials[i];
		Material x;
		x.type = int(m.type), x.tex_a = m.tex_a, x.tex_b = m.tex_b, x.ior = m.ior;
		sc.mats.materials.push_back(x);
	}
	int default_mat = int(sc.mats.materials.size());
	{
		Material x;
		x.type = LAMBERTIAN, x.tex_a = default_tex;
		sc.mats.materials.push_back(x);
	}
	sc.spheres.reserve(d->n_instances);
	for (uint32_t i = 0; i < d->n_instances; i++) {
		const auto& in = d->instances[i];
		Instance x;
		x.set(mat_from(in.T));
		x.material = in.material < 0 ? default_mat : in.material;
		if (in.mesh >= 0) x.mesh = sc.meshes[size_t(in.mesh)];
		else {
			Sphere sp;
			sp.radius = in.sphere_radius;
			sc.spheres.push_back(sp);
			x.sphere = &sc.spheres.back();
		}
		sc.instances_in.push_back(x);
	}
	for (uint32_t i = 0; i < d->n_delta_lights; i++) {
		const auto& l = d->delta_lights[i];
		DeltaLight x;
		x.type = int(l.type), x.color = {l.color[0], l.color[1], l.color[2]}, x.intensity = l.intensity;
		x.inner_angle = l.inner_angle, x.outer_angle = l.outer_angle;
		x.set(mat_from(l.T));
		sc.delta_lights.push_back(x);
	}
	for (uint32_t i = 0; i < d->n_env_lights; i++) sc.env_lights.push_back(EnvLight{int(d->env_lights[i].type), d->env_lights[i].tex});
	sc.use_bvh = d->use_bvh != 0;
	sc.build_top();
	return s;
}

void* orc_scene_create_mesh(const PtxMeshDesc* d) {
	auto* s = new OScene();
	s->mesh_only = true;
	s->scene.meshes.push_back(make_mesh(*d));
	return s;
}

void orc_scene_destroy(void* h) { delete static_cast<OScene*>(h); }

uint32_t orc_scene_mesh_num_nodes(void* h, uint32_t mesh) { return uint32_t(static_cast<OScene*>(h)->scene.meshes[mesh]->bvh.nodes.size()); }

void orc_scene_mesh_get_nodes(void* h, uint32_t mesh, PtxBvhNode* out) {
	const auto& nodes = static_cast<OScene*>(h)->scene.meshes[mesh]->bvh.nodes;
	for (size_t i = 0; i < nodes.size(); i++) {
		const auto& n = nodes[i];
		out[i] = PtxBvhNode{{n.bbox.min.x, n.bbox.min.y, n.bbox.min.z}, {n.bbox.max.x, n.bbox.max.y, n.bbox.max.z},
		                    uint32_t(n.start), uint32_t(n.size), uint32_t(n.l), uint32_t(n.r)};
	}
}

void orc_scene_mesh_get_order(void* h, uint32_t mesh, uint32_t* out) {
	const auto& p = static_cast<OScene*>(h)->scene.meshes[mesh]->prims();
	for (size_t i = 0; i < p.size(); i++) out[i] = p[i].id;
}

void orc_scene_mesh_get_stats(void* h, uint32_t mesh, uint64_t* out) {
	const auto& b = static_cast<OScene*>(h)->scene.meshes[mesh]->bvh;
	out[0] = b.clamped, out[1] = b.predicate_disagreements, out[2] = b.dropped;
}

// The same counters for the instance-level tree (all zero for a List<Instance> scene).
void orc_scene_top_get_stats(void* h, uint64_t* out) {
	const auto& b = static_cast<OScene*>(h)->scene.top;
	out[0] = b.clamped, out[1] = b.predicate_disagreements, out[2] = b.dropped;
}

uint32_t orc_scene_top_num_nodes(void* h) { return uint32_t(static_cast<OScene*>(h)->scene.top.nodes.size()); }

void orc_scene_top_get_nodes(void* h, PtxBvhNode* out) {
	const auto& nodes = static_cast<OScene*>(h)->scene.top.nodes;
	for (size_t i = 0; i < nodes.size(); i++) {
		const auto& n = nodes[i];
		out[i] = PtxBvhNode{{n.bbox.min.x, n.bbox.min.y, n.bbox.min.z}, {n.bbox.max.x, n.bbox.max.y, n.bbox.max.z},
		                    uint32_t(n.start), uint32_t(n.size), uint32_t(n.l), uint32_t(n.r)};
	}
}

void orc_scene_get_instance_order(void* h, uint32_t* out) {
	auto* s = static_cast<OScene*>(h);
	const auto& p = s->scene.use_bvh ? s->scene.top.primitives : s->scene.top_list.prims;
	for (size_t i = 0; i < p.size(); i++) out[i] = p[i].id;
}

// Closest hit for a batch.  Outputs other than `hits` are optional (null to skip):
// bary 2/ray, inst 1/ray, pos 3/ray, normal 3/ray, uv 2/ray, material 1/ray.
// counters (2 x uint64, optional): total nodes tested, total Triangle::hit calls.
void orc_trace_closest(void* h, const PtxRay* rays, uint64_t n, PtxHit* hits, float* bary, uint32_t* inst, float* pos,
                       float* normal, float* uv, int32_t* material, uint64_t* counters, int nthreads) {
	auto* s = static_cast<OScene*>(h);
	std::vector<Counters> ctr(size_t(std::max(1, nthreads)));
	parallel_for(n, nthreads, [&](uint64_t b, uint64_t e, int t) {
		for (uint64_t i = b; i < e; i++) {
			Ray r;
			r.point = {rays[i].o[0], rays[i].o[1], rays[i].o[2]};
			r.dir = {rays[i].d[0], rays[i].d[1], rays[i].d[2]};
			r.dist_bounds = {rays[i].tmin, rays[i].tmax};
			Trace tr = s->hit(r, counters ? &ctr[size_t(t)] : nullptr);
			hits[i].distance = tr.distance;
			hits[i].prim = tr.hit ? tr.prim : PTX_NO_HIT;
			if (bary) bary[i * 2] = tr.bu, bary[i * 2 + 1] = tr.bv;
			if (inst) inst[i] = tr.hit ? tr.inst : PTX_NO_HIT;
			if (pos) pos[i * 3] = tr.position.x, pos[i * 3 + 1] = tr.position.y, pos[i * 3 + 2] = tr.position.z;
			if (normal) normal[i * 3] = tr.normal.x, normal[i * 3 + 1] = tr.normal.y, normal[i * 3 + 2] = tr.normal.z;
			if (uv) uv[i * 2] = tr.uv.x, uv[i * 2 + 1] = tr.uv.y;
			if (material) material[i] = tr.material;
		}
	});
	if (counters) {
		counters[0] = counters[1] = 0;
		for (auto& c : ctr) counters[0] += c.nodes, counters[1] += c.tris;
	}
}

// The reference's occlusion query is closest-hit's boolean (pathtracer.cpp:659-660).
void orc_trace_any(void* h, const PtxRay* rays, uint64_t n, uint8_t* occluded, int nthreads) {
	auto* s = static_cast<OScene*>(h);
	parallel_for(n, nthreads, [&](uint64_t b, uint64_t e, int) {
		for (uint64_t i = b; i < e; i++) {
			Ray r;
			r.point = {rays[i].o[0], rays[i].o[1], rays[i].o[2]};
			r.dir = {rays[i].d[0], rays[i].d[1], rays[i].d[2]};
			r.dist_bounds = {rays[i].tmin, rays[i].tmax};
			occluded[i] = s->hit(r, nullptr).hit ? 1 : 0;
		}
	});
}

uint32_t orc_make_tiles(uint32_t w, uint32_t hgt, uint32_t samples, uint32_t fixed_seed, PtxTile* out, uint32_t cap) {
	auto tiles = make_tiles(w, hgt, samples, fixed_seed);
	if (out)
		for (size_t i = 0; i < tiles.size() && i < cap; i++) {
			const auto& t = tiles[i];
			out[i] = PtxTile{t.seed, t.x_begin, t.x_end, t.y_begin, t.y_end, t.s_begin, t.s_end};
		}
	return uint32_t(tiles.size());
}

// rng_mode 0 = per-tile mt19937 (reference), 1 = Philox per (pixel, sample) keyed by `seed`.
// stats (4 x uint64, optional): trace() calls, shadow rays, and the nodes tested / Triangle::hit calls of all of them.
void orc_render_tiles(void* h, const PtxCamera* cam, const PtxTile* tiles, uint32_t n_tiles, int rng_mode, uint64_t seed,
                      int64_t* accum_rgb, uint32_t* samples, uint64_t* stats, int nthreads) {
	auto* s = static_cast<OScene*>(h);
	Pathtracer pt;
	pt.scene = &s->scene;
	pt.camera.vertical_fov = cam->vertical_fov, pt.camera.aspect_ratio = cam->aspect_ratio;
	pt.camera.film_w = cam->film_width, pt.camera.film_h = cam->film_height;
	pt.camera.film_samples = cam->film_samples, pt.camera.film_max_ray_depth = cam->film_max_ray_depth;
	pt.camera.camera_to_world = mat_from(cam->camera_to_world);
	const size_t npix = size_t(cam->film_width) * cam->film_height;
	// Tiles that cover the same pixels are serialised by the integer accumulator: integer
	// addition commutes, so any interleaving gives the same sums (pathtracer.h:80-83).
	std::vector<std::thread> th;
	std::atomic<uint32_t> next{0};
	std::vector<PathStats> st(size_t(std::max(1, nthreads)));
	std::vector<Accumulator> accs(size_t(std::max(1, nthreads)));
	for (auto& a : accs) a.reset(cam->film_width, cam->film_height);
	auto work = [&](int t) {
		Pathtracer local = pt;
		local.stats = stats ? &st[size_t(t)] : nullptr;
		for (;;) {
			uint32_t i = next.fetch_add(1);
			if (i >= n_tiles) break;
			Tile tile{tiles[i].seed, tiles[i].x_begin, tiles[i].x_end, tiles[i].y_begin, tiles[i].y_end, tiles[i].s_begin, tiles[i].s_end};
			do_trace_tile(local, tile, rng_mode, seed, accs[size_t(t)]);
		}
	};
	if (nthreads <= 1) work(0);
	else {
		for (int t = 0; t < nthreads; t++) th.emplace_back(work, t);
		for (auto& t : th) t.join();
	}
	for (auto& a : accs) {
		for (size_t i = 0; i < npix * 3; i++) accum_rgb[i] += a.rgb[i];
		for (size_t i = 0; i < npix; i++) samples[i] += a.samples[i];
	}
	if (stats) {
		stats[0] = stats[1] = stats[2] = stats[3] = 0;
		for (auto& x : st) stats[0] += x.rays, stats[1] += x.shadow_rays, stats[2] += x.visits.nodes, stats[3] += x.visits.tris;
	}
}

void orc_accumulator_to_image(const int64_t* accum_rgb, const uint32_t* samples, uint64_t npix, float* rgb) {
	Accumulator a;
	a.rgb.assign(accum_rgb, accum_rgb + npix * 3);
	a.samples.assign(samples, samples + npix);
	accumulator_to_image(a, rgb);
}

void orc_tonemap(const float* rgb, uint64_t npix, float exposure, uint8_t* rgba) { tonemap(rgb, size_t(npix), exposure, rgba); }

// ---- single-function probes for the reference's known-answer tests -------------------------
// Triangle::hit on vertices given as 3 x 8 floats; out: hit, distance, position3, normal3, uv2, u, v
int orc_triangle_hit(const float* verts24, const PtxRay* ray, float* out12) {
	Vert v[3];
	for (int i = 0; i < 3; i++) {
		const float* p = verts24 + i * 8;
		v[i] = Vert{{p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7]}};
	}
	Triangle t{0, 1, 2, v, 0};
	Ray r;
	r.point = {ray->o[0], ray->o[1], ray->o[2]}, r.dir = {ray->d[0], ray->d[1], ray->d[2]}, r.dist_bounds = {ray->tmin, ray->tmax};
	Trace tr = t.hit(r);
	float o[12] = {tr.distance, tr.position.x, tr.position.y, tr.position.z, tr.normal.x, tr.normal.y, tr.normal.z, tr.uv.x, tr.uv.y, tr.bu, tr.bv, 0};
	std::memcpy(out12, o, sizeof(o));
	return tr.hit ? 1 : 0;
}

void orc_triangle_bbox(const float* verts24, float* out6) {
	Vert v[3];
	for (int i = 0; i < 3; i++) v[i].position = {verts24[i * 8], verts24[i * 8 + 1], verts24[i * 8 + 2]};
	Triangle t{0, 1, 2, v, 0};
	BBox b = t.bbox();
	out6[0] = b.min.x, out6[1] = b.min.y, out6[2] = b.min.z, out6[3] = b.max.x, out6[4] = b.max.y, out6[5] = b.max.z;
}

int orc_bbox_hit(const float* box6, const PtxRay* ray, float* times2) {
	BBox b{{box6[0], box6[1], box6[2]}, {box6[3], box6[4], box6[5]}};
	Ray r;
	r.point = {ray->o[0], ray->o[1], ray->o[2]}, r.dir = {ray->d[0], ray->d[1], ray->d[2]};
	V2 t{times2[0], times2[1]};
	bool h = b.hit(r, t);
	times2[0] = t.x, times2[1] = t.y;
	return h ? 1 : 0;
}

int orc_sphere_hit(float radius, const PtxRay* ray, float* out9) {
	Sphere s;
	s.radius = radius;
	Ray r;
	r.point = {ray->o[0], ray->o[1], ray->o[2]}, r.dir = {ray->d[0], ray->d[1], ray->d[2]}, r.dist_bounds = {ray->tmin, ray->tmax};
	Trace tr = s.hit(r);
	float o[9] = {tr.distance, tr.position.x, tr.position.y, tr.position.z, tr.normal.x, tr.normal.y, tr.normal.z, tr.uv.x, tr.uv.y};
	std::memcpy(out9, o, sizeof(o));
	return tr.hit ? 1 : 0;
}

// n scatters of one material with constant textures from one mt19937(seed) stream.
// out per scatter: direction3, attenuation3, specular
void orc_material_scatter(uint32_t type, const float* color_a, const float* color_b, float ior, int fresnel_mode, uint32_t seed,
                          const float* out_dirs3, uint32_t n, float* out7) {
	MaterialTable mt;
	mt.fresnel_mode = fresnel_mode;
	Texture a, b;
	a.color = {color_a[0], color_a[1], color_a[2]};
	b.color = {color_b[0], color_b[1], color_b[2]};
	mt.textures = {a, b};
	Material m;
	m.type = int(type), m.tex_a = 0, m.tex_b = 1, m.ior = ior;
	RNG rng(seed);
	for (uint32_t i = 0; i < n; i++) {
		V3 o{out_dirs3[i * 3], out_dirs3[i * 3 + 1], out_dirs3[i * 3 + 2]};
		Scatter s = mt.scatter(m, rng, o, V2{});
		float* w = out7 + size_t(i) * 7;
		w[0] = s.direction.x, w[1] = s.direction.y, w[2] = s.direction.z;
		w[3] = s.attenuation.x, w[4] = s.attenuation.y, w[5] = s.attenuation.z, w[6] = s.specular ? 1.0f : 0.0f;
	}
}

void orc_mat4_inverse(const float* in16, float* out16) {
	M4 r = mat_from(in16).inverse();
	for (int c = 0; c < 4; c++)
		for (int k = 0; k < 4; k++) out16[c * 4 + k] = r.m[c][k];
}

void orc_rotate_to(const float* dir3, float* out16) {
	M4 r = M4::rotate_to(V3{dir3[0], dir3[1], dir3[2]});
	for (int c = 0; c < 4; c++)
		for (int k = 0; k < 4; k++) out16[c * 4 + k] = r.m[c][k];
}

// mt19937 RNG::unit() stream
void orc_rng_units(uint32_t seed, uint32_t n, float* out) {
	RNG r(seed);
	for (uint32_t i = 0; i < n; i++) out[i] = r.unit();
}

// Philox stream as the GPU pipeline consumes it
void orc_philox_units(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float* out) {
	RNG r = RNG::philox(uint32_t(seed), uint32_t(seed >> 32), pixel, sample);
	for (uint32_t i = 0; i < n; i++) out[i] = r.unit();
}

void orc_env_evaluate(void* h, uint32_t light, const float* dir3, float* out3) {
	auto* s = static_cast<OScene*>(h);
	Spec r = s->scene.env_lights[light].evaluate(s->scene.mats, V3{dir3[0], dir3[1], dir3[2]});
	out3[0] = r.x, out3[1] = r.y, out3[2] = r.z;
}

// Camera::sample_ray + ray.transform for one pixel with given unit draws (a -> x, b -> y)
void orc_camera_ray(const PtxCamera* cam, uint32_t px, uint32_t py, float a, float b, PtxRay* out) {
	Camera c;
	c.vertical_fov = cam->vertical_fov, c.aspect_ratio = cam->aspect_ratio, c.film_w = cam->film_width, c.film_h = cam->film_height;
	c.film_max_ray_depth = cam->film_max_ray_depth;
	c.camera_to_world = mat_from(cam->camera_to_world);
	Ray ray = c.ray_through(px, py, V2{a, b});
	ray.transform(c.camera_to_world);
	out->o[0] = ray.point.x, out->o[1] = ray.point.y, out->o[2] = ray.point.z;
	out->d[0] = ray.dir.x, out->d[1] = ray.dir.y, out->d[2] = ray.dir.z;
	out->tmin = ray.dist_bounds.x, out->tmax = ray.dist_bounds.y;
}

// Light_Instance::incoming (pathtracer/instance.h:74-91) over Delta_Lights::*::incoming (scene/delta_light.cpp:7-40).
// out7 per point: radiance rgb, direction xyz, distance.  T16 null: identity.
void orc_delta_incoming(uint32_t type, const float* color3, float intensity, float inner, float outer, const float* T16, const float* pts3,
                        uint32_t n, float* out7) {
	DeltaLight l;
	l.type = int(type), l.color = {color3[0], color3[1], color3[2]}, l.intensity = intensity, l.inner_angle = inner, l.outer_angle = outer;
	if (T16) l.set(mat_from(T16));
	for (uint32_t i = 0; i < n; i++) {
		DeltaLight::Incoming in = l.incoming(V3{pts3[i * 3], pts3[i * 3 + 1], pts3[i * 3 + 2]});
		float* o = out7 + size_t(i) * 7;
		o[0] = in.radiance.x, o[1] = in.radiance.y, o[2] = in.radiance.z;
		o[3] = in.direction.x, o[4] = in.direction.y, o[5] = in.direction.z, o[6] = in.distance;
	}
}

// Samplers::Rect::sample / pdf (pathtracer/samplers.cpp:7-19) as g++ compiles it: the two draws are arguments of
// one constructor call and g++ evaluates them right to left, so the first draw scales y.  out3 = x, y, pdf.
void orc_sampler_rect(uint32_t seed, float sx, float sy, uint32_t n, float* out3) {
	RNG rng(seed);
	for (uint32_t i = 0; i < n; i++) {
		const float first = rng.unit(), second = rng.unit();
		const float x = second * sx, y = first * sy;
		const bool outside = x < 0.0f || x > sx || y < 0.0f || y > sy;
		out3[i * 3] = x, out3[i * 3 + 1] = y, out3[i * 3 + 2] = outside ? 0.0f : 1.0f / (sx * sy);
	}
}

// n consecutive Camera::sample_ray calls (scene/camera.cpp:7-48) through pixel (px, py) on one mt19937(seed) stream, in
// the reference's (g++) draw order, each followed by ray.transform(camera_to_world) as do_trace does (pathtracer.cpp:453-454).
void orc_camera_sample_rays(const PtxCamera* cam, uint32_t seed, uint32_t px, uint32_t py, uint32_t n, PtxRay* out) {
	Camera c;
	c.vertical_fov = cam->vertical_fov, c.aspect_ratio = cam->aspect_ratio, c.film_w = cam->film_width, c.film_h = cam->film_height;
	c.film_max_ray_depth = cam->film_max_ray_depth;
	c.camera_to_world = mat_from(cam->camera_to_world);
	RNG rng(seed);
	for (uint32_t i = 0; i < n; i++) {
		Ray ray = c.sample_ray(rng, px, py, true).first;
		ray.transform(c.camera_to_world);
		out[i] = PtxRay{{ray.point.x, ray.point.y, ray.point.z}, {ray.dir.x, ray.dir.y, ray.dir.z}, ray.dist_bounds.x, ray.dist_bounds.y};
	}
}

// Samplers::Hemisphere::Cosine::sample / pdf (pathtracer/samplers.cpp:67-83): out4 = direction xyz, pdf.
void orc_sampler_cosine(uint32_t seed, uint32_t n, float* out4) {
	RNG rng(seed);
	for (uint32_t i = 0; i < n; i++) {
		V3 d = cosine_hemisphere_sample(rng);
		out4[i * 4] = d.x, out4[i * 4 + 1] = d.y, out4[i * 4 + 2] = d.z, out4[i * 4 + 3] = cosine_hemisphere_pdf(d);
	}
}

// One Philox-4x32-10 block: counter c[4], key k[2] -> out[4] (known-answer checks).
void orc_philox_block(const uint32_t* c4, const uint32_t* k2, uint32_t* out4) { Philox::block(c4[0], c4[1], c4[2], c4[3], k2[0], k2[1], out4); }

}  // extern "C"

// pt_facade.h -- the host-side C++ surface a caller of the reference's path tracer uses, kept
// under the same names and meanings, implemented over the C ABI in include/ptx.h.
//
//   PT::Trace                  pathtracer/trace.h:10-44
//   PT::Tri_Mesh_Vert          pathtracer/tri_mesh.h:13-17
//   PT::Tri_Mesh               pathtracer/tri_mesh.h:44-74   (ctor from vertices + index triples, hit, bbox, n_triangles)
//   PT::BVH<PT::Triangle>      pathtracer/bvh.h:11-60         (build, hit, bbox, n_primitives; public nodes / primitives / root_idx)
//   PT::Pathtracer             pathtracer/pathtracer.h:18-53  (use_bvh, render, in_progress, completion_time, build_scene, set_camera)
//
// Differences a caller sees: geometry lives on the GPU, so single-ray hit() costs a device round
// trip and the batch forms (hit_batch) are the ones to use in a loop; render() takes the flattened
// scene descriptor (PtxSceneDesc, e.g. from ptx_js3d_load) instead of a ::Scene; errors are C++
// exceptions (PT::Error) carrying ptx_last_error() instead of an abort.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/ptx.h"

namespace PT {

struct Error : std::runtime_error {
	using std::runtime_error::runtime_error;
};
inline void check(int rc) {
	if (rc != PTX_OK) throw Error(ptx_last_error());
}

struct Vec2 {
	float x = 0, y = 0;
};
struct Vec3 {
	float x = 0, y = 0, z = 0;
};
struct BBox {
	Vec3 min{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
	Vec3 max{-std::numeric_limits<float>::max(), -std::numeric_limits<float>::max(), -std::numeric_limits<float>::max()};
};

// lib/ray.h:11-49: the constructor normalises the direction with three divisions by the norm.
struct Ray {
	Vec3 point, dir;
	Vec2 dist_bounds{0.0f, std::numeric_limits<float>::infinity()};
	uint32_t depth = 0;
	Ray() = default;
	Ray(Vec3 p, Vec3 d, Vec2 bounds = Vec2{0.0f, std::numeric_limits<float>::infinity()}, uint32_t depth_ = 0) : point(p), dist_bounds(bounds), depth(depth_) {
		const float n = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
		dir = Vec3{d.x / n, d.y / n, d.z / n};
	}
	PtxRay pod() const { return PtxRay{{point.x, point.y, point.z}, {dir.x, dir.y, dir.z}, dist_bounds.x, dist_bounds.y}; }
};

struct Trace {
	bool hit = false;
	float distance = 0.0f;
	Vec3 position, normal, origin;
	Vec2 uv;
	uint32_t primitive = PTX_NO_HIT;   // index into the mesh's primitive array (the reference carries a material pointer here)
};

struct Tri_Mesh_Vert {
	Vec3 position, normal;
	Vec2 uv;
};
static_assert(sizeof(Tri_Mesh_Vert) == 32, "8 floats per vertex, as ptx.h expects");

struct Triangle {
	uint32_t v0 = 0, v1 = 0, v2 = 0;
};

template<typename Primitive> class BVH;

// BVH<Triangle>: the host builder with the reference's public members.
template<> class BVH<Triangle> {
public:
	struct Node {
		BBox bbox;
		size_t start = 0, size = 0, l = 0, r = 0;
		bool is_leaf() const { return l == r; }
	};
	BVH() = default;
	// build(std::vector<Primitive>&&, max_leaf_size) (bvh.cpp:31-60); vertices stay owned by the caller
	void build(const std::vector<Tri_Mesh_Vert>& verts, std::vector<Triangle>&& prims, size_t max_leaf_size = 4) {
		ptx_bvh_t b = nullptr;
		static_assert(sizeof(Triangle) == 12, "index triples");
		check(ptx_bvh_build_triangles(reinterpret_cast<const float*>(verts.data()), uint32_t(verts.size()), reinterpret_cast<const uint32_t*>(prims.data()),
		                              uint32_t(prims.size()), uint32_t(max_leaf_size), &b));
		std::vector<PtxBvhNode> raw(ptx_bvh_num_nodes(b));
		std::vector<uint32_t> order(ptx_bvh_num_prims(b));
		int rc = raw.empty() ? PTX_OK : ptx_bvh_get_nodes(b, raw.data());
		if (rc == PTX_OK && !order.empty()) rc = ptx_bvh_get_order(b, order.data());
		ptx_bvh_destroy(b);
		check(rc);
		nodes.clear();
		for (const PtxBvhNode& n : raw) {
			Node out;
			out.bbox.min = Vec3{n.bmin[0], n.bmin[1], n.bmin[2]}, out.bbox.max = Vec3{n.bmax[0], n.bmax[1], n.bmax[2]};
			out.start = n.start, out.size = n.size, out.l = n.l, out.r = n.r;
			nodes.push_back(out);
		}
		primitives.resize(order.size());
		for (size_t i = 0; i < order.size(); i++) primitives[i] = prims[order[i]];
		root_idx = 0;
	}
	BBox bbox() const { return nodes.empty() ? BBox{} : nodes[root_idx].bbox; }
	size_t n_primitives() const { return primitives.size(); }
	std::vector<PtxBvhNode> pod_nodes() const {
		std::vector<PtxBvhNode> out;
		for (const Node& n : nodes)
			out.push_back(PtxBvhNode{{n.bbox.min.x, n.bbox.min.y, n.bbox.min.z}, {n.bbox.max.x, n.bbox.max.y, n.bbox.max.z}, uint32_t(n.start), uint32_t(n.size), uint32_t(n.l), uint32_t(n.r)});
		return out;
	}
	std::vector<Triangle> primitives;   // in tree order
	std::vector<Node> nodes;
	size_t root_idx = 0;
};

// Tri_Mesh(const Indexed_Mesh&, bool use_bvh): geometry resident on one GPU.
class Tri_Mesh {
public:
	Tri_Mesh() = default;
	Tri_Mesh(const std::vector<Tri_Mesh_Vert>& vertices, const std::vector<uint32_t>& indices, bool use_bvh_ = true, int device = 0) : verts(vertices), use_bvh(use_bvh_) {
		std::vector<Triangle> tris(indices.size() / 3);
		for (size_t i = 0; i < tris.size(); i++) tris[i] = Triangle{indices[3 * i], indices[3 * i + 1], indices[3 * i + 2]};
		PtxMeshDesc d{};
		d.verts = reinterpret_cast<const float*>(verts.data()), d.n_verts = uint32_t(verts.size());
		d.use_bvh = use_bvh ? 1 : 0, d.max_leaf_size = 4;
		std::vector<PtxBvhNode> pod;
		if (use_bvh) {
			triangle_bvh.build(verts, std::move(tris), 4);   // tri_mesh.cpp:173
			pod = triangle_bvh.pod_nodes();
			d.indices = reinterpret_cast<const uint32_t*>(triangle_bvh.primitives.data()), d.n_tris = uint32_t(triangle_bvh.primitives.size());
			d.nodes = pod.data(), d.n_nodes = uint32_t(pod.size());
		} else {
			triangle_bvh.primitives = std::move(tris);
			d.indices = reinterpret_cast<const uint32_t*>(triangle_bvh.primitives.data()), d.n_tris = uint32_t(triangle_bvh.primitives.size());
		}
		check(ptx_scene_create_mesh(device, &d, &scene));
	}
	Tri_Mesh(Tri_Mesh&& o) noexcept { *this = std::move(o); }
	Tri_Mesh& operator=(Tri_Mesh&& o) noexcept {
		if (this != &o) {
			release();
			verts = std::move(o.verts), triangle_bvh = std::move(o.triangle_bvh), use_bvh = o.use_bvh, scene = o.scene, o.scene = nullptr;
		}
		return *this;
	}
	Tri_Mesh(const Tri_Mesh&) = delete;
	Tri_Mesh& operator=(const Tri_Mesh&) = delete;
	~Tri_Mesh() { release(); }

	BBox bbox() const {
		if (use_bvh) return triangle_bvh.bbox();
		BBox b;
		for (const Triangle& t : triangle_bvh.primitives)
			for (uint32_t v : {t.v0, t.v1, t.v2}) {
				const Vec3 p = verts[v].position;
				b.min = Vec3{std::min(b.min.x, p.x), std::min(b.min.y, p.y), std::min(b.min.z, p.z)};
				b.max = Vec3{std::max(b.max.x, p.x), std::max(b.max.y, p.y), std::max(b.max.z, p.z)};
			}
		return b;
	}
	size_t n_triangles() const { return triangle_bvh.primitives.size(); }

	// Trace hit(const Ray&) for a whole batch: exact reference traversal order unless `culled`.
	std::vector<Trace> hit_batch(const std::vector<Ray>& rays, bool culled = true) const {
		std::vector<PtxRay> pod(rays.size());
		for (size_t i = 0; i < rays.size(); i++) pod[i] = rays[i].pod();
		std::vector<PtxHit> hits(rays.size());
		std::vector<float> bary(rays.size() * 2);
		check(ptx_trace_closest(scene, pod.data(), pod.size(), hits.data(), bary.data(), nullptr, culled ? PTX_TRAVERSE_CULLED : PTX_TRAVERSE_REFERENCE));
		std::vector<Trace> out(rays.size());
		for (size_t i = 0; i < rays.size(); i++) {
			Trace& t = out[i];
			t.origin = rays[i].point, t.distance = hits[i].distance;
			if (hits[i].prim == PTX_NO_HIT) continue;
			t.hit = true, t.primitive = hits[i].prim;
			const Triangle& tri = triangle_bvh.primitives[hits[i].prim];
			const Tri_Mesh_Vert &a = verts[tri.v0], &b = verts[tri.v1], &c = verts[tri.v2];
			const float u = bary[2 * i], v = bary[2 * i + 1], w = 1 - u - v;   // tri_mesh.cpp:97-101
			// position = ray.point + dir * t (tri_mesh.cpp:91-97); the ABI returns distance = |dir * t|, which for
			// the unit directions Ray's constructor makes equals t to rounding
			const Vec3 o = rays[i].point, dd = rays[i].dir;
			t.position = Vec3{o.x + dd.x * t.distance, o.y + dd.y * t.distance, o.z + dd.z * t.distance};
			t.normal = Vec3{w * a.normal.x + u * b.normal.x + v * c.normal.x, w * a.normal.y + u * b.normal.y + v * c.normal.y, w * a.normal.z + u * b.normal.z + v * c.normal.z};
			t.uv = Vec2{w * a.uv.x + u * b.uv.x + v * c.uv.x, w * a.uv.y + u * b.uv.y + v * c.uv.y};
		}
		return out;
	}
	Trace hit(const Ray& ray) const { return hit_batch(std::vector<Ray>{ray}, false)[0]; }

private:
	void release() {
		if (scene) ptx_scene_destroy(scene);
		scene = nullptr;
	}
	std::vector<Tri_Mesh_Vert> verts;
	BVH<Triangle> triangle_bvh;
	bool use_bvh = true;
	ptx_scene_t scene = nullptr;
};

// Image handed to the render callback: linear RGB floats, row 0 at the bottom (util/hdr_image.h:13).
struct HDR_Image {
	uint32_t w = 0, h = 0;
	std::vector<float> rgb;
	void tonemap_to(std::vector<uint8_t>& out, float exposure) const {   // util/hdr_image.cpp:201-225
		out.resize(size_t(w) * h * 4);
		check(ptx_tonemap(rgb.data(), uint64_t(w) * h, exposure, out.data()));
	}
};

class Pathtracer {
public:
	using Render_Report = std::pair<float, HDR_Image>;
	Pathtracer() = default;
	~Pathtracer() {
		cancel();
		if (scene) ptx_scene_destroy(scene);
	}
	Pathtracer(const Pathtracer&) = delete;
	Pathtracer& operator=(const Pathtracer&) = delete;

	void use_bvh(bool use) { scene_use_bvh = use; }   // pathtracer.h:37
	void set_device(int d) { device = d; }
	void set_seed(uint32_t fixed_seed) { seed = fixed_seed; }   // RNG::fixed_seed (util/rand.cpp), main.cpp:66
	void set_tiles_per_batch(uint32_t n) { tiles_per_batch = n ? n : 1; }
	void set_report_interval(float seconds) { report_interval = seconds; }   // 0: a report (and a film read-back) after every batch

	// build_scene(Scene&) (pathtracer.cpp:182-387) from the flattened descriptor
	void build_scene(const PtxSceneDesc& in) {
		const auto t0 = std::chrono::steady_clock::now();
		if (scene) ptx_scene_destroy(scene);
		scene = nullptr;
		PtxSceneDesc d = in;
		std::vector<PtxMeshDesc> meshes(d.meshes, d.meshes + d.n_meshes);
		for (auto& m : meshes) m.use_bvh = scene_use_bvh ? 1 : 0;
		d.meshes = meshes.data(), d.use_bvh = scene_use_bvh ? 1 : 0;
		check(ptx_scene_create(device, &d, &scene));
		build_seconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
	}
	void set_camera(const PtxCamera& c) { camera = c; }

	// render(scene, camera, callback, quit, add_samples) (pathtracer.cpp:491-584): asynchronous; the
	// callback receives (fraction done, image so far) from the render thread after every batch of tiles.
	void render(const PtxSceneDesc& scene_desc, const PtxCamera& cam, std::function<void(Render_Report&&)>&& f, bool* quit, bool add_samples = false) {
		cancel();
		build_scene(scene_desc);
		set_camera(cam);
		const size_t npix = size_t(cam.film_width) * cam.film_height;
		if (!add_samples || accumulator.size() != npix * 3) {
			accumulator.assign(npix * 3, 0);
			accumulator_samples.assign(npix, 0);
		}
		tiles.resize(ptx_make_tiles(cam.film_width, cam.film_height, cam.film_samples, seed, nullptr, 0));
		ptx_make_tiles(cam.film_width, cam.film_height, cam.film_samples, seed, tiles.data(), uint32_t(tiles.size()));
		total_tiles = uint32_t(tiles.size()), traced_tiles = 0;
		cancel_flag = quit, report_fn = std::move(f), failed.clear();
		running = true;
		render_start = std::chrono::steady_clock::now();
		worker = std::thread([this] {
			// The film stays on the GPU for the whole render; it comes back for a progress report (at most
			// one per report interval) and when the render ends or is cancelled.
			ptx_frame_t frame = nullptr;
			bool ok = ptx_frame_create(scene, camera.film_width, camera.film_height, &frame) == PTX_OK &&
			          ptx_frame_upload(frame, accumulator.data(), accumulator_samples.data()) == PTX_OK;
			auto last_report = std::chrono::steady_clock::now();
			for (uint32_t begin = 0; ok && begin < total_tiles; begin += tiles_per_batch) {
				if ((cancel_flag && *cancel_flag) || stop) break;
				const uint32_t n = std::min(tiles_per_batch, total_tiles - begin);
				ok = ptx_render_tiles(scene, &camera, tiles.data() + begin, n, seed, ptx_frame_accum(frame), ptx_frame_samples(frame), PTX_MEM_DEVICE) == PTX_OK;
				if (!ok) break;
				traced_tiles += n;
				const auto now = std::chrono::steady_clock::now();
				const bool last = begin + n >= total_tiles;
				if (report_fn && (last || std::chrono::duration<float>(now - last_report).count() >= report_interval)) {
					ok = ptx_frame_download(frame, accumulator.data(), accumulator_samples.data()) == PTX_OK;
					if (!ok) break;
					report_fn(Render_Report{float(traced_tiles) / float(total_tiles), accumulator_to_image()});
					last_report = now;
				}
			}
			if (ok && frame) ok = ptx_frame_download(frame, accumulator.data(), accumulator_samples.data()) == PTX_OK;
			if (!ok) failed = ptx_last_error();
			ptx_frame_destroy(frame);
			render_seconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - render_start).count();
			running = false;
		});
	}
	bool in_progress() const { return running; }   // pathtracer.cpp:479-481
	std::pair<float, float> completion_time() const { return {build_seconds, render_seconds}; }   // pathtracer.cpp:483-485
	const std::string& error() const { return failed; }
	void wait() {
		if (worker.joinable()) worker.join();
	}
	void cancel() {   // pathtracer.cpp:587-594
		stop = true;
		wait();
		stop = false;
	}
	uint32_t samples_per_pixel_min() const { return accumulator_samples.empty() ? 0u : *std::min_element(accumulator_samples.begin(), accumulator_samples.end()); }
	uint32_t samples_per_pixel_max() const { return accumulator_samples.empty() ? 0u : *std::max_element(accumulator_samples.begin(), accumulator_samples.end()); }
	HDR_Image accumulator_to_image() const {   // pathtracer.cpp:429-442
		HDR_Image img;
		img.w = camera.film_width, img.h = camera.film_height;
		img.rgb.resize(accumulator_samples.size() * 3);
		check(ptx_accumulator_to_image(accumulator.data(), accumulator_samples.data(), accumulator_samples.size(), img.rgb.data()));
		return img;
	}

private:
	bool scene_use_bvh = true;
	int device = 0;
	uint32_t seed = 1, tiles_per_batch = 128;   // 128 tiles of 100x100x50 = 64 M paths, one pass of the library
	ptx_scene_t scene = nullptr;
	PtxCamera camera{};
	std::vector<PtxTile> tiles;
	std::vector<int64_t> accumulator;          // 40.24 fixed point, 3 per pixel (pathtracer.h:80-83)
	std::vector<uint32_t> accumulator_samples;
	uint32_t total_tiles = 0;
	std::atomic<uint32_t> traced_tiles{0};
	std::atomic<bool> running{false}, stop{false};
	bool* cancel_flag = nullptr;
	std::function<void(Render_Report&&)> report_fn;
	std::thread worker;
	std::chrono::steady_clock::time_point render_start;
	float build_seconds = 0.0f, render_seconds = 0.0f, report_interval = 0.25f;
	std::string failed;
};

}  // namespace PT

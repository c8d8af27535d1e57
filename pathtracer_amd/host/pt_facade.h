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
#include <mutex>
#include <random>
#include <shared_mutex>
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
			// position = ray.point + dir * t (tri_mesh.cpp:91-97).  The ABI returns distance = |dir * t|, not t: t is
			// recomputed here for the winning triangle with Triangle::hit's own expression (tri_mesh.cpp:58-85).
			const Vec3 o = rays[i].point, dd = rays[i].dir;
			const float tt = triangle_t(a.position, b.position, c.position, o, dd);
			t.position = Vec3{o.x + dd.x * tt, o.y + dd.y * tt, o.z + dd.z * tt};
			t.normal = Vec3{w * a.normal.x + u * b.normal.x + v * c.normal.x, w * a.normal.y + u * b.normal.y + v * c.normal.y, w * a.normal.z + u * b.normal.z + v * c.normal.z};
			t.uv = Vec2{w * a.uv.x + u * b.uv.x + v * c.uv.x, w * a.uv.y + u * b.uv.y + v * c.uv.y};
		}
		return out;
	}
	Trace hit(const Ray& ray) const { return hit_batch(std::vector<Ray>{ray}, false)[0]; }

private:
	static Vec3 sub(Vec3 l, Vec3 r) { return Vec3{l.x - r.x, l.y - r.y, l.z - r.z}; }
	static Vec3 cross(Vec3 l, Vec3 r) { return Vec3{l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x}; }
	static float dot(Vec3 l, Vec3 r) { return l.x * r.x + l.y * r.y + l.z * r.z; }
	// the `t` of Triangle::hit: third component of Vec3(-dot(cross(s,e2),dir), dot(cross(e1,dir),s), -dot(cross(s,e2),e1)) / denom
	static float triangle_t(Vec3 v0, Vec3 v1, Vec3 v2, Vec3 o, Vec3 dir) {
		const Vec3 s = sub(o, v0), e1 = sub(v1, v0), e2 = sub(v2, v0);
		const float denom = dot(cross(e1, dir), e2);
		return -dot(cross(s, e2), e1) / denom;
	}
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
		destroy_scenes();
	}
	Pathtracer(const Pathtracer&) = delete;
	Pathtracer& operator=(const Pathtracer&) = delete;

	void use_bvh(bool use) { scene_use_bvh = use; }   // pathtracer.h:37
	// The GPUs the render runs on.  With more than one entry every GPU holds a copy of the scene, takes every
	// n-th tile of the reference's centre-out tile order into a film of its own on a host thread of its own, and
	// the films are added at the end (ptx_frame_reduce): what the reference's thread pool does with its tiles and
	// its one mutex-guarded accumulator (pathtracer.cpp:566-584).  The same GPU may be listed twice.
	void set_devices(const std::vector<int>& list) {
		cancel();
		destroy_scenes();
		devices = list.empty() ? std::vector<int>{0} : list;
	}
	void set_device(int d) { set_devices({d}); }
	// How the GPUs' films are added: Peer = ptx_frame_reduce (the first GPU reads the others' films over their links),
	// Rccl = ptx_frame_reduce_rccl (ncclReduce over a communicator with one rank per GPU).  Same frame bit for bit.
	// Rccl needs every GPU listed once; a list that names a GPU twice uses Peer.
	enum class Reduce { Peer, Rccl };
	void set_reduce(Reduce r) { reduce_mode = r; }
	// RNG::fixed_seed (util/rand.cpp, main.cpp:66).  0 = none: every render() call draws a seed of its own, as the
	// reference's unseeded RNG does (pathtracer.cpp:533-534).
	void set_seed(uint32_t fixed_seed) { seed = fixed_seed; }
	void set_tiles_per_batch(uint32_t n) { tiles_per_batch = n ? n : 1; }
	void set_report_interval(float seconds) { report_interval = seconds; }   // 0: a report (and a film read-back) after every batch

	// build_scene(Scene&) (pathtracer.cpp:182-387) from the flattened descriptor, once per GPU
	void build_scene(const PtxSceneDesc& in) {
		const auto t0 = std::chrono::steady_clock::now();
		destroy_scenes();
		PtxSceneDesc d = in;
		std::vector<PtxMeshDesc> meshes(d.meshes, d.meshes + d.n_meshes);
		for (auto& m : meshes) m.use_bvh = scene_use_bvh ? 1 : 0;
		d.meshes = meshes.data(), d.use_bvh = scene_use_bvh ? 1 : 0;
		scenes.assign(devices.size(), nullptr);
		std::vector<std::string> errors(devices.size());
		std::vector<std::thread> th;
		for (size_t k = 0; k < devices.size(); k++)
			th.emplace_back([&, k] {
				if (ptx_scene_create(devices[k], &d, &scenes[k]) != PTX_OK) errors[k] = ptx_last_error();   // the error text is per thread
			});
		for (auto& t : th) t.join();
		for (const std::string& e : errors)
			if (!e.empty()) {
				destroy_scenes();
				throw Error(e);
			}
		build_seconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
	}
	void set_camera(const PtxCamera& c) { camera = c; }

	// render(scene, camera, callback, quit, add_samples) (pathtracer.cpp:491-584): asynchronous; the callback
	// receives (fraction done, image so far) from a render thread after batches of tiles.
	// add_samples keeps the film and the scene built by the previous call when the film size is unchanged
	// (pathtracer.cpp:504-519) and adds film_samples NEW samples per pixel: the random streams are named by
	// (seed, pixel, sample index), so the pass continues the sample index where the film's count stands.  (With a
	// fixed seed the reference re-seeds its tile-seed generator and adds the same samples a second time; with no
	// fixed seed it draws fresh ones.  Continuing the index gives fresh samples in both cases.)
	void render(const PtxSceneDesc& scene_desc, const PtxCamera& cam, std::function<void(Render_Report&&)>&& f, bool* quit, bool add_samples = false) {
		cancel();
		const size_t npix = size_t(cam.film_width) * cam.film_height;
		if (accumulator_samples.size() != npix || accumulator_w != cam.film_width || scenes.empty()) add_samples = false;   // pathtracer.cpp:504-506
		set_camera(cam);
		if (!add_samples) {
			build_scene(scene_desc);
			accumulator_w = cam.film_width;
			accumulator.assign(npix * 3, 0);
			accumulator_samples.assign(npix, 0);
			render_seed = seed ? seed : std::random_device{}();
		}
		const uint32_t s_done = samples_per_pixel_max();   // the pass continues the sample index from here
		tiles.resize(ptx_make_tiles(cam.film_width, cam.film_height, cam.film_samples, render_seed, nullptr, 0));
		ptx_make_tiles(cam.film_width, cam.film_height, cam.film_samples, render_seed, tiles.data(), uint32_t(tiles.size()));
		for (PtxTile& t : tiles) t.s_begin += s_done, t.s_end += s_done;
		total_tiles = uint32_t(tiles.size()), traced_tiles = 0;
		cancel_flag = quit, report_fn = std::move(f), failed.clear();
		running = true;
		render_start = std::chrono::steady_clock::now();
		worker = std::thread([this] { run(); });
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
	const std::vector<int64_t>& accumulator_rgb() const { return accumulator; }   // 40.24 fixed point, 3 per pixel
	HDR_Image accumulator_to_image() const {   // pathtracer.cpp:429-442
		HDR_Image img;
		img.w = camera.film_width, img.h = camera.film_height;
		img.rgb.resize(accumulator_samples.size() * 3);
		check(ptx_accumulator_to_image(accumulator.data(), accumulator_samples.data(), accumulator_samples.size(), img.rgb.data()));
		return img;
	}

private:
	void destroy_scenes() {
		for (ptx_scene_t s : scenes)
			if (s) ptx_scene_destroy(s);
		scenes.clear();
	}

	// The render thread.  Film layout: `total` on the first GPU holds what the film held before this pass (uploaded
	// once); every GPU, the first included, adds its tiles into a zeroed film of its own (`parts`); a report or the
	// end of the render computes total + sum of parts into `shown` and reads that back.  The films stay on the GPUs
	// in between.  Workers render under a shared lock, the reduce takes it exclusively: no part has work in flight
	// while it is read.
	void run() {
		const size_t n_dev = scenes.size();
		std::vector<ptx_frame_t> parts(n_dev, nullptr);
		ptx_frame_t total = nullptr, shown = nullptr;
		std::string first_error;
		std::mutex error_mutex;
		auto note_error = [&] {
			std::lock_guard<std::mutex> g(error_mutex);
			if (first_error.empty()) first_error = ptx_last_error();
		};
		bool ok = ptx_frame_create(scenes[0], camera.film_width, camera.film_height, &total) == PTX_OK &&
		          ptx_frame_create(scenes[0], camera.film_width, camera.film_height, &shown) == PTX_OK &&
		          ptx_frame_upload(total, accumulator.data(), accumulator_samples.data()) == PTX_OK;
		for (size_t k = 0; ok && k < n_dev; k++) ok = ptx_frame_create(scenes[k], camera.film_width, camera.film_height, &parts[k]) == PTX_OK;
		if (!ok) note_error();
		std::shared_mutex films;
		ptx_comm_t comm = nullptr;
		if (ok && reduce_mode == Reduce::Rccl) {   // also with one GPU when asked for: a one-rank communicator
			std::vector<int> sorted = devices;
			std::sort(sorted.begin(), sorted.end());
			if (std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end() && ptx_comm_create(devices.data(), uint32_t(n_dev), &comm) != PTX_OK) ok = false, note_error();
		}
		auto read_back = [&]() -> bool {   // caller holds `films` exclusively
			if (comm) {   // shown = sum of parts (RCCL), then + total where it lies
				const ptx_frame_t base[1] = {total};
				return ptx_frame_reduce_rccl(comm, shown, parts.data(), uint32_t(n_dev)) == PTX_OK && ptx_frame_reduce(shown, base, 1) == PTX_OK &&
				       ptx_frame_download(shown, accumulator.data(), accumulator_samples.data()) == PTX_OK;
			}
			std::vector<ptx_frame_t> all{total};
			all.insert(all.end(), parts.begin(), parts.end());
			return ptx_frame_clear(shown) == PTX_OK && ptx_frame_reduce(shown, all.data(), uint32_t(all.size())) == PTX_OK &&
			       ptx_frame_download(shown, accumulator.data(), accumulator_samples.data()) == PTX_OK;
		};
		std::atomic<bool> abort_all{!ok};
		std::atomic<bool> final_reported{false};
		auto last_report = std::chrono::steady_clock::now();
		std::mutex report_mutex;
		auto device_loop = [&](size_t k) {
			// tiles k, k + n_dev, ... of the centre-out order (neighbouring tiles cost about the same), in batches
			std::vector<PtxTile> mine;
			for (size_t i = k; i < tiles.size(); i += n_dev) mine.push_back(tiles[i]);
			for (size_t begin = 0; begin < mine.size() && !abort_all; begin += tiles_per_batch) {
				if ((cancel_flag && *cancel_flag) || stop) break;
				const uint32_t n = uint32_t(std::min<size_t>(tiles_per_batch, mine.size() - begin));
				{
					std::shared_lock<std::shared_mutex> lock(films);
					if (ptx_render_tiles(scenes[k], &camera, mine.data() + begin, n, render_seed, ptx_frame_accum(parts[k]), ptx_frame_samples(parts[k]), PTX_MEM_DEVICE) != PTX_OK) {
						note_error();
						abort_all = true;
						break;
					}
				}
				const uint32_t done = traced_tiles.fetch_add(n) + n;
				if (!report_fn) continue;
				std::unique_lock<std::mutex> rl(report_mutex, std::try_to_lock);   // one reporter at a time; the others go on rendering
				if (!rl.owns_lock()) continue;
				const auto now = std::chrono::steady_clock::now();
				if (done >= total_tiles) continue;   // the report at 1.0 comes from run(), after every worker has finished
				if (std::chrono::duration<float>(now - last_report).count() >= report_interval) {
					std::unique_lock<std::shared_mutex> lock(films);
					if (!read_back()) {
						note_error();
						abort_all = true;
						break;
					}
					lock.unlock();
					const uint32_t so_far = traced_tiles.load();
					if (so_far < total_tiles) report_fn(Render_Report{float(so_far) / float(total_tiles), accumulator_to_image()});   // 1.0 is run()'s to report
					last_report = now;
				}
			}
		};
		if (ok) {
			std::vector<std::thread> th;
			for (size_t k = 1; k < n_dev; k++) th.emplace_back(device_loop, k);
			device_loop(0);
			for (auto& t : th) t.join();
			if (!abort_all) {
				std::unique_lock<std::shared_mutex> lock(films);
				if (!read_back()) {
					note_error();
				} else if (report_fn && traced_tiles.load() >= total_tiles && !final_reported.exchange(true)) {
					// the reference ends every finished render with report_fn({1.0f, image}) (pathtracer.cpp:574-579)
					lock.unlock();
					report_fn(Render_Report{1.0f, accumulator_to_image()});
				}
			}
		}
		if (comm) ptx_comm_destroy(comm);
		failed = first_error;
		for (ptx_frame_t p : parts) ptx_frame_destroy(p);
		ptx_frame_destroy(total), ptx_frame_destroy(shown);
		render_seconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - render_start).count();
		running = false;
	}

	bool scene_use_bvh = true;
	Reduce reduce_mode = Reduce::Peer;
	std::vector<int> devices{0};
	uint32_t seed = 1, render_seed = 1, tiles_per_batch = 128;   // 128 tiles of 100x100x50 = 64 M paths, one pass of the library
	std::vector<ptx_scene_t> scenes;   // one per entry of `devices`
	PtxCamera camera{};
	std::vector<PtxTile> tiles;
	std::vector<int64_t> accumulator;          // 40.24 fixed point, 3 per pixel (pathtracer.h:80-83)
	std::vector<uint32_t> accumulator_samples;
	uint32_t accumulator_w = 0;
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

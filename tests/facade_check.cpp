// Exercises the C++ facade (pathtracer_amd/host/pt_facade.h) the way the reference's own tests use
// PT::Tri_Mesh and PT::BVH (tests/a3/test.a3.task3.bvh.hit.cpp, test.a3.task3.bvh.build.cpp).
// Host-only part runs anywhere; pass "gpu" to also trace rays.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../pathtracer_amd/host/pt_facade.h"

#define EXPECT(c)                                                         \
	do {                                                                  \
		if (!(c)) {                                                       \
			std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);    \
			return 1;                                                     \
		}                                                                 \
	} while (0)

int main(int argc, char** argv) {
	const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
	// four stacked triangles, max_leaf_size 2 (test.a3.task3.bvh.build.cpp:118-128)
	std::vector<PT::Tri_Mesh_Vert> verts;
	std::vector<PT::Triangle> tris;
	for (float z : {0.0f, 1.0f, 3.0f, 4.0f}) {
		const uint32_t b = uint32_t(verts.size());
		verts.push_back({{0, 0, z}, {0, 1, 0}, {}}), verts.push_back({{1, 0, z}, {0, 1, 0}, {}}), verts.push_back({{0, 1, z}, {0, 1, 0}, {}});
		tris.push_back({b, b + 1, b + 2});
	}
	PT::BVH<PT::Triangle> bvh;
	auto copy = tris;
	bvh.build(verts, std::move(copy), 2);
	EXPECT(bvh.n_primitives() == 4);
	const auto& root = bvh.nodes[bvh.root_idx];
	EXPECT(root.size == 4 && !root.is_leaf());
	EXPECT(bvh.nodes[root.l].size + bvh.nodes[root.r].size == 4);
	EXPECT(bvh.nodes[root.l].start + bvh.nodes[root.l].size == bvh.nodes[root.r].start);
	EXPECT(std::fabs(bvh.bbox().min.z) < 1e-3f && std::fabs(bvh.bbox().max.z - 4.0f) < 1e-3f);   // flat triangles are padded by 1e-4 (tri_mesh.cpp:28-39)
	for (const auto& n : bvh.nodes)
		if (n.is_leaf()) EXPECT(n.size <= 2);
	PT::Ray ray({0.25f, 0.25f, -1.0f}, {0, 0, 2});
	EXPECT(std::fabs(ray.dir.z - 1.0f) < 1e-7f);
	if (!gpu) {
		std::printf("facade host checks passed\n");
		return 0;
	}
	std::vector<uint32_t> idx;
	for (const auto& t : tris) idx.push_back(t.v0), idx.push_back(t.v1), idx.push_back(t.v2);
	for (bool use_bvh : {true, false}) {
		PT::Tri_Mesh mesh(verts, idx, use_bvh);
		EXPECT(mesh.n_triangles() == 4);
		PT::Trace t = mesh.hit(ray);
		EXPECT(t.hit && std::fabs(t.distance - 1.0f) < 1e-6f);
		EXPECT(std::fabs(t.position.z) < 1e-6f && std::fabs(t.normal.y - 1.0f) < 1e-6f);
		PT::Trace miss = mesh.hit(PT::Ray({5, 5, -1}, {0, 0, 1}));
		EXPECT(!miss.hit);
		std::vector<PT::Ray> batch(1000, PT::Ray({0.25f, 0.25f, 5.0f}, {0, 0, -1}));
		for (const PT::Trace& b : mesh.hit_batch(batch)) EXPECT(b.hit && std::fabs(b.distance - 1.0f) < 1e-6f);
	}
	// PT::Pathtracer::render (pathtracer.cpp:491-584): an emissive square above a diffuse floor, camera looking
	// down at the floor.  Asynchronous render with progress reports, then more samples on top, then a cancelled one.
	{
		const float quad[4 * 8] = {-1, 0, -1, 0, 1, 0, 0, 0, -1, 0, 1, 0, 1, 0, 0, 1, 1, 0, -1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 1};
		const uint32_t quad_idx[6] = {0, 1, 2, 2, 1, 3};
		PtxMeshDesc mesh{quad, 4, quad_idx, 2, 1, 0, nullptr, 0};
		PtxTextureDesc tex[2] = {{PTX_TEX_CONSTANT, {0.8f, 0.8f, 0.8f}, 1.0f, 0, 0, 0, nullptr}, {PTX_TEX_CONSTANT, {1, 1, 1}, 5.0f, 0, 0, 0, nullptr}};
		PtxMaterialDesc mat[2] = {{PTX_MAT_LAMBERTIAN, 0, -1, 1.0f}, {PTX_MAT_EMISSIVE, 1, -1, 1.0f}};
		PtxInstanceDesc inst[2] = {};
		const float floor_T[16] = {4, 0, 0, 0, 0, 1, 0, 0, 0, 0, 4, 0, 0, 0, 0, 1};                 // scale 4 in x and z (column major)
		const float lamp_T[16] = {1, 0, 0, 0, 0, -1, 0, 0, 0, 0, -1, 0, 0, 3, 0, 1};               // turned over, 3 above the floor
		inst[0].mesh = 0, inst[0].material = 0, std::memcpy(inst[0].T, floor_T, sizeof(floor_T));
		inst[1].mesh = 0, inst[1].material = 1, std::memcpy(inst[1].T, lamp_T, sizeof(lamp_T));
		PtxSceneDesc desc{};
		desc.meshes = &mesh, desc.n_meshes = 1, desc.instances = inst, desc.n_instances = 2, desc.textures = tex, desc.n_textures = 2;
		desc.materials = mat, desc.n_materials = 2, desc.use_bvh = 1;
		PtxCamera cam{};
		cam.vertical_fov = 60.0f, cam.aspect_ratio = 1.0f, cam.film_width = 250, cam.film_height = 250, cam.film_samples = 60, cam.film_max_ray_depth = 3;
		const float cam_T[16] = {1, 0, 0, 0, 0, 0, -1, 0, 0, 1, 0, 0, 0, 2, 0, 1};                  // at height 2, looking down -y
		std::memcpy(cam.camera_to_world, cam_T, sizeof(cam_T));

		PT::Pathtracer pt;
		pt.set_tiles_per_batch(4);      // 3x3 pixel tiles x 2 sample chunks = 18 tiles: five batches
		pt.set_report_interval(0.0f);
		std::vector<float> progress;
		double last_mean = 0.0;
		bool quit = false;
		pt.render(desc, cam, [&](PT::Pathtracer::Render_Report&& r) {
			progress.push_back(r.first);
			double m = 0.0;
			for (float v : r.second.rgb) m += v;
			last_mean = m / double(r.second.rgb.size());
		}, &quit);
		pt.wait();
		EXPECT(pt.error().empty() && !pt.in_progress());
		EXPECT(progress.size() == 5 && progress.back() == 1.0f);
		for (size_t k = 1; k < progress.size(); k++) EXPECT(progress[k] > progress[k - 1]);
		EXPECT(last_mean > 0.05 && last_mean < 5.0);
		const double first_mean = last_mean;
		PT::HDR_Image img = pt.accumulator_to_image();
		EXPECT(img.w == 250 && img.h == 250 && img.rgb.size() == 250u * 250u * 3u);

		const std::vector<int64_t> first_acc = pt.accumulator_rgb();

		// the same render with its tiles dealt to two workers (two scene copies and two films on this GPU, added at the
		// end -- the path several GPUs take): the accumulator is the single-worker one bit for bit
		{
			PT::Pathtracer two;
			two.set_devices({0, 0});
			two.set_tiles_per_batch(4);
			two.render(desc, cam, nullptr, &quit);
			two.wait();
			EXPECT(two.error().empty());
			EXPECT(two.samples_per_pixel_min() == 60u && two.samples_per_pixel_max() == 60u);
			EXPECT(two.accumulator_rgb() == first_acc);
			PT::Pathtracer three;
			three.set_devices({0, 0, 0});
			three.set_report_interval(0.0f);
			int reports = 0, final_reports = 0;
			float last_fraction = 0.0f;
			three.render(desc, cam, [&](PT::Pathtracer::Render_Report&& r) { reports++, final_reports += r.first == 1.0f, last_fraction = r.first; }, &quit);
			three.wait();
			// every finished render ends with exactly one report at 1.0 (pathtracer.cpp:574-579), whichever worker finished last
			EXPECT(three.error().empty() && reports >= 1 && final_reports == 1 && last_fraction == 1.0f);
			EXPECT(three.accumulator_rgb() == first_acc);
			// the films added by RCCL (a one-rank communicator on a one-GPU box) instead of peer reads: same frame
			PT::Pathtracer rccl;
			rccl.set_reduce(PT::Pathtracer::Reduce::Rccl);
			rccl.set_tiles_per_batch(4);
			rccl.set_report_interval(0.0f);
			reports = 0;
			rccl.render(desc, cam, [&](PT::Pathtracer::Render_Report&&) { reports++; }, &quit);
			rccl.wait();
			EXPECT(rccl.error().empty() && reports == 5);
			EXPECT(rccl.accumulator_rgb() == first_acc);
		}

		// add_samples: the same film gets 60 more samples per pixel -- NEW samples (the sample index continues), so the
		// accumulator is not twice the first pass, and the estimate stays where it was
		progress.clear();
		pt.render(desc, cam, [&](PT::Pathtracer::Render_Report&& r) {
			progress.push_back(r.first);
			double m = 0.0;
			for (float v : r.second.rgb) m += v;
			last_mean = m / double(r.second.rgb.size());
		}, &quit, true);
		pt.wait();
		EXPECT(pt.error().empty() && progress.back() == 1.0f);
		EXPECT(pt.samples_per_pixel_min() == 120u && pt.samples_per_pixel_max() == 120u);
		EXPECT(std::fabs(last_mean - first_mean) < 0.05 * first_mean);
		{
			const std::vector<int64_t>& acc2 = pt.accumulator_rgb();
			size_t doubled = 0, lit = 0;
			for (size_t i = 0; i < acc2.size(); i++)
				if (first_acc[i] != 0) lit++, doubled += acc2[i] == 2 * first_acc[i];
			// a replay of the same samples would double every entry; fresh samples double one only by chance (a floor pixel's
			// sum is 4 x the number of its samples that reached the lamp, so equal counts in both passes are not rare)
			EXPECT(lit > 1000 && doubled < lit / 2);
			// ... and it equals one 120-sample render of a fresh film with the same seed: same streams, same sums per tile
			PT::Pathtracer whole;
			PtxCamera cam120 = cam;
			cam120.film_samples = 120;
			whole.render(desc, cam120, nullptr, &quit);
			whole.wait();
			EXPECT(whole.error().empty() && whole.samples_per_pixel_min() == 120u);
			double m = 0.0;
			for (float v : whole.accumulator_to_image().rgb) m += v;
			EXPECT(std::fabs(m / double(acc2.size()) - last_mean) < 1e-4 * last_mean);
		}

		// quit set before the first batch: nothing is traced, the film is cleared (no add_samples) and stays empty
		quit = true;
		progress.clear();
		pt.render(desc, cam, [&](PT::Pathtracer::Render_Report&& r) { progress.push_back(r.first); }, &quit);
		pt.wait();
		EXPECT(pt.error().empty() && progress.empty());
		EXPECT(pt.samples_per_pixel_max() == 0u);
	}
	std::printf("facade gpu checks passed\n");
	return 0;
}

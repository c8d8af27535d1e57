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
	std::printf("facade gpu checks passed\n");
	return 0;
}

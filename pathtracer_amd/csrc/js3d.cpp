// js3d.cpp -- reader for the reference's JSON scene files (".js3d", FORMAT "js3d-v1"), the
// read-only subset a render needs.
//
// Follows scene/io.cpp:15-70 (root object with "scene"), scene/load-save-json.cpp:108-333
// (storages are JSON objects keyed by name, references are names), util/to_json.cpp:301-502
// (half-edge mesh blobs "s3d-hm-1": unpadded base64 with a type prefix), geometry/indexed.cpp:10-34
// (SplitEdges: every face corner becomes a vertex, faces become triangle fans),
// scene/transform.cpp:4-24 + lib/quat.h:128-134 (translate * rotate * scale, parent chain) and
// Pathtracer::build_scene (pathtracer.cpp:182-387: which instances become objects and lights).
//
// Names are processed in sorted order: the reference iterates std::unordered_map storages, whose
// order is an implementation detail of its standard library; no result depends on it except the
// index an instance gets.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ptx.h"
#include "json.h"

namespace ptx {
int fail(int code, const std::string& msg);   // records the message for ptx_last_error()
}

namespace ptx {
namespace {

using json::Value;

// ---- base64 blobs (util/to_json.cpp:542-635)
std::vector<uint8_t> decode_blob(const Value& v, const char* type) {
	if (!v.is_string()) throw std::runtime_error("blob is not a string");
	const std::string& s = v.str;
	const size_t tl = std::strlen(type);
	if (s.compare(0, tl, type) != 0) throw std::runtime_error(std::string("blob does not start with '") + type + "'");
	std::vector<uint8_t> out(((s.size() - tl) * 6) / 8);
	uint32_t buffer = 0, bits = 0;
	size_t o = 0;
	for (size_t i = tl; i < s.size(); i++) {
		const char c = s[i];
		uint32_t d;
		if (c >= 'A' && c <= 'Z') d = uint32_t(c - 'A');
		else if (c >= 'a' && c <= 'z') d = uint32_t(c - 'a') + 26;
		else if (c >= '0' && c <= '9') d = uint32_t(c - '0') + 52;
		else if (c == '+') d = 62;
		else if (c == '/') d = 63;
		else throw std::runtime_error("invalid base64 character");
		buffer = (buffer << 6) | d, bits += 6;
		if (bits >= 8) {
			if (o < out.size()) out[o++] = uint8_t(buffer >> (bits - 8));
			bits -= 8;
		}
	}
	return out;
}

template<typename T>
std::vector<T> decode_vector(const Value& obj, const char* key, const char* type) {
	const Value* v = obj.find(key);
	if (!v) return {};
	const std::vector<uint8_t> bytes = decode_blob(*v, type);
	if (bytes.size() % sizeof(T) != 0) throw std::runtime_error(std::string(key) + ": byte count is not a multiple of the item size");
	std::vector<T> out(bytes.size() / sizeof(T));
	if (!out.empty()) std::memcpy(out.data(), bytes.data(), bytes.size());
	return out;
}

std::vector<bool> decode_bools(const Value& obj, const char* key) {   // packed MSB first, trailing 1 + zero padding trimmed
	const Value* v = obj.find(key);
	if (!v) return {};
	const std::vector<uint8_t> packed = decode_blob(*v, "bool:");
	std::vector<bool> out;
	for (uint8_t byte : packed)
		for (int bit = 7; bit >= 0; bit--) out.push_back((byte >> bit) & 1);
	while (!out.empty()) {
		const bool back = out.back();
		out.pop_back();
		if (back) break;
	}
	return out;
}

struct FlatMesh {
	std::vector<float> verts;       // 8 floats per corner vertex
	std::vector<uint32_t> indices;
};

// Half-edge blob -> corner-split triangle fans.
FlatMesh flatten_mesh(const Value& m) {
	std::vector<uint32_t> nexts = decode_vector<uint32_t>(m, "halfedge_nexts", "uint32:");
	std::vector<float> uvs = decode_vector<float>(m, "halfedge_corner_uvs", "vec2:");
	std::vector<float> normals = decode_vector<float>(m, "halfedge_corner_normals", "vec3:");
	std::vector<uint32_t> vertex_halfedges = decode_vector<uint32_t>(m, "vertex_halfedges", "uint32:");
	std::vector<float> positions = decode_vector<float>(m, "vertex_positions", "vec3:");
	std::vector<uint32_t> face_halfedges = decode_vector<uint32_t>(m, "face_halfedges", "uint32:");
	std::vector<bool> face_boundaries = decode_bools(m, "face_boundaries");
	const size_t nh = nexts.size();
	if (nh % 2 != 0) throw std::runtime_error("halfedge without twin");
	uvs.resize(nh * 2, 0.0f), normals.resize(nh * 3, 0.0f);
	positions.resize(vertex_halfedges.size() * 3, 0.0f);
	face_boundaries.resize(face_halfedges.size(), false);
	std::vector<uint8_t> seen(nh, 0);
	for (uint32_t n : nexts) {
		if (n >= nh) throw std::runtime_error("halfedge with out-of-range next");
		if (seen[n]++) throw std::runtime_error("two halfedges with the same next");
	}
	// a vertex owns the half-edges reached by h -> twin(h).next from its half-edge; twin(i) = i ^ 1
	std::vector<uint32_t> vertex_of(nh, 0xFFFFFFFFu);
	for (uint32_t v = 0; v < vertex_halfedges.size(); v++) {
		const uint32_t first = vertex_halfedges[v];
		if (first >= nh) throw std::runtime_error("vertex with out-of-range halfedge");
		uint32_t h = first;
		do {
			if (vertex_of[h] != 0xFFFFFFFFu) throw std::runtime_error("two vertices which claim the same halfedge");
			vertex_of[h] = v;
			h = nexts[h ^ 1u];
		} while (h != first);
	}
	FlatMesh out;
	for (size_t f = 0; f < face_halfedges.size(); f++) {
		if (face_boundaries[f]) continue;
		const uint32_t first = face_halfedges[f];
		if (first >= nh) throw std::runtime_error("face with out-of-range halfedge");
		const uint32_t corners_begin = uint32_t(out.verts.size() / 8);
		uint32_t h = first;
		size_t guard = 0;
		do {
			const uint32_t v = vertex_of[h];
			if (v == 0xFFFFFFFFu) throw std::runtime_error("halfedge without a vertex");
			const float rec[8] = {positions[v * 3], positions[v * 3 + 1], positions[v * 3 + 2], normals[h * 3], normals[h * 3 + 1], normals[h * 3 + 2],
			                      uvs[h * 2], uvs[h * 2 + 1]};
			out.verts.insert(out.verts.end(), rec, rec + 8);
			h = nexts[h];
			if (++guard > nh) throw std::runtime_error("face loop does not close");
		} while (h != first);
		const uint32_t corners_end = uint32_t(out.verts.size() / 8);
		for (uint32_t i = corners_begin + 1; i + 1 < corners_end; i++) {
			out.indices.push_back(corners_begin), out.indices.push_back(i), out.indices.push_back(i + 1);
		}
	}
	return out;
}

// ---- 4x4 matrices, column major, with the reference's operation order (lib/mat4.h:162-173)
struct M4 {
	float c[4][4];
};
M4 identity() {
	M4 m{};
	for (int i = 0; i < 4; i++) m.c[i][i] = 1.0f;
	return m;
}
M4 mul(const M4& a, const M4& b) {   // a * b
	M4 r;
	for (int i = 0; i < 4; i++)
		for (int j = 0; j < 4; j++) {
			float acc = 0.0f;
			for (int k = 0; k < 4; k++) acc += b.c[i][k] * a.c[k][j];
			r.c[i][j] = acc;
		}
	return r;
}

struct TransformRec {
	std::string parent;
	float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
};

M4 local_to_parent(const TransformRec& tr) {   // translate * rotation.to_mat() * scale (scene/transform.cpp:4-6)
	M4 T = identity(), R = identity(), S = identity();
	T.c[3][0] = tr.t[0], T.c[3][1] = tr.t[1], T.c[3][2] = tr.t[2];
	const float x = tr.q[0], y = tr.q[1], z = tr.q[2], w = tr.q[3];
	R.c[0][0] = 1 - 2 * y * y - 2 * z * z, R.c[0][1] = 2 * x * y + 2 * z * w, R.c[0][2] = 2 * x * z - 2 * y * w;
	R.c[1][0] = 2 * x * y - 2 * z * w, R.c[1][1] = 1 - 2 * x * x - 2 * z * z, R.c[1][2] = 2 * y * z + 2 * x * w;
	R.c[2][0] = 2 * x * z + 2 * y * w, R.c[2][1] = 2 * y * z - 2 * x * w, R.c[2][2] = 1 - 2 * x * x - 2 * y * y;
	S.c[0][0] = tr.s[0], S.c[1][1] = tr.s[1], S.c[2][2] = tr.s[2];
	return mul(mul(T, R), S);
}

void read_floats(const Value* v, float* out, size_t n) {
	if (!v || v->kind != Value::Array) return;
	for (size_t i = 0; i < n && i < v->items.size(); i++)
		if (v->items[i].is_number()) out[i] = float(v->items[i].num);
}
float number_or(const Value& obj, const char* key, float fallback) {
	const Value* v = obj.find(key);
	return v && v->is_number() ? float(v->num) : fallback;
}
std::string string_or(const Value& obj, const char* key, const std::string& fallback = "") {
	const Value* v = obj.find(key);
	return v && v->is_string() ? v->str : fallback;
}
bool bool_or(const Value& obj, const char* key, bool fallback) {
	const Value* v = obj.find(key);
	return v && v->kind == Value::Bool ? v->b : fallback;
}

std::vector<std::pair<std::string, const Value*>> sorted_members(const Value* obj) {
	std::vector<std::pair<std::string, const Value*>> out;
	if (obj && obj->kind == Value::Object)
		for (const auto& m : obj->members) out.emplace_back(m.first, &m.second);
	std::sort(out.begin(), out.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
	return out;
}

}  // namespace
}  // namespace ptx

struct PtxJs3d {
	std::vector<ptx::FlatMesh> mesh_data;
	std::vector<std::string> mesh_names;
	std::vector<std::vector<float>> texel_data;
	std::vector<PtxMeshDesc> meshes;
	std::vector<PtxTextureDesc> textures;
	std::vector<PtxMaterialDesc> materials;
	std::vector<PtxInstanceDesc> instances;
	std::vector<std::string> instance_names;
	std::vector<PtxDeltaLightDesc> delta_lights;
	std::vector<PtxEnvLightDesc> env_lights;
	std::vector<std::pair<std::string, PtxCamera>> cameras;
	PtxSceneDesc desc{};
	std::string warnings;
};

namespace ptx {
namespace {

void load_into(PtxJs3d& out, const std::string& path) {
	std::ifstream file(path, std::ios::binary);
	if (!file) throw std::runtime_error("cannot open file");
	std::stringstream ss;
	ss << file.rdbuf();
	const std::string text = ss.str();
	if (!text.empty() && text[0] == 's') throw std::runtime_error("binary .s3d scenes are not supported, only js3d");
	const Value root = json::parse(text);
	const Value* scene = root.find("scene");
	if (!scene) throw std::runtime_error("file does not contain a scene");
	auto warn = [&](const std::string& w) { out.warnings += w + "\n"; };

	// transforms
	std::map<std::string, TransformRec> transforms;
	for (const auto& [name, v] : sorted_members(scene->find("transforms"))) {
		TransformRec tr;
		const Value* p = v->find("parent");
		if (p && p->is_string()) tr.parent = p->str;
		read_floats(v->find("translation"), tr.t, 3);
		read_floats(v->find("rotation"), tr.q, 4);
		read_floats(v->find("scale"), tr.s, 3);
		transforms[name] = tr;
	}
	auto local_to_world = [&](const std::string& name) {
		// parent->local_to_world() * local_to_parent() (scene/transform.cpp:12-24): products are
		// formed from the root of the chain downwards, like the reference's recursion does
		std::vector<const TransformRec*> chain;
		for (std::string cur = name; !cur.empty();) {
			auto it = transforms.find(cur);
			if (it == transforms.end()) throw std::runtime_error("transform '" + cur + "' is not in storage");
			chain.push_back(&it->second);
			cur = it->second.parent;
			if (chain.size() > transforms.size()) throw std::runtime_error("transform parents form a cycle");
		}
		if (chain.empty()) return identity();
		M4 m = local_to_parent(*chain.back());
		for (size_t i = chain.size() - 1; i-- > 0;) m = mul(m, local_to_parent(*chain[i]));
		return m;
	};
	auto put_matrix = [](const M4& m, float* dst) { std::memcpy(dst, m.c, sizeof(float) * 16); };

	// meshes
	std::map<std::string, int32_t> mesh_index;
	for (const auto& [name, v] : sorted_members(scene->find("meshes"))) {
		try {
			out.mesh_data.push_back(flatten_mesh(*v));
		} catch (const std::exception& e) {
			throw std::runtime_error("mesh '" + name + "': " + e.what());
		}
		mesh_index[name] = int32_t(out.mesh_data.size() - 1);
		out.mesh_names.push_back(name);
	}
	if (scene->find("skinned_meshes") && !scene->find("skinned_meshes")->members.empty()) warn("skinned meshes are not read");

	// textures
	std::map<std::string, int32_t> texture_index;
	for (const auto& [name, v] : sorted_members(scene->find("textures"))) {
		PtxTextureDesc t{};
		t.scale = 1.0f;
		const std::string type = string_or(*v, "type");
		if (type == "Constant") {
			t.type = PTX_TEX_CONSTANT;
			read_floats(v->find("color"), t.color, 3);
			t.scale = number_or(*v, "scale", 1.0f);
		} else if (type == "Image") {
			t.type = PTX_TEX_IMAGE;
			// "trilinear" without a level of detail is a bilinear lookup of level 0 (scene/texture.cpp:68-80)
			t.sampler = string_or(*v, "sampler", "bilinear") == "nearest" ? PTX_SAMPLER_NEAREST : PTX_SAMPLER_BILINEAR;
			const Value* img = v->find("image");
			if (!img || !img->is_string() || img->str.compare(0, 6, "hdr64:") != 0) throw std::runtime_error("texture '" + name + "': only embedded hdr64 images are read");
			const std::vector<uint8_t> bytes = decode_blob(*img, "hdr64:");   // util/hdr_image.cpp:108-133
			if (bytes.size() < 12 || std::memcmp(bytes.data(), "rawf", 4) != 0) throw std::runtime_error("texture '" + name + "': unrecognised image format");
			std::memcpy(&t.w, bytes.data() + 4, 4), std::memcpy(&t.h, bytes.data() + 8, 4);
			if (bytes.size() - 12 != size_t(t.w) * t.h * 12) throw std::runtime_error("texture '" + name + "': wrong byte count for a raw float image");
			out.texel_data.emplace_back(size_t(t.w) * t.h * 3);
			std::memcpy(out.texel_data.back().data(), bytes.data() + 12, bytes.size() - 12);
		} else {
			throw std::runtime_error("texture '" + name + "': unknown type '" + type + "'");
		}
		texture_index[name] = int32_t(out.textures.size());
		out.textures.push_back(t);
	}
	auto tex_ref = [&](const Value& obj, const char* key) -> int32_t {
		const Value* v = obj.find(key);
		if (!v || !v->is_string()) return -1;
		auto it = texture_index.find(v->str);
		return it == texture_index.end() ? -1 : it->second;
	};

	// materials
	std::map<std::string, int32_t> material_index;
	for (const auto& [name, v] : sorted_members(scene->find("materials"))) {
		PtxMaterialDesc m{};
		m.tex_a = m.tex_b = -1, m.ior = 1.5f;
		const std::string type = string_or(*v, "type");
		if (type == "Lambertian") m.type = PTX_MAT_LAMBERTIAN, m.tex_a = tex_ref(*v, "albedo");
		else if (type == "Mirror") m.type = PTX_MAT_MIRROR, m.tex_a = tex_ref(*v, "reflectance");
		else if (type == "Refract") m.type = PTX_MAT_REFRACT, m.tex_b = tex_ref(*v, "transmittance"), m.ior = number_or(*v, "ior", 1.5f);
		else if (type == "Glass") m.type = PTX_MAT_GLASS, m.tex_a = tex_ref(*v, "reflectance"), m.tex_b = tex_ref(*v, "transmittance"), m.ior = number_or(*v, "ior", 1.5f);
		else if (type == "Emissive") m.type = PTX_MAT_EMISSIVE, m.tex_a = tex_ref(*v, "emissive");
		else throw std::runtime_error("material '" + name + "': unknown type '" + type + "'");
		material_index[name] = int32_t(out.materials.size());
		out.materials.push_back(m);
	}
	auto material_ref = [&](const Value& obj) -> int32_t {
		const Value* v = obj.find("material");
		if (!v || !v->is_string()) return -1;   // default material (pathtracer.cpp:240-252)
		auto it = material_index.find(v->str);
		return it == material_index.end() ? -1 : it->second;
	};

	// shapes (Shapes::Sphere only, scene/shape.h)
	std::map<std::string, float> shape_radius;
	for (const auto& [name, v] : sorted_members(scene->find("shapes"))) shape_radius[name] = number_or(*v, "radius", 1.0f);

	// delta lights and environment lights
	std::map<std::string, PtxDeltaLightDesc> delta_lights;
	for (const auto& [name, v] : sorted_members(scene->find("delta_lights"))) {
		PtxDeltaLightDesc l{};
		const std::string type = string_or(*v, "type");
		l.type = type == "Directional" ? PTX_LIGHT_DIRECTIONAL : type == "Spot" ? PTX_LIGHT_SPOT : PTX_LIGHT_POINT;
		l.color[0] = l.color[1] = l.color[2] = 1.0f;
		read_floats(v->find("color"), l.color, 3);
		l.intensity = number_or(*v, "intensity", 1.0f);
		l.inner_angle = number_or(*v, "inner_angle", 30.0f), l.outer_angle = number_or(*v, "outer_angle", 45.0f);
		delta_lights[name] = l;
	}
	// every environment light resource lights the scene, instanced or not (pathtracer.cpp:259-276, 130-136)
	for (const auto& [name, v] : sorted_members(scene->find("env_lights"))) {
		PtxEnvLightDesc l{};
		l.type = string_or(*v, "type") == "Hemisphere" ? PTX_ENV_HEMISPHERE : PTX_ENV_SPHERE;
		l.tex = tex_ref(*v, "radiance");
		if (l.tex < 0) {
			warn("environment light '" + name + "' has no radiance texture; skipped");
			continue;
		}
		out.env_lights.push_back(l);
	}

	// instances (pathtracer.cpp:284-375): meshes, then shapes; invisible or dangling ones are skipped
	const Value* instances = scene->find("instances");
	const Value none;
	auto section = [&](const char* key) { return instances ? instances->find(key) : nullptr; };
	for (const auto& [name, v] : sorted_members(section("meshes"))) {
		if (!bool_or(*v, "visible", true)) continue;
		auto it = mesh_index.find(string_or(*v, "mesh"));
		if (it == mesh_index.end()) continue;
		PtxInstanceDesc in{};
		in.mesh = it->second, in.sphere_radius = 1.0f, in.material = material_ref(*v);
		put_matrix(local_to_world(string_or(*v, "transform")), in.T);
		out.instances.push_back(in), out.instance_names.push_back(name);
	}
	for (const auto& [name, v] : sorted_members(section("shapes"))) {
		if (!bool_or(*v, "visible", true)) continue;
		auto it = shape_radius.find(string_or(*v, "shape"));
		if (it == shape_radius.end()) continue;
		PtxInstanceDesc in{};
		in.mesh = -1, in.sphere_radius = it->second, in.material = material_ref(*v);
		put_matrix(local_to_world(string_or(*v, "transform")), in.T);
		out.instances.push_back(in), out.instance_names.push_back(name);
	}
	if (section("skinned_meshes") && !section("skinned_meshes")->members.empty()) warn("skinned mesh instances are not read");
	if (section("particles") && !section("particles")->members.empty()) warn("particle instances are not read");
	for (const auto& [name, v] : sorted_members(section("delta_lights"))) {
		if (!bool_or(*v, "visible", true)) continue;
		auto it = delta_lights.find(string_or(*v, "light"));
		if (it == delta_lights.end()) continue;
		PtxDeltaLightDesc l = it->second;
		put_matrix(local_to_world(string_or(*v, "transform")), l.T);
		out.delta_lights.push_back(l);
	}

	// cameras (scene/camera.h:18-30 defaults)
	std::map<std::string, const Value*> camera_resources;
	for (const auto& [name, v] : sorted_members(scene->find("cameras"))) camera_resources[name] = v;
	for (const auto& [name, v] : sorted_members(section("cameras"))) {
		auto it = camera_resources.find(string_or(*v, "camera"));
		if (it == camera_resources.end()) continue;
		const Value& c = *it->second;
		PtxCamera cam{};
		cam.vertical_fov = number_or(c, "vertical_fov", 60.0f), cam.aspect_ratio = number_or(c, "aspect_ratio", 1.77778f);
		cam.film_width = uint32_t(number_or(c, "film.width", 1280)), cam.film_height = uint32_t(number_or(c, "film.height", 720));
		cam.film_samples = uint32_t(number_or(c, "film.samples", 256)), cam.film_max_ray_depth = uint32_t(number_or(c, "film.max_ray_depth", 8));
		put_matrix(local_to_world(string_or(*v, "transform")), cam.camera_to_world);
		out.cameras.emplace_back(name, cam);
	}

	// descriptor over the owned arrays
	size_t image = 0;
	for (auto& t : out.textures)
		if (t.type == PTX_TEX_IMAGE) t.pixels = out.texel_data[image++].data();
	for (const FlatMesh& m : out.mesh_data) {
		PtxMeshDesc d{};
		d.verts = m.verts.data(), d.n_verts = uint32_t(m.verts.size() / 8);
		d.indices = m.indices.data(), d.n_tris = uint32_t(m.indices.size() / 3);
		d.use_bvh = 1, d.max_leaf_size = 4;
		out.meshes.push_back(d);
	}
	PtxSceneDesc& d = out.desc;
	d.meshes = out.meshes.data(), d.n_meshes = uint32_t(out.meshes.size());
	d.instances = out.instances.data(), d.n_instances = uint32_t(out.instances.size());
	d.textures = out.textures.data(), d.n_textures = uint32_t(out.textures.size());
	d.materials = out.materials.data(), d.n_materials = uint32_t(out.materials.size());
	d.delta_lights = out.delta_lights.data(), d.n_delta_lights = uint32_t(out.delta_lights.size());
	d.env_lights = out.env_lights.data(), d.n_env_lights = uint32_t(out.env_lights.size());
	d.use_bvh = 1, d.fresnel_mode = PTX_FRESNEL_GCC_INT_ABS;
}

}  // namespace
}  // namespace ptx

extern "C" {

int ptx_js3d_load(const char* path, ptx_js3d_t* out) {
	if (!path || !out) return ptx::fail(PTX_ERR_INVALID, "ptx_js3d_load: null argument");
	auto* s = new PtxJs3d();
	try {
		ptx::load_into(*s, path);
	} catch (const std::exception& e) {
		delete s;
		return ptx::fail(PTX_ERR_INVALID, std::string("Failed to load '") + path + "' as js3d: " + e.what());
	}
	*out = s;
	return PTX_OK;
}

void ptx_js3d_free(ptx_js3d_t s) { delete s; }

const PtxSceneDesc* ptx_js3d_scene_desc(ptx_js3d_t s) { return s ? &s->desc : nullptr; }

void ptx_js3d_set_modes(ptx_js3d_t s, uint32_t use_bvh, uint32_t fresnel_mode) {
	if (!s) return;
	s->desc.use_bvh = use_bvh ? 1 : 0, s->desc.fresnel_mode = fresnel_mode;
	for (auto& m : s->meshes) m.use_bvh = use_bvh ? 1 : 0;   // --no_bvh applies to meshes and the instance level alike (pathtracer.cpp:217,381-385)
}

uint32_t ptx_js3d_num_cameras(ptx_js3d_t s) { return s ? uint32_t(s->cameras.size()) : 0; }
const char* ptx_js3d_camera_name(ptx_js3d_t s, uint32_t i) { return s && i < s->cameras.size() ? s->cameras[i].first.c_str() : nullptr; }

int ptx_js3d_get_camera(ptx_js3d_t s, const char* name, PtxCamera* out) {
	if (!s || !name || !out) return ptx::fail(PTX_ERR_INVALID, "ptx_js3d_get_camera: null argument");
	for (const auto& c : s->cameras)
		if (c.first == name) {
			*out = c.second;
			return PTX_OK;
		}
	return ptx::fail(PTX_ERR_INVALID, std::string("Failed to find camera: ") + name);
}

uint32_t ptx_js3d_num_meshes(ptx_js3d_t s) { return s ? uint32_t(s->mesh_names.size()) : 0; }
const char* ptx_js3d_mesh_name(ptx_js3d_t s, uint32_t i) { return s && i < s->mesh_names.size() ? s->mesh_names[i].c_str() : nullptr; }
const char* ptx_js3d_instance_name(ptx_js3d_t s, uint32_t i) { return s && i < s->instance_names.size() ? s->instance_names[i].c_str() : nullptr; }
const char* ptx_js3d_warnings(ptx_js3d_t s) { return s ? s->warnings.c_str() : ""; }

}  // extern "C"

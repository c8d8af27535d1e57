// ptx_render -- headless renderer with the reference's command line (main.cpp:50-75, 96-358):
//   -s,--scene FILE  -c,--camera NAME  -o,--output FILE.png  --trace  --no_bvh  --exposure F  --seed N
//   --film-width N  --film-height N  --film-samples N  --film-max-ray-depth N
// plus  --device N  (one GPU),  --devices N  (GPUs 0..N-1: tiles dealt across them, films added at the end;
// --devices a,b,c names them),  --reduce peer|rccl  (how the GPUs' films are added: the first GPU reading the others'
// films over their links, or ncclReduce over a communicator with one rank per GPU; same frame bit for bit)  and
// --fresnel gcc|float  (which behaviour of Materials::schlick to reproduce, scene/material.cpp:58-71).  Exit codes follow the reference: 0 on success, 1 on any reported error.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <random>
#include <string>
#include <thread>

#include "../host/pt_facade.h"

static void usage() {
	std::cerr << "usage: ptx_render --trace -s scene.js3d -c camera_instance [-o out.png] [--no_bvh] [--exposure f] [--seed n]\n"
	             "                  [--film-width n] [--film-height n] [--film-samples n] [--film-max-ray-depth n] [--device n | --devices n | --devices a,b,..] [--reduce peer|rccl] [--fresnel gcc|float]\n";
}

int main(int argc, char** argv) {
	std::string scene_file, camera_name, output_file, fresnel = "gcc", reduce = "peer";
	bool pathtrace = false, no_bvh = false;
	float exposure = 1.0f;
	uint32_t seed = 0, film_width = -1U, film_height = -1U, film_samples = -1U, film_max_ray_depth = -1U;
	std::vector<int> devices{0};
	for (int i = 1; i < argc; i++) {
		const std::string a = argv[i];
		auto value = [&]() -> const char* {
			if (i + 1 >= argc) {
				std::cerr << "ERROR: " << a << " needs a value." << std::endl;
				std::exit(1);
			}
			return argv[++i];
		};
		if (a == "-s" || a == "--scene") scene_file = value();
		else if (a == "-c" || a == "--camera") camera_name = value();
		else if (a == "-o" || a == "--output") output_file = value();
		else if (a == "--trace") pathtrace = true;
		else if (a == "--no_bvh") no_bvh = true;
		else if (a == "--exposure") exposure = std::strtof(value(), nullptr);
		else if (a == "--seed") seed = uint32_t(std::strtoul(value(), nullptr, 0));
		else if (a == "--film-width") film_width = uint32_t(std::strtoul(value(), nullptr, 0));
		else if (a == "--film-height") film_height = uint32_t(std::strtoul(value(), nullptr, 0));
		else if (a == "--film-samples") film_samples = uint32_t(std::strtoul(value(), nullptr, 0));
		else if (a == "--film-max-ray-depth") film_max_ray_depth = uint32_t(std::strtoul(value(), nullptr, 0));
		else if (a == "--device") devices = {std::atoi(value())};
		else if (a == "--devices") {
			const std::string v = value();
			devices.clear();
			if (v.find(',') == std::string::npos) {
				for (int d = 0; d < std::atoi(v.c_str()); d++) devices.push_back(d);
			} else {
				for (size_t pos = 0; pos <= v.size();) {
					const size_t end = std::min(v.find(',', pos), v.size());
					devices.push_back(std::atoi(v.substr(pos, end - pos).c_str()));
					pos = end + 1;
				}
			}
			if (devices.empty()) {
				std::cerr << "ERROR: --devices needs a count of at least 1 or a list." << std::endl;
				return 1;
			}
		}
		else if (a == "--fresnel") fresnel = value();
		else if (a == "--reduce") {
			reduce = value();
			if (reduce != "peer" && reduce != "rccl") {
				std::cerr << "ERROR: --reduce takes peer or rccl." << std::endl;
				return 1;
			}
		}
		else if (a == "-h" || a == "--help") {
			usage();
			return 0;
		} else {
			std::cerr << "ERROR: unknown option " << a << std::endl;
			usage();
			return 1;
		}
	}
	if (!pathtrace) {
		std::cerr << "ERROR: only --trace is available (no GUI, no rasterizer)." << std::endl;
		return 1;
	}
	if (scene_file.empty()) {
		std::cerr << "ERROR: must specify a scene file via --scene when doing --trace or --rasterize or --write." << std::endl;
		return 1;
	}
	ptx_js3d_t js = nullptr;
	if (ptx_js3d_load(scene_file.c_str(), &js) != PTX_OK) {
		std::cerr << "ERROR: Failed to load scene '" << scene_file << "': " << ptx_last_error() << std::endl;
		return 1;
	}
	if (*ptx_js3d_warnings(js)) std::cerr << ptx_js3d_warnings(js);
	PtxCamera cam{};
	if (ptx_js3d_get_camera(js, camera_name.c_str(), &cam) != PTX_OK) {   // main.cpp:125-134
		std::cerr << "ERROR: Failed to find camera: " << camera_name << std::endl << "Camera instances in scene:";
		for (uint32_t i = 0; i < ptx_js3d_num_cameras(js); i++) std::cerr << "\n    '" << ptx_js3d_camera_name(js, i) << "'";
		std::cerr << std::endl;
		ptx_js3d_free(js);
		return 1;
	}
	// film overrides (main.cpp:139-162)
	if (film_width != -1U && film_height != -1U) {
		cam.film_width = film_width, cam.film_height = film_height;
		cam.aspect_ratio = cam.film_width / float(cam.film_height);
		std::cout << "  Set film size to [" << cam.film_width << "x" << cam.film_height << "]." << std::endl;
	} else if (film_width != -1U) {
		cam.film_width = film_width, cam.film_height = uint32_t(std::round(cam.film_width / cam.aspect_ratio));
		std::cout << "  Set film size to [" << cam.film_width << "x" << cam.film_height << "] (height determined from aspect ratio)." << std::endl;
	} else if (film_height != -1U) {
		cam.film_height = film_height, cam.film_width = uint32_t(std::round(cam.film_height * cam.aspect_ratio));
		std::cout << "  Set film size to [" << cam.film_width << "x" << cam.film_height << "] (width determined from aspect ratio)." << std::endl;
	}
	if (film_samples != -1U) {
		cam.film_samples = film_samples;
		std::cout << "  Set film path tracer samples to " << cam.film_samples << "." << std::endl;
	}
	if (film_max_ray_depth != -1U) {
		cam.film_max_ray_depth = film_max_ray_depth;
		std::cout << "  Set film max ray depth to " << cam.film_max_ray_depth << "." << std::endl;
	}
	if (seed == 0) seed = (std::random_device())();   // main.cpp:186-188
	ptx_js3d_set_modes(js, no_bvh ? 0 : 1, fresnel == "float" ? PTX_FRESNEL_FLOAT_ABS : PTX_FRESNEL_GCC_INT_ABS);

	std::printf("Render settings:\n\twidth: %u\n\theight: %u\n\texposure: %f\n\tseed: 0x%X\n\tsamples: %u\n\tmax depth: %u\n\tdevices:", cam.film_width,
	            cam.film_height, exposure, seed, cam.film_samples, cam.film_max_ray_depth);
	for (int d : devices) std::printf(" %d", d);
	if (devices.size() > 1) std::printf("\n\tfilm reduce: %s", reduce.c_str());
	std::printf("\n");
	if (no_bvh) std::printf("\tusing object list instead of BVH\n");
	std::printf("\tpathtracing...\n");

	std::mutex report_mut;
	float percent_done = 0.0f;
	PT::HDR_Image display_hdr;
	bool quit = false;
	int status = 0;
	try {
		PT::Pathtracer pathtracer;
		pathtracer.set_devices(devices);
		pathtracer.set_reduce(reduce == "rccl" ? PT::Pathtracer::Reduce::Rccl : PT::Pathtracer::Reduce::Peer);
		pathtracer.set_seed(seed);
		pathtracer.use_bvh(!no_bvh);
		pathtracer.render(*ptx_js3d_scene_desc(js), cam,
		                  [&](PT::Pathtracer::Render_Report&& report) {
			                  std::lock_guard<std::mutex> lock(report_mut);
			                  if (report.first > percent_done) percent_done = report.first, display_hdr = std::move(report.second);
		                  },
		                  &quit);
		while (pathtracer.in_progress()) {
			{
				std::lock_guard<std::mutex> lock(report_mut);
				std::printf("Progress: %6.2f%%    \r", 100.0f * percent_done);
				std::fflush(stdout);
			}
			std::this_thread::sleep_for(std::chrono::milliseconds(250));
		}
		pathtracer.wait();
		std::printf("\n");
		if (!pathtracer.error().empty()) {
			std::cerr << "ERROR: " << pathtracer.error() << std::endl;
			status = 1;
		} else {
			const auto [build_s, render_s] = pathtracer.completion_time();
			const double paths = double(cam.film_width) * cam.film_height * cam.film_samples;
			std::printf("\tdone. scene build %.3f s, render %.3f s (%.2f Mspp/s)\n", build_s, render_s, paths / render_s / 1e6);
		}
	} catch (const std::exception& e) {
		std::cerr << "ERROR: " << e.what() << std::endl;
		status = 1;
	}
	if (status == 0) {
		if (output_file.empty()) {
			std::cout << "No output was requested, not writing any file." << std::endl;
		} else {
			std::vector<uint8_t> data;
			display_hdr.tonemap_to(data, exposure);
			if (ptx_write_png(output_file.c_str(), data.data(), display_hdr.w, display_hdr.h, 1) != PTX_OK) {   // main.cpp:339-343
				std::cerr << "ERROR: Failed to write output to '" << output_file << "'" << std::endl;
				status = 1;
			} else {
				std::cout << "Wrote result to '" << output_file << "'." << std::endl;
			}
		}
	}
	ptx_js3d_free(js);
	return status;
}

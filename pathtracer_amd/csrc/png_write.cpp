// png_write.cpp -- 8-bit RGBA PNG output (what main.cpp:333-344 does through stb_image_write).
// The image data goes into zlib "stored" blocks (no compression): rendered frames are written
// once, and an encoder without a compressor has nothing to get wrong.
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/ptx.h"

namespace ptx {
int fail(int code, const std::string& msg);
}

namespace {

uint32_t crc32_of(const uint8_t* data, size_t n, uint32_t crc) {
	static uint32_t table[256];
	static bool ready = false;
	if (!ready) {
		for (uint32_t i = 0; i < 256; i++) {
			uint32_t c = i;
			for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
			table[i] = c;
		}
		ready = true;
	}
	crc = ~crc;
	for (size_t i = 0; i < n; i++) crc = table[(crc ^ data[i]) & 0xFFu] ^ (crc >> 8);
	return ~crc;
}

void be32(std::vector<uint8_t>& v, uint32_t x) {
	v.push_back(uint8_t(x >> 24)), v.push_back(uint8_t(x >> 16)), v.push_back(uint8_t(x >> 8)), v.push_back(uint8_t(x));
}

void chunk(std::vector<uint8_t>& file, const char* type, const std::vector<uint8_t>& body) {
	be32(file, uint32_t(body.size()));
	const size_t start = file.size();
	file.insert(file.end(), type, type + 4);
	file.insert(file.end(), body.begin(), body.end());
	be32(file, crc32_of(file.data() + start, file.size() - start, 0));
}

}  // namespace

extern "C" int ptx_write_png(const char* path, const uint8_t* rgba, uint32_t width, uint32_t height, int flip_rows) {
	if (!path || !rgba || !width || !height) return ptx::fail(PTX_ERR_INVALID, "ptx_write_png: null argument or empty image");
	// raw scanlines: filter byte 0 + RGBA
	std::vector<uint8_t> raw;
	raw.reserve(size_t(height) * (size_t(width) * 4 + 1));
	for (uint32_t y = 0; y < height; y++) {
		const uint32_t src = flip_rows ? height - 1 - y : y;
		raw.push_back(0);
		raw.insert(raw.end(), rgba + size_t(src) * width * 4, rgba + size_t(src + 1) * width * 4);
	}
	// zlib stream of stored blocks + Adler-32
	std::vector<uint8_t> z{0x78, 0x01};
	uint32_t a = 1, b = 0;
	for (size_t off = 0; off < raw.size();) {
		const size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
		z.push_back(off + n == raw.size() ? 1 : 0);
		z.push_back(uint8_t(n)), z.push_back(uint8_t(n >> 8)), z.push_back(uint8_t(~n)), z.push_back(uint8_t((~n) >> 8));
		z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
		for (size_t i = 0; i < n; i++) {
			a = (a + raw[off + i]) % 65521u;
			b = (b + a) % 65521u;
		}
		off += n;
	}
	be32(z, (b << 16) | a);
	std::vector<uint8_t> file{0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
	std::vector<uint8_t> ihdr;
	be32(ihdr, width), be32(ihdr, height);
	ihdr.insert(ihdr.end(), {8, 6, 0, 0, 0});   // 8 bits, RGBA, deflate, no filter method, no interlace
	chunk(file, "IHDR", ihdr);
	chunk(file, "IDAT", z);
	chunk(file, "IEND", {});
	FILE* f = std::fopen(path, "wb");
	if (!f) return ptx::fail(PTX_ERR_INVALID, std::string("Failed to write output to '") + path + "'");
	const bool ok = std::fwrite(file.data(), 1, file.size(), f) == file.size();
	std::fclose(f);
	return ok ? PTX_OK : ptx::fail(PTX_ERR_INVALID, std::string("Failed to write output to '") + path + "'");
}

// pxz_encode — the reference CLI's `image_to_pix` flow (src/bin/main.rs:142-175) on the C++ mirror:
//   raw image -> Pixlzr::from_image -> [shrink_by | shrink_directionally] -> save
// Used by tests to compare whole .pixlzr files with the oracle; raw input instead of PNG decode
// (the `image` crate's decoder is outside the path).
//   pxz_encode <raw> <width> <height> <channels> <block_w> <block_h> <mode: none|by|dir> <filter 0..4> <factor> <out.pixlzr>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <iterator>
#include <vector>

#include "../../include/pixlzr.hpp"

int main(int argc, char **argv)
{
	if (argc != 11) {
		std::fprintf(stderr, "usage: %s raw w h c bw bh none|by|dir filter factor out\n", argv[0]);
		return 2;
	}
	try {
		std::ifstream f(argv[1], std::ios::binary);
		std::vector<uint8_t> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
		const uint32_t w = (uint32_t)std::atoi(argv[2]), h = (uint32_t)std::atoi(argv[3]), c = (uint32_t)std::atoi(argv[4]);
		if (raw.size() != (size_t)w * h * c) throw std::runtime_error("raw size does not match w*h*c");
		const pixlzr::ImageView view{raw.data(), w, h, c, w * c};
		pixlzr::Pixlzr pix = pixlzr::Pixlzr::from_image(view, (uint32_t)std::atoi(argv[5]), (uint32_t)std::atoi(argv[6]));
		const auto filter = (pixlzr::FilterType)std::atoi(argv[8]);
		const float factor = (float)std::atof(argv[9]);
		if (!std::strcmp(argv[7], "by")) pix.shrink_by(filter, factor);
		else if (!std::strcmp(argv[7], "dir")) pix.shrink_directionally(filter, factor);
		pix.save(argv[10]);
		std::printf("%zu blocks\n", pix.blocks.size());
		return 0;
	} catch (const std::exception &e) {
		std::fprintf(stderr, "pxz_encode: %s\n", e.what());
		return 1;
	}
}
